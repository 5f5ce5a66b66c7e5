"""Work per ray by bounce: the instrumented kernels' counts of a render at depth k+1 minus those at depth k
are the counts of bounce k alone (same paths, same draws).

usage (GPU box): python tools/per_bounce_work.py scene spp width height max_bounce
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi  # noqa: E402
from raytracer_amd.render import Context  # noqa: E402
from raytracer_amd.scene import HostScene  # noqa: E402

name, spp, w, h, top = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
scene = HostScene.load_frozen(os.path.join(ROOT, "tests", "golden", name + ".rtscene"))
scene.set_image(w, h)
ctx = Context(0)
ctx.upload(scene)
g = lambda st, n: float(st[abi.STAT_NAMES.index(n)])  # noqa: E731
prev = np.zeros(abi.RT_STAT_COUNT)
print(f"{'bounce':>6s} {'rays':>12s} {'node visits/ray':>16s} {'box tests/ray':>14s} {'leaf tests/ray':>15s} {'hit share':>10s}")
for depth in range(1, top + 1):
    _, st = ctx.render(ctx.params(w, h, 0, spp, depth, seed=1), dtype=np.float32, stats=True)
    st = st.astype(np.float64)
    d = st - prev
    prev = st
    rays = g(d, "rays")
    if rays == 0:
        break
    leaf = g(d, "sphere_tests") + g(d, "msphere_tests") + g(d, "rect_tests")
    print(f"{depth - 1:6d} {int(rays):12d} {g(d, 'node_visits') / rays:16.2f} {g(d, 'box_tests') / rays:14.2f} "
          f"{leaf / rays:15.2f} {g(d, 'hits') / rays:10.3f}")
ctx.close()
