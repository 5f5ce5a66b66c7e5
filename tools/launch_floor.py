"""What one launch of the fused kernel costs with next to no rays: a small window, one sample, every bounce
a launch of its own (MI355RT_TAIL_PATHS=0 keeps the tail variant out).

usage: python tools/launch_floor.py [scene] [window_w window_h]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MI355RT_TAIL_PATHS", "0")
from raytracer_amd.render import Context  # noqa: E402
from raytracer_amd.scene import HostScene  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "book_cover"
ww = int(sys.argv[2]) if len(sys.argv) > 2 else 64
wh = int(sys.argv[3]) if len(sys.argv) > 3 else 64
scene = HostScene.load_frozen(os.path.join(ROOT, "tests", "golden", name + ".rtscene"))
scene.set_image(1920, 1080)
ctx = Context(0)
ctx.upload(scene)
for spp in (1, 4, 16, 64):
    p = ctx.params(1920, 1080, 0, spp, 50, seed=1, window=(900, 500, ww, wh))
    ctx.render(p)
    _, st = ctx.render(ctx.params(1920, 1080, 0, spp, 50, seed=1, window=(900, 500, ww, wh)), timing=True)
    n = int(st[21])
    print(f"{name} window {ww}x{wh} spp {spp}: {ww * wh * spp} paths, {n} launches, "
          f"{st[20] / 1e3 / max(n, 1):.1f} us per launch")
