"""book2 at full size, fused against binned, by depth and sample count (one grid size)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi
from raytracer_amd.render import Context
from raytracer_amd.scene import HostScene
w, h = 1920, 1080
scene = HostScene.load_frozen(os.path.join(ROOT, "tests", "golden", "book2.rtscene"))
scene.set_image(w, h)
ctx = Context(0)
ctx.upload(scene)
for spp in (1, 4, 32):
    for d in (1, 2, 3, 32):
        a = ctx.render(ctx.params(w, h, 0, spp, d, seed=3), dtype=np.float64); sa = np.zeros(24)
        b, sb = ctx.render(ctx.params(w, h, 0, spp, d, seed=3, flags=abi.RT_FLAG_BINNED), dtype=np.float64, stats=True)
        bad = np.any(a != b, axis=-1)
        ys, xs = np.nonzero(bad)
        print(f"spp {spp} depth {d}: {int(bad.sum())} pixels differ; rays {sa[0]} vs {sb[0]}, hits {sa[8]} vs {sb[8]}, "
              f"isotropic {sa[14]} vs {sb[14]}, perlin {sa[16]} vs {sb[16]}, image {sa[17]} vs {sb[17]}; "
              f"y range {ys.min() if len(ys) else '-'}..{ys.max() if len(ys) else '-'}", flush=True)
ctx.close()
