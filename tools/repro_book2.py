"""book2 at full size: does whole == lo + hi, and fused == binned, under several grid sizes?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi
from raytracer_amd.render import Context
from raytracer_amd.scene import HostScene
w, h, spp = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 32
scene = HostScene.load_frozen(os.path.join(ROOT, "tests", "golden", "book2.rtscene"))
scene.set_image(w, h)
d = scene.max_depth
ref = None
for per_cu in sys.argv[2:] or ["3", "16"]:
    os.environ["MI355RT_TRACE_BLOCKS_PER_CU"] = per_cu
    ctx = Context(0)
    ctx.upload(scene)
    whole = ctx.render(ctx.params(w, h, 0, spp, d, seed=3), dtype=np.float64)
    again = ctx.render(ctx.params(w, h, 0, spp, d, seed=3), dtype=np.float64)
    if ref is None:
        ref = ctx.render(ctx.params(w, h, 0, spp, d, seed=3, flags=abi.RT_FLAG_BINNED), dtype=np.float64)
    bad = np.any(whole != ref, axis=-1)
    bad2 = np.any(again != ref, axis=-1)
    print(f"per_cu {per_cu}: fused != binned at {int(bad.sum())} pixels (second run {int(bad2.sum())}); "
          f"max |diff| {np.abs(whole - ref).max():.3g}; rows {np.flatnonzero(bad.any(axis=1))[:5]} cols {np.flatnonzero(bad.any(axis=0))[:5]}",
          flush=True)
    ctx.close()
