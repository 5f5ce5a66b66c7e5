"""Scaled-up book_cover (ground + (2*half)^2 small spheres + 3 big ones): build and render timing.

usage: python tools/cover_scale.py half_extent [spp] [width] [height]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi  # noqa: E402
from raytracer_amd.render import Context  # noqa: E402
from raytracer_amd.scene import HostScene  # noqa: E402

half = int(sys.argv[1])
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
w = int(sys.argv[3]) if len(sys.argv) > 3 else 3840
h = int(sys.argv[4]) if len(sys.argv) > 4 else 2160
t0 = time.perf_counter()
scene = HostScene.generate_cover(half, scene_seed=1, width=w, height=h, spp=spp, max_depth=50)
t1 = time.perf_counter()
ctx = Context(0)
ctx.upload(scene)
t2 = time.perf_counter()
print(f"half_extent {half}: {scene.view.n_objects} objects; generate {t1 - t0:.2f} s, upload+BVH {t2 - t1:.2f} s",
      flush=True)
best = 1e9
for rep in range(2):
    p = ctx.params(w, h, 0, spp, 50, seed=1)
    t = time.perf_counter()
    img, st = ctx.render(p, dtype=np.float32, timing=True)
    best = min(best, time.perf_counter() - t)
print(f"  {w}x{h} {spp} spp: {best * 1e3:.1f} ms, {w * h * spp / best / 1e6:.1f} M pixel-samples/s, "
      f"kernel {st[20] / 1e6:.1f} ms, mean {float(img.mean()) / spp:.4f}", flush=True)
_, st = ctx.render(ctx.params(w, h, 0, 1, 50, seed=1), dtype=np.float32, stats=True)
rays = float(st[0])
print(f"  rays/sample {rays / (w * h):.2f}, box tests/ray {st[1] / rays:.1f}, sphere tests/ray {st[2] / rays:.2f}")
ctx.close()
