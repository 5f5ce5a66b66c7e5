"""Small fixed workload for rocprofv3: one named configuration at a few spp through the C ABI.

usage: python tools/profile_run.py scene spp [width height]     scene: a golden scene, or cover1m
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd.render import Context  # noqa: E402
from raytracer_amd.scene import HostScene  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "book_cover"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
w = int(sys.argv[3]) if len(sys.argv) > 3 else (3840 if name == "cover1m" else 1920)
h = int(sys.argv[4]) if len(sys.argv) > 4 else (2160 if name == "cover1m" else 1080)
if name == "cover1m":  # the 1,000,004-sphere book_cover-style scene
    scene = HostScene.generate_cover(500, scene_seed=1, width=w, height=h, spp=spp, max_depth=50)
else:
    scene = HostScene.load_frozen(os.path.join(ROOT, "tests", "golden", name + ".rtscene"))
    scene.set_image(w, h)
ctx = Context(0)
t0 = time.perf_counter()
ctx.upload(scene)
up = time.perf_counter() - t0
depth = int(os.environ.get('SWEEP_DEPTH', scene.max_depth))
img = ctx.render(ctx.params(w, h, 0, spp, depth, seed=1), dtype=np.float32)
print(name, w, h, spp, "mean", float(img.mean()) / spp, f"upload {up * 1e3:.0f} ms")
ctx.close()
