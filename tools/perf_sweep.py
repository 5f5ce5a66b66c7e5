"""Time the render under different settings read from the environment at upload.

usage: python tools/perf_sweep.py scene spp width height  VAR=v1,v2,...  [VAR2=...]
"""
import itertools
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd.render import Context  # noqa: E402
from raytracer_amd.scene import HostScene  # noqa: E402

name, spp, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
sweeps = [a.split("=") for a in sys.argv[5:]]
keys = [k for k, _ in sweeps]
values = [v.split(",") for _, v in sweeps]
if name == "cover1m":  # the 1,000,004-sphere scale-up of book_cover.ron
    scene = HostScene.generate_cover(500, scene_seed=1, width=w, height=h, spp=spp, max_depth=50)
else:
    scene = HostScene.load_frozen(os.path.join(ROOT, "tests", "golden", name + ".rtscene"))
    scene.set_image(w, h)
depth = int(os.environ.get('SWEEP_DEPTH', scene.max_depth))
ref = None
for combo in itertools.product(*values) if values else [()]:
    flags = 0
    timing = True
    for k, v in zip(keys, combo):
        if k == "TIMING":  # 0: no events around the launches (what they cost shows in the wall time)
            timing = v != "0"
        elif k == "FLAGS":  # render flags (4 = force the binned extend/shade pipeline), not an env var
            flags = int(v)
        elif v == "-":
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    ctx = Context(0)
    ctx.upload(scene)
    best = 1e9
    for rep in range(3):
        p = ctx.params(w, h, 0, spp, depth, seed=1, flags=flags)
        t0 = time.perf_counter()
        img, st = ctx.render(p, dtype=np.float32, timing=True) if timing else (ctx.render(p, dtype=np.float32), np.zeros(24))
        best = min(best, time.perf_counter() - t0)
    chk = float(img.astype(np.float64).sum())
    if ref is None:
        ref = chk
    print(dict(zip(keys, combo)), f"{best * 1e3:8.1f} ms  {w * h * spp / best / 1e6:8.1f} Mps/s  "
          f"extend {st[20] / 1e6:7.1f} ms shade {st[22] / 1e6:7.1f} ms  same_image={chk == ref}", flush=True)
    ctx.close()
