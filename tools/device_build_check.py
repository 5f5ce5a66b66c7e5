"""Host-built against device-built hierarchy: upload time, then first hits and radiance must be equal.

usage (GPU box): python tools/device_build_check.py [grid_half ...]     (generate_cover(g): 4 g^2 + 4 spheres)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd.render import Context  # noqa: E402
from raytracer_amd.scene import HostScene  # noqa: E402

grids = [int(a) for a in sys.argv[1:]] or [100, 500]
for g in grids:
    w, h = (3840, 2160) if g >= 500 else (960, 540)
    scene = HostScene.generate_cover(g, scene_seed=1, width=w, height=h, spp=4, max_depth=50)
    out = {}
    for mode in ("0", "1"):
        os.environ["MI355RT_DEVICE_BUILD"] = mode
        ctx = Context(0)
        ctx.upload(scene)  # (the first upload of a process also pays the module load)
        t0 = time.perf_counter()
        ctx.upload(scene)
        up = time.perf_counter() - t0
        ids, t, n = ctx.first_hit(w, h)
        t0 = time.perf_counter()
        img = ctx.render(ctx.params(w, h, 0, 4, 50, seed=1), dtype=np.float64)
        rt = time.perf_counter() - t0
        out[mode] = (ids, t, n, img)
        print(f"grid {g} ({scene.view.n_objects} objects) device_build={mode}: upload {up * 1e3:7.1f} ms, "
              f"render 4 spp {rt * 1e3:7.1f} ms", flush=True)
        ctx.close()
    same = all(np.array_equal(a, b) for a, b in zip(out["0"], out["1"]))
    print(f"grid {g}: first hits and radiance equal: {same}", flush=True)
    if not same:
        sys.exit(1)
