"""Algorithmic work, kernel time and roofline fractions of one configuration (one pass's worth).

usage: python tools/config_roofline.py scene spp width height out.json
Counts come from the instrumented kernels (RT_FLAG_STATS), times from events around every launch
(RT_FLAG_TIMING) of the uninstrumented ones; costs per operation are bench.py's.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from raytracer_amd import abi  # noqa: E402
from raytracer_amd.render import Context  # noqa: E402
from raytracer_amd.scene import HostScene  # noqa: E402

name, spp, w, h, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
if name == "cover1m":
    scene = HostScene.generate_cover(500, scene_seed=1, width=w, height=h, spp=spp, max_depth=50)
else:
    scene = HostScene.load_frozen(os.path.join(ROOT, "tests", "golden", name + ".rtscene"))
    scene.set_image(w, h)
ctx = Context(0)
t0 = time.perf_counter()
ctx.upload(scene)
upload_s = time.perf_counter() - t0
depth = scene.max_depth
best, st_t = 1e9, None
for _ in range(3):
    t0 = time.perf_counter()
    _, t = ctx.render(ctx.params(w, h, 0, spp, depth, seed=1), dtype=np.float32, timing=True)
    dt = time.perf_counter() - t0
    if dt < best:
        best, st_t = dt, t
_, st = ctx.render(ctx.params(w, h, 0, spp, depth, seed=1), dtype=np.float32, stats=True)
g = lambda n: float(st[abi.STAT_NAMES.index(n)])  # noqa: E731
trace_s = float(st_t[abi.STAT_NAMES.index("extend_ns")]) * 1e-9
shade_s = float(st_t[abi.STAT_NAMES.index("shade_ns")]) * 1e-9
flops = bench.extend_flops(st) + bench.shade_flops(st)
box = g("box_tests") * bench.F_BOX
kernel_s = trace_s + shade_s
ideal = box / (bench.PEAK_F32_TFLOPS * 1e12) + (flops - box) / (bench.PEAK_F64_TFLOPS * 1e12)
res = {
    "config": f"{name} {w}x{h}, {spp} spp in one call, max_depth {depth}",
    "objects": int(scene.view.n_objects), "upload_ms": upload_s * 1e3,
    "wall_ms": best * 1e3, "trace_kernel_ms": trace_s * 1e3, "deferred_shade_kernel_ms": shade_s * 1e3,
    "launches": int(st_t[abi.STAT_NAMES.index("extend_launches")]),
    "pixel_samples_per_s": w * h * spp / best,
    "rays": g("rays"), "rays_per_pixel_sample": g("rays") / (w * h * spp), "grays_per_s_kernel": g("rays") / kernel_s / 1e9,
    "grays_per_s_wall": g("rays") / best / 1e9,
    "box_tests_per_ray": g("box_tests") / g("rays"), "node_visits_per_ray": g("node_visits") / g("rays"),
    "sphere_tests_per_ray": g("sphere_tests") / g("rays"), "rect_tests_per_ray": g("rect_tests") / g("rays"),
    "medium_tests_per_ray": g("medium_tests") / g("rays"),
    "flops_per_pixel_sample": flops / (w * h * spp),
    "achieved_tflops": flops / kernel_s / 1e12, "frac_f64_peak": flops / kernel_s / 1e12 / bench.PEAK_F64_TFLOPS,
    "frac_mixed_peak": ideal / kernel_s,
}
with open(out, "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps(res))
ctx.close()
