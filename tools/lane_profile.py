"""Loop-trip and lane-utilisation counts of the traversal/shade loops (development build).

usage (GPU box): make laneprof  (here), then
    MI355RT_LIB_DIR=build/laneprof python tools/lane_profile.py book_cover 4 960 540
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import abi  # noqa: E402
from raytracer_amd.render import Context  # noqa: E402
from raytracer_amd.scene import HostScene  # noqa: E402

LOOPS = ["ray", "outer", "node", "leafprim", "pop", "shade", "reject", "roots"]
PHASES = ["setup", "traverse", "shade", "vector", "store"]
name = sys.argv[1] if len(sys.argv) > 1 else "book_cover"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
w = int(sys.argv[3]) if len(sys.argv) > 3 else 960
h = int(sys.argv[4]) if len(sys.argv) > 4 else 540
flags = int(sys.argv[5]) if len(sys.argv) > 5 else 0
scene = HostScene.load_frozen(os.path.join(ROOT, "tests", "golden", name + ".rtscene"))
scene.set_image(w, h)
ctx = Context(0)
ctx.upload(scene)
lib = abi.device_lib()
out = (ctypes.c_ulonglong * 64)()
lib.rt_debug_lane_trips(out, 64)
depth = int(os.environ.get("LP_DEPTH", scene.max_depth))
img, st = ctx.render(ctx.params(w, h, 0, spp, depth, seed=1, flags=flags), dtype=np.float32, stats=True)
n = lib.rt_debug_lane_trips(out, 64)
rays = int(st[0])
print(f"{name} {w}x{h} spp {spp}: rays {rays}")
print(f"{'loop':10s} {'lane trips':>14s} {'wave trips':>12s} {'lanes/trip':>10s} {'lane trips/ray':>14s} {'wave trips/(rays/64)':>20s}")
for k, nm in enumerate(LOOPS):
    lt, wt = out[2 * k], out[2 * k + 1]
    if wt:
        print(f"{nm:10s} {lt:14d} {wt:12d} {lt / wt:10.1f} {lt / rays:14.2f} {wt / (rays / 64):20.2f}")
total = sum(out[2 * (len(LOOPS) + k)] for k in range(len(PHASES)))
if total:
    print("phases of a trace_kernel trip (clock ticks summed over waves; queue staging and flushes not included):")
    for k, nm in enumerate(PHASES):
        t, c = out[2 * (len(LOOPS) + k)], out[2 * (len(LOOPS) + k) + 1]
        print(f"  {nm:10s} {100.0 * t / total:5.1f} %   {t / max(c, 1):9.0f} ticks per trip")
ctx.close()
