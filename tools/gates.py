"""Run the first-hit and image gates at the named configurations' sizes on the GPU box and write
the evidence as JSON (larger crops than the test-suite's).

usage: python tools/gates.py out.json [crop_w crop_h]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import gates  # noqa: E402
from raytracer_amd.render import Context  # noqa: E402

out_path = sys.argv[1]
cw = int(sys.argv[2]) if len(sys.argv) > 2 else 24
ch = int(sys.argv[3]) if len(sys.argv) > 3 else 16
ctx = Context(0)
result = {"first_hit": {}, "image_4096spp": {}, "oracle": "oracle/oracle.cpp, reference BVH topology, recursive ray_color"}


def save():
    with open(out_path, "w") as f:
        json.dump(result, f, indent=1)


for name, (w, h, _) in gates.CONFIGS.items():
    scene = gates.load(name, w, h)
    ctx.upload(scene)
    t0 = time.perf_counter()
    result["first_hit"][name] = gates.first_hit_gate(ctx, scene, w, h)
    print(f"first hit {name} {w}x{h}: ok in {time.perf_counter() - t0:.1f} s", flush=True)
    save()
for name, (w, h, _) in gates.CONFIGS.items():
    scene = gates.load(name, w, h)
    ctx.upload(scene)
    t0 = time.perf_counter()
    c = (cw // 2, ch // 2) if name == "book2" else (cw, ch)  # book2's oracle is ten times slower
    result["image_4096spp"][name] = gates.image_gate(ctx, scene, name, w, h, 4096, *c)
    print(f"image gate {name}: ok in {time.perf_counter() - t0:.1f} s", flush=True)
    save()
w, h = 3840, 2160
scene = gates.cover_scene(500, w, h)
t0 = time.perf_counter()
ctx.upload(scene)
up = time.perf_counter() - t0
windows = [(1888, 1060, 64, 40), (1600, 44, 96, 24), (40, 2090, 64, 40), (3700, 1000, 64, 40)]
result["cover_1M_3840x2160"] = gates.large_scene_gate(ctx, scene, w, h, windows, (1900, 1100, 32, 18))
result["cover_1M_3840x2160"]["upload_seconds"] = up
print("1 M-sphere cover: ok", flush=True)
save()
ctx.close()
