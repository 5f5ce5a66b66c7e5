"""One-off: the first-hit gate at 1920x1080 on every golden scene (the suite runs it at full size on
the three named configurations and at 160x120 on the rest).

usage (GPU box): python tools/first_hit_all.py out.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import gates  # noqa: E402
from raytracer_amd.render import Context  # noqa: E402

SCENES = ["material_demo", "book_cover", "cornell", "book2", "medium_demo", "emissive", "checker_mirror",
          "motion_blur", "globe", "refraction", "perlin"]
ctx = Context(0)
out = {}
for name in SCENES:
    w, h = (1024, 1024) if name == "cornell" else (1920, 1080)
    scene = gates.load(name, w, h)
    ctx.upload(scene)
    out[name] = gates.first_hit_gate(ctx, scene, w, h)
    print(name, out[name]["pixels"], "pixels,", out[name]["id_mismatches"], "id mismatches, t bit-exact",
          out[name]["t_bit_exact"], flush=True)
    with open(sys.argv[1], "w") as f:
        json.dump(out, f, indent=1)
ctx.close()
