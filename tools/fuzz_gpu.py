"""One-off differential run: many seeded random scenes, device against the CPU oracle.

usage: python tools/fuzz_gpu.py first_seed count [textured] [wide]     (wide: lenses and focal planes that strain the tile beams)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402
from random_scenes import random_scene  # noqa: E402
from raytracer_amd import abi  # noqa: E402
from raytracer_amd.render import Context  # noqa: E402
from raytracer_amd.scene import HostScene  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
textured = "textured" in sys.argv[3:]
wide = "wide" in sys.argv[3:]
os.environ.setdefault("MI355RT_STREAM_MIN_PATHS", "0")  # small renders too run the streamed path state
ctx = Context(0)
bad = 0
for seed in range(first, first + count):
    text = random_scene(seed, textured=textured, width=64, height=48, wide_lens=wide)
    scene = HostScene.load_ron_text(text)
    ctx.upload(scene)
    w, h = scene.width, scene.height
    o = po.Oracle(scene.view_ptr)
    ids, t, n = ctx.first_hit(w, h)
    rid, rt_, rn = o.first_hit(w, h)
    ok = np.array_equal(ids, rid) and np.array_equal(t, rt_) and np.array_equal(n, rn)
    ref = o.render(po.make_params(w, h, 0, 2, scene.max_depth, 5), po.MODE_ITERATIVE, threads=8)
    for flags in (0, abi.RT_FLAG_BINNED):
        img = ctx.render(ctx.params(w, h, 0, 2, scene.max_depth, seed=5, flags=flags), dtype=np.float64)
        if textured or "ConstantMedium" in text:
            ok = ok and np.isclose(img, ref, rtol=1e-9, atol=1e-12).mean() > 0.995
        else:
            ok = ok and np.array_equal(img, ref)
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, "objects", scene.view.n_objects, flush=True)
    if (seed - first) % 100 == 99:
        print("...", seed, flush=True)
print("done:", count, "scenes,", bad, "mismatches")
ctx.close()
