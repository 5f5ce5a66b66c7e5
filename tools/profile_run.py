"""Small fixed workload for rocprofv3: book_cover 1920x1080 at a few spp through the C ABI."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd.render import Context  # noqa: E402
from raytracer_amd.scene import HostScene  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "book_cover"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
w = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
h = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
scene = HostScene.load_frozen(os.path.join(ROOT, "tests", "golden", name + ".rtscene"))
scene.set_image(w, h)
ctx = Context(0)
ctx.upload(scene)
depth = int(os.environ.get('SWEEP_DEPTH', scene.max_depth))
img = ctx.render(ctx.params(w, h, 0, spp, depth, seed=1), dtype=np.float32)
print(name, w, h, spp, "mean", float(img.mean()) / spp)
ctx.close()
