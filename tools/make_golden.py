"""Generate tests/golden/: frozen scenes + oracle outputs.

Run here (needs /root/reference/scenes and PIL); the outputs are committed so
that GPU-side tests, smoke() and bench.py never read the reference tree.

  *.rtscene      the reference's scene files expanded by this build's loader at
                 scene seed 1 (this build's own binary format, gz)
  *.golden.npz   oracle outputs on those scenes: first-hit buffers and a small
                 window of per-pixel radiance sums (reference-topology,
                 recursive ray_color)
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from raytracer_amd.scene import HostScene, add_asset_dir  # noqa: E402
import pyoracle as po  # noqa: E402

REF = "/root/reference/scenes"
OUT = os.path.join(ROOT, "tests", "golden")

# name -> (first-hit size, render window (x0,y0,w,h) at that size, spp)
SCENES = {
    "material_demo": ((160, 90), (40, 30, 64, 32), 8),
    "book_cover": ((160, 90), (40, 30, 64, 32), 8),
    "cornell": ((96, 96), (16, 16, 48, 48), 8),
    "book2": ((160, 90), (48, 20, 64, 40), 4),
    "medium_demo": ((160, 90), (40, 30, 64, 32), 8),
    "emissive": ((160, 90), (40, 30, 64, 32), 8),
    "checker_mirror": ((160, 120), (40, 40, 64, 32), 8),
    "motion_blur": ((160, 50), (40, 10, 64, 32), 8),
    "globe": ((160, 120), (40, 40, 64, 32), 8),
    "refraction": ((160, 50), (40, 10, 64, 32), 8),
    "perlin": ((160, 120), (40, 40, 64, 32), 8),
}


def main():
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp()
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "decode_image.py"),
                           os.path.join(REF, "earthmap.jpg"), os.path.join(tmp, "earthmap.ppm")])
    add_asset_dir(tmp)
    for name, ((w, h), window, spp) in SCENES.items():
        scene = HostScene.load_ron(os.path.join(REF, name + ".ron"), scene_seed=1)
        scene.save_frozen(os.path.join(OUT, name + ".rtscene"))
        scene.set_image(w, h)
        o = po.Oracle(scene.view_ptr)
        ids, t, n = o.first_hit(w, h)
        p = po.make_params(w, h, 0, spp, scene.max_depth, seed=7, window=window)
        img = o.render(p, po.MODE_REFERENCE, threads=8)
        np.savez_compressed(os.path.join(OUT, name + ".golden.npz"), width=w, height=h,
                            first_id=ids, first_t=t, first_normal=n, window=np.array(window),
                            spp=spp, seed=7, max_depth=scene.max_depth, render_sum=img)
        size = os.path.getsize(os.path.join(OUT, name + ".rtscene"))
        print(f"{name}: {scene.view.n_objects} objects, frozen {size} bytes, "
              f"hit fraction {(ids != 0xFFFFFFFF).mean():.3f}")


if __name__ == "__main__":
    main()
