"""The first-hit and image gates at the sizes of the named configurations (see tests/gates.py).

book_cover 1920x1080, cornell 1024x1024 (depth 500), book2 1920x1080 (media, Perlin, image
texture, shutter open) and the 1,000,004-sphere cover at 3840x2160.  First hits cover the whole
frame (windows for the 1 M-sphere scene: its reference-topology oracle needs 0.6 ms a ray);
the image gate runs at 4096 spp on three spread crops per scene, small enough for the CPU.
"""
import numpy as np
import pytest

import gates

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["book_cover", "cornell", "book2"])
def test_first_hit_gate_at_full_size(gpu_ctx, name):
    w, h, _ = gates.CONFIGS[name]
    scene = gates.load(name, w, h)
    gpu_ctx.upload(scene)
    ev = gates.first_hit_gate(gpu_ctx, scene, w, h)
    assert ev["pixels"] == w * h and ev["hits"] > 0
    # beyond the stated 1e-5: no libm call on this path, so t and normal match bit for bit
    assert ev["t_bit_exact"] and ev["normal_bit_exact"]


@pytest.mark.parametrize("name,cw,ch", [("book_cover", 16, 12), ("cornell", 16, 12), ("book2", 8, 6)])
def test_image_gate_at_4096_spp(gpu_ctx, name, cw, ch):
    w, h, _ = gates.CONFIGS[name]
    scene = gates.load(name, w, h)
    gpu_ctx.upload(scene)
    ev = gates.image_gate(gpu_ctx, scene, name, w, h, 4096, cw, ch)
    if name != "book2":  # no libm call in these scenes: the seed-A images are the same numbers
        assert ev["max_abs_mean_diff_same_seed"] <= 1e-12


@pytest.mark.parametrize("name,w,h,dev_spp,cpu_spp", [("book_cover", 1920, 1080, 64, 16), ("cornell", 256, 256, 256, 64),
                                                     ("book2", 480, 272, 64, 16)])
def test_whole_frame_statistic(gpu_ctx, name, w, h, dev_spp, cpu_spp):
    """Every pixel of the frame, device seed C against oracle seed B (gates.frame_statistic_gate);
    cornell (a light of radiance 15 in a closed box: heavy-tailed samples) with four times the samples."""
    scene = gates.load(name, w, h)
    gpu_ctx.upload(scene)
    ev = gates.frame_statistic_gate(gpu_ctx, scene, w, h, dev_spp=dev_spp, cpu_spp=cpu_spp)
    print(name, ev)
    assert ev["tile_channels"] > 0.5 * (w // 8) * (h // 8) * 3


def test_million_sphere_cover_at_3840x2160(gpu_ctx):
    w, h = 3840, 2160
    scene = gates.cover_scene(500, w, h)
    assert scene.view.n_objects == 1_000_004
    gpu_ctx.upload(scene)
    # centre, the horizon (grazing rays over thousands of spheres), a lower corner
    windows = [(1888, 1060, 64, 40), (1600, 44, 96, 24), (40, 2090, 64, 40)]
    ev = gates.large_scene_gate(gpu_ctx, scene, w, h, windows, (1900, 1100, 32, 18))
    assert all(e["id_mismatches"] == 0 and e["t_bit_exact"] for e in ev["first_hit"])
