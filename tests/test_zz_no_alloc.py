"""rt_render_device_noalloc (the entry point that allocates nothing) against rt_render_device.

Two calls in a row through the C ABI on one context, the second with a smaller input: the
scratch array keeps the first call's contents, so anything the second call fails to clear shows.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(ctx, torch, w, h, window, s0, s1, depth, seed):
    hh, ww = (h, w) if window is None else (window[3], window[2])
    a = torch.full((hh, ww, 3), -1.0, dtype=torch.float32, device="cuda:0")
    b = torch.full((hh, ww, 3), -2.0, dtype=torch.float32, device="cuda:0")
    ctx.render_into(ctx.params(w, h, s0, s1, depth, seed=seed, window=window), a.data_ptr())
    torch.cuda.synchronize()
    # the entry point must leave the device's free memory where it found it: an allocation of a
    # megabyte or more (path buffers are hundreds) would show here
    free_before, _ = torch.cuda.mem_get_info(0)
    ctx.render_into_noalloc(ctx.params(w, h, s0, s1, depth, seed=seed, window=window), b.data_ptr())
    free_after, _ = torch.cuda.mem_get_info(0)
    assert free_after == free_before, f"free device memory moved by {free_before - free_after} bytes"
    torch.cuda.synchronize()
    return a.cpu().numpy(), b.cpu().numpy()


@pytest.mark.parametrize("name", ["book_cover", "book2", "cornell"])
def test_noalloc_equals_render_device(gpu_ctx, golden_dir, name):
    import torch
    from raytracer_amd import abi
    from raytracer_amd.scene import HostScene

    scene = HostScene.load_frozen(os.path.join(golden_dir, name + ".rtscene"))
    w, h = 320, 180
    scene.set_image(w, h)
    gpu_ctx.upload(scene)
    depth = min(scene.max_depth, 50)
    # first call: the whole frame, several passes through the scratch array
    a, b = _pair(gpu_ctx, torch, w, h, None, 0, 12, depth, 5)
    assert np.array_equal(a, b)
    assert a.max() > 0
    # second call: a smaller window and fewer samples, right after
    a, b = _pair(gpu_ctx, torch, w, h, (40, 30, 96, 50), 3, 7, depth, 9)
    assert np.array_equal(a, b)
    # and the limit: a window whose one-sample pass cannot fit is refused, not truncated
    big = HostScene.load_frozen(os.path.join(golden_dir, name + ".rtscene"))
    big.set_image(1920, 1080)
    gpu_ctx.upload(big)
    out = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda:0")
    with pytest.raises(abi.RtError) as e:
        gpu_ctx.render_into_noalloc(gpu_ctx.params(1920, 1080, 0, 1, depth), out.data_ptr())
    assert e.value.code == -7 and "scratch" in e.value.message
