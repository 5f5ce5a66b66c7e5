"""Device groups behind the C ABI, the sharded render on a device tensor, and the CLI's extra outputs.

A single-GPU machine rehearses a group by naming its device twice: two contexts, two host
threads, two sample ranges, and the shares added through host memory (a communicator needs
distinct devices; that path runs wherever a node has more than one GPU).
"""
import os
import subprocess

import numpy as np
import pytest

from raytracer_amd import abi
from raytracer_amd.render import Context
from raytracer_amd.scene import HostScene

pytestmark = pytest.mark.gpu


def _scene(golden_dir, name, w, h):
    scene = HostScene.load_frozen(os.path.join(golden_dir, name + ".rtscene"))
    scene.set_image(w, h)
    return scene


@pytest.mark.parametrize("name", ["book_cover", "book2"])
def test_group_of_two_shares_equals_one_context(gpu_ctx, golden_dir, name):
    w, h, spp = 320, 180, 9
    scene = _scene(golden_dir, name, w, h)
    gpu_ctx.upload(scene)
    depth = scene.max_depth
    whole, st = gpu_ctx.render(gpu_ctx.params(w, h, 0, spp, depth, seed=6), dtype=np.float32, stats=True)
    lo = gpu_ctx.render(gpu_ctx.params(w, h, 0, 4, depth, seed=6), dtype=np.float32)   # 9 * 1 // 2
    hi = gpu_ctx.render(gpu_ctx.params(w, h, 4, spp, depth, seed=6), dtype=np.float32)
    group = Context(devices=[0, 0])
    try:
        group.upload(scene)
        got, gst = group.render(group.params(w, h, 0, spp, depth, seed=6), dtype=np.float32, stats=True)
        assert np.array_equal(got, lo + hi)  # the float shares, added in member order
        assert np.abs(got - whole).max() <= 2e-6 * np.abs(whole).max()
        assert list(gst[:20]) == list(st[:20])  # the same rays, whoever traced them
        import torch
        out = torch.full((h, w, 3), -1.0, dtype=torch.float32, device="cuda:0")
        group.render_into(group.params(w, h, 0, spp, depth, seed=6), out.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), got)
        ids = group.first_hit(w, h)[0]
        assert np.array_equal(ids, gpu_ctx.first_hit(w, h)[0])
        with pytest.raises(abi.RtError):
            group.render(group.params(w, h, 0, spp, depth, seed=6), dtype=np.float64)
    finally:
        group.close()


def test_group_of_one_device_is_the_plain_context(gpu_ctx, golden_dir):
    w, h = 160, 90
    scene = _scene(golden_dir, "material_demo", w, h)
    gpu_ctx.upload(scene)
    want = gpu_ctx.render(gpu_ctx.params(w, h, 0, 3, 50, seed=2), dtype=np.float64)
    one = Context(devices=[0])
    try:
        one.upload(scene)
        assert np.array_equal(one.render(one.params(w, h, 0, 3, 50, seed=2), dtype=np.float64), want)
    finally:
        one.close()


def test_group_of_one_runs_the_rccl_reduce(gpu_ctx, golden_dir, monkeypatch):
    """MI355RT_FORCE_RCCL=1: a group of one member with a one-rank communicator.  Everything the
    multi-GPU reduce consists of runs on a single GPU: the run-time lookup of librccl and its six
    entry points, ncclCommInitAll, ncclGroupStart / ncclReduce(float32, sum) on the member's
    stream / ncclGroupEnd, the waits, ncclCommDestroy.  A one-rank sum is the share itself."""
    import torch
    w, h, spp = 320, 180, 5
    scene = _scene(golden_dir, "book_cover", w, h)
    gpu_ctx.upload(scene)
    assert gpu_ctx.group_info()[:2] == (1, "none")
    want, st = gpu_ctx.render(gpu_ctx.params(w, h, 0, spp, scene.max_depth, seed=4), dtype=np.float32, stats=True)
    monkeypatch.setenv("MI355RT_FORCE_RCCL", "1")
    one = Context(devices=[0])
    try:
        assert one.group_info()[:2] == (1, "rccl")
        one.upload(scene)
        for _ in range(2):  # the communicator serves call after call
            got, gst = one.render(one.params(w, h, 0, spp, scene.max_depth, seed=4), dtype=np.float32, stats=True)
            assert np.array_equal(got, want) and list(gst[:20]) == list(st[:20])
        out = torch.full((h, w, 3), -1.0, dtype=torch.float32, device="cuda:0")
        one.render_into(one.params(w, h, 0, spp, scene.max_depth, seed=4), out.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want)
        n, path, ms = one.group_info()
        assert ms["render_ms_max"] >= ms["render_ms_min"] > 0.0 and ms["reduce_ms"] > 0.0
        with pytest.raises(abi.RtError):  # a group adds float shares
            one.render(one.params(w, h, 0, spp, scene.max_depth, seed=4), dtype=np.float64)
    finally:
        one.close()


def test_group_says_how_it_adds_its_shares(golden_dir, monkeypatch):
    """Two shares on one device can have no communicator (host-staged, and says so); distinct
    devices use RCCL, or fail loudly without it unless the staged path is asked for."""
    import torch
    rehearsal = Context(devices=[0, 0])
    try:
        assert rehearsal.group_info()[:2] == (2, "host-staged")
    finally:
        rehearsal.close()
    if torch.cuda.device_count() < 2:
        return
    w, h, spp = 160, 90, 6
    scene = _scene(golden_dir, "book_cover", w, h)
    single = Context(device=0)
    try:
        single.upload(scene)
        lo = single.render(single.params(w, h, 0, 3, 50, seed=6), dtype=np.float32)
        hi = single.render(single.params(w, h, 3, spp, 50, seed=6), dtype=np.float32)
    finally:
        single.close()
    for staged in ("0", "1"):
        monkeypatch.setenv("MI355RT_STAGED_REDUCE", staged)
        pair = Context(devices=[0, 1])
        try:
            assert pair.group_info()[:2] == (2, "host-staged" if staged == "1" else "rccl")
            pair.upload(scene)
            assert np.array_equal(pair.render(pair.params(w, h, 0, spp, 50, seed=6), dtype=np.float32), lo + hi)
        finally:
            pair.close()


def test_sharded_render_into_a_device_tensor(gpu_ctx, golden_dir):
    """render_sharded with the device render (one rank): the tensor holds the whole range's sums."""
    import torch
    from raytracer_amd.dist import render_sharded
    w, h = 200, 120
    scene = _scene(golden_dir, "cornell", w, h)
    gpu_ctx.upload(scene)
    out = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")

    def render_fn(lo, hi, buf):
        gpu_ctx.render_into(gpu_ctx.params(w, h, lo, hi, 20, seed=8), buf.data_ptr())

    for _ in range(2):  # twice in a row on the same tensor
        out.fill_(7.0)
        assert render_sharded(render_fn, out, 0, 5, 0, 1, None) == (0, 5)
    want = gpu_ctx.render(gpu_ctx.params(w, h, 0, 5, 20, seed=8), dtype=np.float32)
    assert np.array_equal(out.cpu().numpy(), want)


def test_cli_linear_dump_and_first_hit(gpu_ctx, golden_dir, tmp_path):
    exe = os.path.join(os.path.dirname(abi.device_lib()._name), "raytracer")
    frozen = os.path.join(golden_dir, "material_demo.rtscene")
    pfm, rtfh, png = tmp_path / "lin.pfm", tmp_path / "hits.rtfh", tmp_path / "image.png"
    r = subprocess.run([exe, frozen, "--width", "96", "--height", "54", "--spp", "4", "--seed", "3", "--devices", "0,0",
                        "--out", str(png), "--dump-pfm", str(pfm), "--first-hit", str(rtfh)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    w, h = 96, 54
    scene = _scene(golden_dir, "material_demo", w, h)
    gpu_ctx.upload(scene)
    raw = pfm.read_bytes()
    head = b"PF\n96 54\n-1.0\n"
    assert raw.startswith(head)
    lin = np.frombuffer(raw[len(head):], dtype="<f4").reshape(h, w, 3)[::-1]
    want = gpu_ctx.render(gpu_ctx.params(w, h, 0, 4, scene.max_depth, seed=3), dtype=np.float64) / 4
    assert np.abs(lin - want).max() <= 1e-6 * np.abs(want).max()
    raw = rtfh.read_bytes()
    assert raw[:4] == b"RTFH" and tuple(np.frombuffer(raw[4:12], dtype="<u4")) == (w, h)
    n = w * h
    ids = np.frombuffer(raw[12:12 + 4 * n], dtype="<u4").reshape(h, w)
    t = np.frombuffer(raw[12 + 4 * n:12 + 12 * n], dtype="<f8").reshape(h, w)
    rid, rt_, _ = gpu_ctx.first_hit(w, h)
    assert np.array_equal(ids, rid) and np.array_equal(t, rt_)


def test_progress_is_reported_after_every_pass(gpu_ctx, golden_dir):
    """rt_set_progress (the reference's progress bar, main.rs:58-61,99): done grows pass by pass to total,
    the image is the one rendered without a callback, and removing the callback stops the reports."""
    from raytracer_amd.render import Context
    scene = HostScene.load_frozen(os.path.join(golden_dir, "book_cover.rtscene"))
    w, h = scene.width, scene.height
    gpu_ctx.upload(scene)
    plain = gpu_ctx.render(gpu_ctx.params(w, h, 0, 7, 50, seed=3), dtype=np.float64)
    small = Context(device=0, max_paths_in_flight=2 * w * h + 5)  # 2 samples at a time: 4 passes of 2, 2, 2, 1
    try:
        small.upload(scene)
        seen = []
        small.set_progress(lambda done, total: seen.append((done, total)))
        image = small.render(small.params(w, h, 0, 7, 50, seed=3), dtype=np.float64)
        assert np.array_equal(image, plain)
        assert seen == [(k * w * h, 7 * w * h) for k in (2, 4, 6, 7)]
        del seen[:]
        small.render(small.params(w, h, 2, 5, 50, seed=3, window=(8, 4, 16, 8)), dtype=np.float64)
        assert seen == [(3 * 16 * 8, 3 * 16 * 8)]  # one pass; each call counts from zero, in its own window
        del seen[:]
        small.set_progress(None)
        small.render(small.params(w, h, 0, 3, 50, seed=3), dtype=np.float64)
        assert seen == []
    finally:
        small.close()


def test_progress_of_a_device_group_counts_every_member(gpu_ctx, golden_dir):
    from raytracer_amd.render import Context
    scene = HostScene.load_frozen(os.path.join(golden_dir, "cornell.rtscene"))
    w, h = scene.width, scene.height
    group = Context(devices=[0, 0], max_paths_in_flight=2 * w * h)
    try:
        group.upload(scene)
        seen = []
        group.set_progress(lambda done, total: seen.append((done, total)))
        group.render(group.params(w, h, 0, 6, 20, seed=9))
        # two members of three samples each, two passes each (2 + 1 samples): four reports, in any order
        assert len(seen) == 4 and all(t == 6 * w * h for _, t in seen)
        done = [d for d, _ in seen]
        assert done == sorted(done) and done[-1] == 6 * w * h
    finally:
        group.close()
