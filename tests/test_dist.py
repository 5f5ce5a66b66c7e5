"""Sample-range sharding and the reduce, on CPU with gloo (world size 2)."""
import os
import sys

import numpy as np
import pytest

from raytracer_amd.dist import gather_rank_times, render_sharded, shard_samples

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shards_tile_the_range():
    for begin, end, world in [(0, 1024, 8), (0, 1024, 3), (5, 17, 4), (0, 3, 8), (0, 0, 2)]:
        parts = [shard_samples(begin, end, r, world) for r in range(world)]
        assert parts[0][0] == begin and parts[-1][1] == end
        assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
        sizes = [hi - lo for lo, hi in parts]
        assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    import pyoracle as po
    from raytracer_amd.scene import HostScene
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = HostScene.load_frozen(os.path.join(ROOT, "tests", "golden", "material_demo.rtscene"))
    w, h, spp = 48, 27, 6
    scene.set_image(w, h)
    oracle = po.Oracle(scene.view_ptr)

    def render_fn(lo, hi, buf):  # stand-in for the device: the CPU oracle, float32 sums
        img = oracle.render(po.make_params(w, h, lo, hi, 50, 9), po.MODE_ITERATIVE)
        buf.copy_(torch.from_numpy(img.astype(np.float32)))

    out = torch.zeros((h, w, 3), dtype=torch.float32)
    lo, hi = render_sharded(render_fn, out, 0, spp, rank, world, dist)
    # the per-rank timing report of the benchmark: every rank gets every rank's figures, in rank order
    times = gather_rank_times(dist, world, 0.010 * (rank + 1), 0.050, 5)
    assert np.allclose(times["render"], [2.0 * (k + 1) for k in range(world)])
    assert np.allclose(times["reduce_and_wait"], [10.0 - 2.0 * (k + 1) for k in range(world)])
    if rank == 0:
        np.save(os.path.join(out_dir, "sharded.npy"), out.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_render_matches_single(tmp_path):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    from raytracer_amd.scene import HostScene
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "sharded.npy")
    scene = HostScene.load_frozen(os.path.join(ROOT, "tests", "golden", "material_demo.rtscene"))
    scene.set_image(48, 27)
    oracle = po.Oracle(scene.view_ptr)
    lo = oracle.render(po.make_params(48, 27, 0, 3, 50, 9), po.MODE_ITERATIVE).astype(np.float32)
    hi = oracle.render(po.make_params(48, 27, 3, 6, 50, 9), po.MODE_ITERATIVE).astype(np.float32)
    assert np.array_equal(got, lo + hi)
    whole = oracle.render(po.make_params(48, 27, 0, 6, 50, 9), po.MODE_ITERATIVE)
    assert np.abs(got - whole).max() <= 1e-5 * np.abs(whole).max()
