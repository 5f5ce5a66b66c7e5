"""Sample-range sharding over ranks (one process per GPU) and the one collective.

Every pixel-sample is independent (src/main.rs:87-97); the only coupling is the
per-pixel sum (main.rs:97).  Rank r of W renders samples
[begin + r*n/W, begin + (r+1)*n/W) of every pixel into its own float32 buffer;
one reduce(sum) to rank 0 adds the buffers.  Draws are addressed by
(pixel, sample), so the union of samples is the same for every W.
"""


def shard_samples(sample_begin, sample_end, rank, world):
    """The half-open sample range of `rank`; ranges tile [begin, end) exactly."""
    n = sample_end - sample_begin
    lo = sample_begin + (n * rank) // world
    hi = sample_begin + (n * (rank + 1)) // world
    return lo, hi


def render_sharded(render_fn, out, sample_begin, sample_end, rank, world, dist=None):
    """Render this rank's share into `out` and reduce onto rank 0.

    render_fn(lo, hi, out): fills `out` (a tensor on this rank's device) with the
    float32 sums over samples [lo, hi).  `dist` is torch.distributed (initialised)
    or None for a single process.
    """
    lo, hi = shard_samples(sample_begin, sample_end, rank, world)
    # render_fn writes `out` on the library's own stream and returns when that is done; what
    # torch still has queued on `out` (a fill, the previous step's reduce) must be over before
    # it starts, and the reduce must be over before the caller reads or the next step overwrites
    def settle():
        if out.is_cuda:
            import torch
            torch.cuda.current_stream(out.device).synchronize()

    settle()
    render_fn(lo, hi, out)
    if dist is not None and world > 1:
        work = dist.reduce(out, dst=0, op=dist.ReduceOp.SUM, async_op=True)
        work.wait()
        settle()
    return lo, hi


def gather_rank_times(dist, world, render_s, step_s, steps, device=None):
    """What every rank spent per step inside the library (its share of the samples) and, the rest
    of its step, in the reduce and the waits around it, in milliseconds: rank-ordered lists
    {"render": [...], "reduce_and_wait": [...]} on every rank.  A scaling run then explains itself:
    a slow rank shows in `render`, a slow collective or an uneven split in `reduce_and_wait`."""
    import torch
    mine = torch.tensor([render_s, step_s - render_s], dtype=torch.float64, device=device)
    if dist is None or world <= 1:
        gathered = [mine]
    else:
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
    per = 1e3 / max(steps, 1)
    return {"render": [per * float(g[0]) for g in gathered],
            "reduce_and_wait": [per * float(g[1]) for g in gathered]}
