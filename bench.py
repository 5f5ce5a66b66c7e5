#!/usr/bin/env python3
"""Headline benchmark: pixel-samples/s on book_cover at 1920x1080, 1024 spp.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

One step = one whole frame: every rank renders its share of the 1024 samples of
every pixel on its GPU (sample-range sharding), then one reduce(sum) over RCCL
puts the float32 radiance sums on rank 0.  Total work per step is fixed, so the
scaling is strong.  Inputs (the scene) are resident on the device before the
timed region; the timed region ends when the reduced buffer is complete on
rank 0's device.

Rank 0 prints one JSON line.  At N=1 it also carries
  roofline     the dominant kernel against the MI355X f64 vector peak: algorithmic
               flops of its launches in the timed steps / their summed device time
               (HIP events recorded around every launch on the library's stream).
               book_cover has only SolidColor textures, so the library runs its fused
               trace kernel (traversal + primitive tests + hit record + scatter in one
               launch per bounce) and the flops are those of all four parts
  cpu_baseline the CPU restatement of the reference (reference BVH topology,
               recursive ray_color) on the host cores, on a bounded sample
  parity       RMSE of the GPU frame against that CPU renderer on a crop, next to
               the CPU renderer's own two-seed RMSE at equal spp
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from raytracer_amd import abi  # noqa: E402
from raytracer_amd.dist import render_sharded  # noqa: E402
from raytracer_amd.render import Context  # noqa: E402
from raytracer_amd.scene import HostScene  # noqa: E402

SCENE = os.path.join(ROOT, "tests", "golden", "book_cover.rtscene")
WIDTH, HEIGHT, SPP, SEED = 1920, 1080, 1024, 1

# Algorithmic cost per operation, in flops (add/sub/mul/div/sqrt/min/max/compare = 1,
# no FMA credit), from the reference's source lines:
F_BOX = 24          # aabb.rs:36-47 with 1/d hoisted: per axis 2 sub, 2 mul, 2 min/max, select, cmp
F_SPHERE = 24       # sphere.rs:32-38 up to the discriminant test
F_SPHERE_ROOT = 8   # sphere.rs:42-50 sqrt, neg, sub, div, compares
F_MSPHERE_EXTRA = 14  # moving_sphere.rs:21-22
F_RECT = 10         # rect.rs:45-50
F_TRANSLATE = 6     # instance.rs:13
F_ROTATE = 12       # instance.rs:55-59
F_MEDIUM_EXTRA = 12  # constant_medium.rs:37-62 besides its two boundary hits
F_HIT_RECORD = 22    # p, outward normal, face (sphere.rs:52-63)
F_SHADE = {"shade_lambertian": 40, "shade_metal": 55, "shade_dielectric": 80, "shade_light": 6,
           "shade_isotropic": 30, "shade_miss": 6}
F_TEX = {"tex_checker": 15, "tex_image": 8, "tex_perlin": 900}
PEAK_F64_TFLOPS = 78.6  # MI355X vector FP64 (spec)
PEAK_F32_TFLOPS = 157.3  # MI355X vector FP32 (spec): the slab tests run in f32


def extend_flops(st):
    """Algorithmic flops of the traversal + primitive tests counted in a stats array."""
    g = lambda name: float(st[abi.STAT_NAMES.index(name)])  # noqa: E731
    return (g("box_tests") * F_BOX + g("sphere_tests") * F_SPHERE + g("sphere_roots") * F_SPHERE_ROOT
            + g("msphere_tests") * (F_SPHERE + F_MSPHERE_EXTRA) + g("rect_tests") * F_RECT
            + g("op_translate") * F_TRANSLATE + g("op_rotate") * F_ROTATE
            + g("medium_tests") * F_MEDIUM_EXTRA)


def shade_flops(st):
    """Algorithmic flops of hit records, scatter/emit and textures counted in a stats array."""
    names = abi.STAT_NAMES
    g = lambda name: float(st[names.index(name)]) if name in names else 0.0  # noqa: E731
    total = g("hits") * F_HIT_RECORD
    for k, f in list(F_SHADE.items()) + list(F_TEX.items()):
        total += g(k) * f
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--spp", type=int, default=SPP)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline and parity legs")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)

    scene = HostScene.load_frozen(SCENE)
    scene.set_image(args.width, args.height)
    depth = scene.max_depth
    ctx = Context(device=local_rank)
    ctx.upload(scene)
    out = torch.zeros((args.height, args.width, 3), dtype=torch.float32, device=device)

    timing_totals = np.zeros(abi.RT_STAT_COUNT, dtype=np.float64)

    def render_fn(lo, hi, buf, timing=False):
        p = ctx.params(args.width, args.height, lo, hi, depth, seed=SEED)
        st = ctx.render_into(p, buf.data_ptr(), timing=timing)
        if timing:
            timing_totals[:] += st.astype(np.float64)

    def step(timing):
        render_sharded(lambda lo, hi, buf: render_fn(lo, hi, buf, timing), out, 0, args.spp, rank,
                       world, dist if use_dist else None)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    pixel_samples = float(args.width) * args.height * args.spp * args.steps
    result = {
        "metric": "pixel-samples/s",
        "value": pixel_samples / elapsed,
        "unit": "pixel-samples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / max(args.steps, 1),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic: book_cover.ron's random sphere pattern expanded by the seeded loader "
                "(scene seed 1, 488 spheres), frozen under tests/golden",
        "config": {
            "workload": f"book_cover {args.width}x{args.height}, {args.spp} spp, max_depth {depth}",
            "scene": "tests/golden/book_cover.rtscene",
            "width": args.width, "height": args.height, "spp": args.spp, "max_depth": depth,
            "sampling_seed": SEED,
            "parallelism": f"sample-range x{world} + one reduce(sum) to rank 0",
        },
    }

    if world == 1 and rank == 0:
        # operation counts of exactly the work of one step, from the instrumented kernels
        p = ctx.params(args.width, args.height, 0, args.spp, depth, seed=SEED)
        _, st = ctx.render(p, dtype=np.float32, stats=True)
        ext_s = timing_totals[abi.STAT_NAMES.index("extend_ns")] * 1e-9
        ext_launches = timing_totals[abi.STAT_NAMES.index("extend_launches")]
        shade_s = timing_totals[abi.STAT_NAMES.index("shade_ns")] * 1e-9
        # the fused kernel leaves (next to nothing in) the shade slot: its launches carry all the work
        fused = shade_s < 0.05 * ext_s
        flops_per_step = extend_flops(st) + (shade_flops(st) if fused else 0.0)
        achieved = flops_per_step * args.steps / ext_s / 1e12 if ext_s > 0 else 0.0
        # the same time split by what each part of the work is rated against: the box tests run
        # in f32 (157.3 TFLOP/s peak), everything else in f64 (78.6)
        box_flops = float(st[abi.STAT_NAMES.index("box_tests")]) * F_BOX
        ideal_s = (box_flops / (PEAK_F32_TFLOPS * 1e12) + (flops_per_step - box_flops) / (PEAK_F64_TFLOPS * 1e12))
        frac_mixed = ideal_s * args.steps / ext_s if ext_s > 0 else 0.0
        rays = float(st[abi.STAT_NAMES.index("rays")])
        result["roofline"] = {
            "bound": "valu",
            "kernel": "trace_kernel (extend + shade fused)" if fused else "extend_kernel",
            "achieved": achieved,
            "peak": PEAK_F64_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / PEAK_F64_TFLOPS,
            "frac_mixed_peak": frac_mixed,
            "traffic": None,
            "launches": int(ext_launches),
            "avg_launch_ms": 1e3 * ext_s / max(ext_launches, 1),
            "kernel_time_share": ext_s / elapsed,
            "shade_time_share": shade_s / elapsed,
            "flops_per_pixel_sample": flops_per_step / (args.width * args.height * args.spp),
            "rays_per_pixel_sample": rays / (args.width * args.height * args.spp),
            "box_tests_per_ray": float(st[abi.STAT_NAMES.index("box_tests")]) / rays,
            "sphere_tests_per_ray": float(st[abi.STAT_NAMES.index("sphere_tests")]) / rays,
            "grays_per_s": rays * args.steps / elapsed / 1e9,
        }
        # HBM bytes per launch come from separate rocprofv3 counter passes over the same kernel and
        # pass size (committed next to the kernel-trace summary); the counters cannot run in here
        tpath = os.path.join(ROOT, "profiles", "r02_trace_kernel_hbm_traffic.json")
        if fused and os.path.exists(tpath) and (args.width, args.height) == (WIDTH, HEIGHT) and args.spp >= 64:
            with open(tpath) as f:
                tj = json.load(f)
            result["roofline"]["traffic"] = tj["hbm_bytes_per_launch"]
            result["roofline"]["traffic_unit"] = ("bytes per launch (HBM, separate PMC passes over one pass of %d spp)"
                                                  % tj.get("samples_per_pass", 64))
        if not args.no_cpu:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import pyoracle as po
            o = po.Oracle(scene.view_ptr)
            cores = po.hardware_threads()  # every host core this process may use
            # bounded sample of the same workload: the whole frame at 8 spp (~5 s on 16 cores)
            cpu_full_spp = 8
            pf = po.make_params(args.width, args.height, 0, cpu_full_spp, depth, SEED)
            c0 = time.perf_counter()
            o.render(pf, po.MODE_REFERENCE, threads=cores)
            cpu_s = time.perf_counter() - c0
            result["cpu_baseline"] = {
                "value": args.width * args.height * cpu_full_spp / cpu_s,
                "unit": "pixel-samples/s",
                "cores": cores,
                "kind": "port",
                "sample": f"the whole {args.width}x{args.height} frame at {cpu_full_spp} spp "
                          f"({args.width * args.height * cpu_full_spp} pixel-samples), reference BVH "
                          "topology, recursive ray_color, f64",
            }
            # image comparison on a centred crop at the full spp
            cw, ch = 64, 36
            window = ((args.width - cw) // 2, (args.height - ch) // 2, cw, ch)
            cpu_spp = args.spp
            pc = po.make_params(args.width, args.height, 0, cpu_spp, depth, SEED, window=window)
            cpu_a = o.render(pc, po.MODE_REFERENCE, threads=cores)
            pc2 = po.make_params(args.width, args.height, 0, cpu_spp, depth, SEED + 1, window=window)
            cpu_b = o.render(pc2, po.MODE_REFERENCE, threads=cores)
            pg = ctx.params(args.width, args.height, 0, cpu_spp, depth, seed=SEED, window=window)
            gpu_a = ctx.render(pg, dtype=np.float64)
            clamp = lambda a: np.clip(a / cpu_spp, 0.0, 255.0 / 256.0)  # noqa: E731
            rmse = lambda a, b: float(np.sqrt(np.mean((clamp(a) - clamp(b)) ** 2)))  # noqa: E731
            result["parity"] = {
                "window": list(window), "spp": cpu_spp,
                "rmse_gpu_vs_cpu_same_seed": rmse(gpu_a, cpu_a),
                "rmse_gpu_seedA_vs_cpu_seedB": rmse(gpu_a, cpu_b),
                "rmse_cpu_two_seed": rmse(cpu_a, cpu_b),
                "max_abs_diff_same_seed": float(np.abs(gpu_a - cpu_a).max() / cpu_spp),
            }

    if rank == 0:
        print(json.dumps(result), flush=True)
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
