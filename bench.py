#!/usr/bin/env python3
"""Headline benchmark: pixel-samples/s on book_cover at 1920x1080, 1024 spp.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python bench.py --config cornell|book2|cover1m      (the other named configurations, same JSON)
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
from raytracer_amd.dist import gather_rank_times, render_sharded  # noqa: E402
from raytracer_amd.render import Context  # noqa: E402
from raytracer_amd.scene import HostScene  # noqa: E402

SCENE = os.path.join(ROOT, "tests", "golden", "book_cover.rtscene")
WIDTH, HEIGHT, SPP, SEED = 1920, 1080, 1024, 1
# The named configurations: scene, frame, samples; `cpu` = the bounded sample the CPU restatement is
# timed on (window or None for the whole frame, samples), sized for 5-30 s on 16 host threads;
# `stats_spp` = samples of the instrumented pass the operation counts come from (scaled to spp);
# `parity_spp` = samples of the crop compared with the CPU restatement.
CONFIGS = {
    "book_cover": dict(scene="book_cover", width=1920, height=1080, spp=1024, cpu=(None, 8), stats_spp=1024, parity_spp=1024,
                       data="synthetic: book_cover.ron's random sphere pattern expanded by the seeded loader "
                            "(scene seed 1, 488 spheres), frozen under tests/golden"),
    "material_demo": dict(scene="material_demo", width=400, height=225, spp=64, cpu=(None, 64), stats_spp=64,
                          parity_spp=64, data="material_demo.ron (4 spheres: the reference's own CPU-runnable case), "
                                              "frozen under tests/golden"),
    "cornell": dict(scene="cornell", width=1024, height=1024, spp=4096, cpu=(None, 4), stats_spp=256, parity_spp=1024,
                    data="cornell.ron (18 rectangles, two rotated and translated boxes), frozen under tests/golden"),
    "book2": dict(scene="book2", width=1920, height=1080, spp=2048, cpu=(None, 2), stats_spp=128, parity_spp=512,
                  data="synthetic: book2.ron expanded by the seeded loader (scene seed 1, 10 370 objects, "
                       "Perlin tables, decoded earthmap texels), frozen under tests/golden"),
    "cover1m": dict(scene=None, width=3840, height=2160, spp=1024, cpu=((1800, 1012, 240, 135), 1), stats_spp=32, parity_spp=16, parity_crop=(32, 18),
                    data="synthetic: book_cover.ron's pattern scaled to a 1000x1000 grid by the seeded generator "
                         "(scene seed 1, 1 000 004 spheres), hierarchy built on the device"),
}

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


def kernel_source_sha():
    """Hash of the device sources a profile belongs to: a committed counter file names the hash it
    was measured on, and its numbers are quoted only while the sources still have it."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "raytracer_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")):
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


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


def note(msg):
    """Progress on stderr: a long configuration must not look hung to whoever runs it."""
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="book_cover")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline and parity legs")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    full_size = args.width is None and args.height is None and args.spp is None
    args.width = args.width or cfg["width"]
    args.height = args.height or cfg["height"]
    args.spp = args.spp or cfg["spp"]

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

    if cfg["scene"] is None:  # the 1 000 004-sphere scale-up of book_cover.ron
        scene = HostScene.generate_cover(500, scene_seed=1, width=args.width, height=args.height, spp=args.spp,
                                         max_depth=50)
    else:
        scene = HostScene.load_frozen(os.path.join(ROOT, "tests", "golden", cfg["scene"] + ".rtscene"))
        scene.set_image(args.width, args.height)
    depth = scene.max_depth
    ctx = Context(device=local_rank)
    ctx.upload(scene)
    out = torch.zeros((args.height, args.width, 3), dtype=torch.float32, device=device)

    timing_totals = np.zeros(abi.RT_STAT_COUNT, dtype=np.float64)

    render_s = [0.0]

    def render_fn(lo, hi, buf, timing=False):
        p = ctx.params(args.width, args.height, lo, hi, depth, seed=SEED)
        r0 = time.perf_counter()
        st = ctx.render_into(p, buf.data_ptr(), timing=timing)
        if timing:
            render_s[0] += time.perf_counter() - r0
            timing_totals[:] += st.astype(np.float64)

    def step(timing):
        render_sharded(lambda lo, hi, buf: render_fn(lo, hi, buf, timing), out, 0, args.spp, rank,
                       world, dist if use_dist else None)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    note(f"{args.config}: scene uploaded, {args.warmup} warm-up + {args.steps} timed steps")
    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    per_rank = None
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        per_rank = gather_rank_times(dist, world, render_s[0], elapsed, args.steps, device)
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
        "data": cfg["data"],
        "config": {
            "workload": f"{args.config} {args.width}x{args.height}, {args.spp} spp, max_depth {depth}",
            "scene": f"tests/golden/{cfg['scene']}.rtscene" if cfg["scene"] else "rt_host_generate_cover(500)",
            "width": args.width, "height": args.height, "spp": args.spp, "max_depth": depth,
            "sampling_seed": SEED,
            "parallelism": f"sample-range x{world} + one reduce(sum) to rank 0",
        },
    }

    if per_rank is not None:
        result["per_rank_ms_per_step"] = per_rank

    if world == 1 and rank == 0:
        # operation counts of the work of one step, from the instrumented kernels (the headline
        # configuration: exactly its samples; the longer ones: a stated number of samples, scaled)
        note(f"timed steps done: {1e3 * elapsed / max(args.steps, 1):.1f} ms per step; instrumented pass")
        stats_spp = min(args.spp, cfg["stats_spp"])
        p = ctx.params(args.width, args.height, 0, stats_spp, depth, seed=SEED)
        _, st = ctx.render(p, dtype=np.float32, stats=True)
        st = st.astype(np.float64) * (args.spp / stats_spp)
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
            "rect_tests_per_ray": float(st[abi.STAT_NAMES.index("rect_tests")]) / rays,
            "grays_per_s": rays * args.steps / elapsed / 1e9,
            "counts_from": f"instrumented pass of {stats_spp} of the {args.spp} samples"
                           + ("" if stats_spp == args.spp else ", scaled"),
        }
        # HBM bytes come from separate rocprofv3 counter passes over the same kernels
        # (tools/profile_configs.sh; the counters cannot run in here).  The committed file names the
        # hash of the device sources it was measured on: numbers of other sources are not quoted.
        tpath = os.path.join(ROOT, "profiles", f"r03_{args.config}_roofline.json")
        if full_size and os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            d = tj.get("derived", {})
            if tj.get("kernel_source_sha") == kernel_source_sha() and "hbm_bytes_per_pixel_sample" in d:
                # per launch of THIS run: bytes per pixel-sample of the profiled pass x this run's
                # pixel-samples per launch
                per_ps = d["hbm_bytes_per_pixel_sample"]
                result["roofline"]["traffic"] = per_ps * args.width * args.height * args.spp * args.steps / max(ext_launches, 1)
                result["roofline"]["traffic_unit"] = "HBM bytes per launch of this run"
                result["roofline"]["traffic_per_pixel_sample"] = per_ps
                result["roofline"]["traffic_per_ray"] = d.get("hbm_bytes_per_ray")
                result["roofline"]["traffic_source"] = f"profiles/r03_{args.config}_roofline.json (" + tj.get("config", "") + ")"
                for k in ("lanes_active", "valu_issue_share", "waves_per_simd", "l2_hit_rate", "hbm_gbps"):
                    if k in d:
                        result["roofline"][k] = d[k]
            else:
                result["roofline"]["traffic_source"] = (f"profiles/r03_{args.config}_roofline.json is of other device "
                                                        "sources than this build: not quoted")
        if not args.no_cpu:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import pyoracle as po
            o = po.Oracle(scene.view_ptr)
            cores = po.hardware_threads()  # every host core this process may use
            # bounded sample of the same workload (see CONFIGS)
            note("CPU restatement: bounded sample")
            cpu_window, cpu_full_spp = cfg["cpu"]
            pf = po.make_params(args.width, args.height, 0, cpu_full_spp, depth, SEED, window=cpu_window)
            cpu_pixels = args.width * args.height if cpu_window is None else cpu_window[2] * cpu_window[3]
            c0 = time.perf_counter()
            o.render(pf, po.MODE_REFERENCE, threads=cores)
            cpu_s = time.perf_counter() - c0
            result["cpu_baseline"] = {
                "value": cpu_pixels * cpu_full_spp / cpu_s,
                "unit": "pixel-samples/s",
                "cores": cores,
                "kind": "port",
                "sample": (f"the whole {args.width}x{args.height} frame" if cpu_window is None else
                           f"the window {cpu_window} of the {args.width}x{args.height} frame")
                          + f" at {cpu_full_spp} spp ({cpu_pixels * cpu_full_spp} pixel-samples, {cpu_s:.1f} s), "
                          "reference BVH topology, recursive ray_color, f64",
            }
            # image comparison on a centred crop at the full spp
            cw, ch = cfg.get("parity_crop", (64, 36))
            window = ((args.width - cw) // 2, (args.height - ch) // 2, cw, ch)
            note(f"CPU restatement: {cpu_s:.1f} s; parity crop, two seeds")
            cpu_spp = min(args.spp, cfg["parity_spp"])
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
