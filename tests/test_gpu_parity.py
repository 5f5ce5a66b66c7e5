"""GPU parity: the HIP path through the C ABI against the CPU oracle.

Bars (written where they are applied):
  * first-hit object id: exact.  t and normal: the north-star allows 1e-5
    relative; scenes whose hit arithmetic uses only + - * / sqrt are required to
    match bit for bit, the others to 1e-12.
  * per-pixel radiance sums: bit-exact against the oracle's iterative estimator
    for scenes without libm calls in the path (sin/acos/atan2/log10 differ in the
    last place between glibc and the device library); 1e-9 relative otherwise;
    and 1e-12 relative against the golden sums of the literal recursive
    ray_color (association of the attenuation products differs).
"""
import os

import numpy as np
import pytest

from raytracer_amd.scene import HostScene
import pyoracle as po

pytestmark = pytest.mark.gpu

SCENES = ["material_demo", "book_cover", "cornell", "book2", "medium_demo", "emissive",
          "checker_mirror", "motion_blur", "globe", "refraction", "perlin"]
# scenes whose render path never calls a libm function on the device
EXACT_RENDER = {"material_demo", "book_cover", "cornell", "motion_blur", "refraction"}


def load(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".golden.npz"))
    scene = HostScene.load_frozen(os.path.join(golden_dir, name + ".rtscene"))
    scene.set_image(int(g["width"]), int(g["height"]))
    return scene, g


@pytest.mark.parametrize("name", SCENES)
def test_first_hit_matches_golden(gpu_ctx, golden_dir, name):
    scene, g = load(golden_dir, name)
    gpu_ctx.upload(scene)
    ids, t, n = gpu_ctx.first_hit(scene.width, scene.height)
    assert np.array_equal(ids, g["first_id"]), f"{(ids != g['first_id']).sum()} ids differ"
    hit = ids != 0xFFFFFFFF
    assert hit.any()
    # only + - * / sqrt on this path: IEEE-exact on both sides
    assert np.array_equal(t[hit], g["first_t"][hit])
    assert np.array_equal(n[hit], g["first_normal"][hit])


@pytest.mark.parametrize("pipeline", ["default", "binned"])
@pytest.mark.parametrize("name", SCENES)
def test_render_window_matches_oracle(gpu_ctx, golden_dir, name, pipeline):
    """default: fused trace kernel for solid-texture scenes; binned: extend + shade per material."""
    from raytracer_amd import abi
    scene, g = load(golden_dir, name)
    gpu_ctx.upload(scene)
    window = tuple(int(x) for x in g["window"])
    spp, seed, depth = int(g["spp"]), int(g["seed"]), int(g["max_depth"])
    flags = abi.RT_FLAG_BINNED if pipeline == "binned" else 0
    p = gpu_ctx.params(scene.width, scene.height, 0, spp, depth, seed=seed, window=window,
                       flags=flags)
    got = gpu_ctx.render(p, dtype=np.float64)
    golden = g["render_sum"]
    scale = np.abs(golden).max()
    o = po.Oracle(scene.view_ptr)
    pi = po.make_params(scene.width, scene.height, 0, spp, depth, seed, window=window)
    iterative = o.render(pi, po.MODE_ITERATIVE, threads=4)
    if name in EXACT_RENDER:
        assert np.array_equal(got, iterative), f"max |diff| {np.abs(got - iterative).max()}"
        assert np.abs(got - golden).max() <= 1e-12 * scale
    else:
        # a last-place difference in sin/acos/atan2/log10 moves a colour or a free-flight
        # distance by one ulp; a handful of paths may flip a discrete choice
        close = np.isclose(got, iterative, rtol=1e-9, atol=1e-9 * scale)
        assert close.mean() > 0.995, f"{(~close).sum()} of {close.size} values differ"


def test_float_output_is_rounded_double(gpu_ctx, golden_dir):
    scene, g = load(golden_dir, "material_demo")
    gpu_ctx.upload(scene)
    p = gpu_ctx.params(scene.width, scene.height, 0, 4, 50, seed=3)
    a = gpu_ctx.render(p, dtype=np.float64)
    p = gpu_ctx.params(scene.width, scene.height, 0, 4, 50, seed=3)
    b = gpu_ctx.render(p, dtype=np.float32)
    assert np.array_equal(a.astype(np.float32), b)


def test_sample_ranges_add_up(gpu_ctx, golden_dir):
    """Sharding the sample range (what each GPU does) changes only the order of the sum."""
    scene, g = load(golden_dir, "book_cover")
    gpu_ctx.upload(scene)
    w, h = scene.width, scene.height
    whole = gpu_ctx.render(gpu_ctx.params(w, h, 0, 8, 50, seed=5), dtype=np.float64)
    lo = gpu_ctx.render(gpu_ctx.params(w, h, 0, 4, 50, seed=5), dtype=np.float64)
    hi = gpu_ctx.render(gpu_ctx.params(w, h, 4, 8, 50, seed=5), dtype=np.float64)
    assert np.abs(whole - (lo + hi)).max() <= 1e-13 * np.abs(whole).max()


@pytest.mark.parametrize("pipeline", ["default", "binned", "streamed"])
def test_pass_size_does_not_change_the_image(gpu_ctx, golden_dir, pipeline, monkeypatch):
    """A small path budget splits the samples into several passes; the per-pixel sums stay in sample order.
    ("streamed": the small passes too keep their path state at queue positions, so the two sets of
    arrays and the queues are reused pass after pass.)"""
    from raytracer_amd import abi
    from raytracer_amd.render import Context
    scene, g = load(golden_dir, "book_cover")
    w, h = scene.width, scene.height
    flags = abi.RT_FLAG_BINNED if pipeline == "binned" else 0
    if pipeline == "streamed":
        monkeypatch.setenv("MI355RT_STREAM_MIN_PATHS", "0")  # (read at upload; the fixture's context keeps its own setting)
    gpu_ctx.upload(scene)
    one = gpu_ctx.render(gpu_ctx.params(w, h, 0, 7, 50, seed=11, flags=flags), dtype=np.float64)
    small = Context(device=0, max_paths_in_flight=2 * w * h + 5)  # 2 samples per pass: 4 passes
    try:
        small.upload(scene)
        many = small.render(small.params(w, h, 0, 7, 50, seed=11, flags=flags), dtype=np.float64)
    finally:
        small.close()
    assert np.array_equal(one, many)


@pytest.mark.parametrize("pipeline", ["default", "binned"])
def test_overhanging_cuboid_sides_follow_the_reference_boxes(gpu_ctx, pipeline):
    """First hits and radiance against the reference-topology oracle on a scene whose cuboid Y
    sides reach outside their bounding boxes (plain, rotated+translated, and as a medium)."""
    from raytracer_amd import abi
    from raytracer_amd.scene import HostScene
    from test_abi import OVERHANG_SCENE
    scene = HostScene.load_ron_text(OVERHANG_SCENE)
    gpu_ctx.upload(scene)
    w, h = 64, 48
    o = po.Oracle(scene.view_ptr)
    ids, t, n = gpu_ctx.first_hit(w, h)
    rid, rt_, rn = o.first_hit(w, h)
    assert np.array_equal(ids, rid)
    assert np.array_equal(t, rt_) and np.array_equal(n, rn)
    flags = abi.RT_FLAG_BINNED if pipeline == "binned" else 0
    img = gpu_ctx.render(gpu_ctx.params(w, h, 0, 3, 6, seed=4, flags=flags), dtype=np.float64)
    ref = o.render(po.make_params(w, h, 0, 3, 6, 4), po.MODE_ITERATIVE, threads=2)
    assert np.allclose(img, ref, rtol=1e-9, atol=1e-12)  # log10 in the medium: last-place libm differences


def many_media_scene(n):
    """n concentric constant media (thin, so most rays cross all of them) in front of a sphere."""
    media = "".join(
        f"ConstantMedium(boundary: Sphere(center: (0, 0, -3), radius: {0.4 + 0.1 * k}, material: Shared(\"grey\")),"
        f" density: {0.05 + 0.01 * k}, texture: Solid({0.1 + 0.07 * k}, 0.5, {0.9 - 0.07 * k})),\n" for k in range(n))
    return """Scene(
  image: (width: 48, height: 36, samples_per_pixel: 2, max_depth: 8),
  camera: (look_from: (0, 0.5, 2), look_at: (0, 0, -3), vertical_fov: 50, aperture: 0),
  background: (0.6, 0.7, 1.0),
  materials: { "grey": Lambertian(albedo: Solid(0.5, 0.5, 0.5)) },
  objects: [
    Sphere(center: (0, -1002, -3), radius: 1000, material: Shared("grey")),
    Sphere(center: (0.3, 0, -3), radius: 0.2, material: Shared("grey")),
""" + media + "  ]\n)"


@pytest.mark.parametrize("pipeline", ["default", "binned"])
def test_more_media_on_a_ray_than_the_postponed_list_holds(gpu_ctx, pipeline):
    """Twelve nested media: a traversal sets media aside in a list of eight, so a ray through the
    middle fills it, has it settled mid-walk and carries on.  Against the reference-topology oracle."""
    from raytracer_amd import abi
    scene = HostScene.load_ron_text(many_media_scene(12))
    gpu_ctx.upload(scene)
    w, h = 48, 36
    o = po.Oracle(scene.view_ptr)
    flags = abi.RT_FLAG_BINNED if pipeline == "binned" else 0
    img = gpu_ctx.render(gpu_ctx.params(w, h, 0, 4, 8, seed=9, flags=flags), dtype=np.float64)
    ref = o.render(po.make_params(w, h, 0, 4, 8, 9), po.MODE_ITERATIVE, threads=2)
    assert np.allclose(img, ref, rtol=1e-9, atol=1e-12)  # log10 in the medium: last-place libm differences
    ids, t, n = gpu_ctx.first_hit(w, h)  # (the first-hit pass skips media)
    rid, rt_, rn = o.first_hit(w, h)
    assert np.array_equal(ids, rid) and np.array_equal(t, rt_) and np.array_equal(n, rn)


def coincident_scene(n):
    """n copies of one sphere and n of another (every surface n times: all hits tie in t) on a ground sphere."""
    objs = "".join(f"Sphere(center: ({-0.6 if k % 2 else 0.6}, 0, -3), radius: 0.5, material: Shared(\"m{k % 3}\")),\n"
                   for k in range(2 * n))
    return """Scene(
  image: (width: 64, height: 48, samples_per_pixel: 2, max_depth: 6),
  camera: (look_from: (0, 0.5, 1), look_at: (0, 0, -3), vertical_fov: 40, aperture: 0),
  background: (0.6, 0.7, 1.0),
  materials: { "m0": Lambertian(albedo: Solid(0.8, 0.3, 0.3)), "m1": Metal(albedo: (0.8, 0.8, 0.8), fuzziness: 0.1),
               "m2": Lambertian(albedo: Solid(0.2, 0.3, 0.8)) },
  objects: [
    Sphere(center: (0, -1000.5, -3), radius: 1000, material: Shared("m0")),
""" + objs + "  ]\n)"


def nested_scene(n):
    """n concentric glass shells: clustering merges them one at a time into a hierarchy n levels deep."""
    objs = "".join(f"Sphere(center: (0, 0, -3), radius: {0.2 + 0.004 * k}, material: Shared(\"glass\")),\n"
                   for k in range(n))
    return """Scene(
  image: (width: 64, height: 48, samples_per_pixel: 2, max_depth: 6),
  camera: (look_from: (0, 0.5, 1), look_at: (0, 0, -3), vertical_fov: 40, aperture: 0),
  background: (0.6, 0.7, 1.0),
  materials: { "glass": Dielectric(index_of_refraction: 1.5), "m0": Lambertian(albedo: Solid(0.8, 0.3, 0.3)) },
  objects: [
    Sphere(center: (0, -1000.5, -3), radius: 1000, material: Shared("m0")),
""" + objs + "  ]\n)"


@pytest.mark.parametrize("which", ["cover", "book2", "two_spheres", "coincident", "nested"])
def test_device_built_hierarchy_gives_the_host_built_results(gpu_ctx, golden_dir, monkeypatch, which):
    """The hierarchy and the reference's resolution order made on the device (large scenes get
    them by default) against the host's: the order itself must be the host's (checked inside the
    upload), first hits and radiance must not depend on which hierarchy is walked."""
    if which == "cover":
        scene = HostScene.generate_cover(60, scene_seed=5, width=160, height=90, spp=2, max_depth=12)
    elif which == "book2":
        scene, _ = load(golden_dir, "book2")
    elif which == "coincident":  # identical boxes by the hundred, and every hit a tie decided by rank
        scene = HostScene.load_ron_text(coincident_scene(300))
    elif which == "nested":  # deeper than the device build may go: it hands over to the host's
        scene = HostScene.load_ron_text(nested_scene(200))
    else:
        scene, _ = load(golden_dir, "material_demo")
    w, h = scene.width, scene.height
    out = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("MI355RT_DEVICE_BUILD", mode)
        monkeypatch.setenv("MI355RT_CHECK_ORDER", "1")
        gpu_ctx.upload(scene)
        ids, t, n = gpu_ctx.first_hit(w, h)
        img = gpu_ctx.render(gpu_ctx.params(w, h, 0, 2, 12, seed=3), dtype=np.float64)
        out[mode] = (ids, t, n, img)
    for a, b in zip(out["0"], out["1"]):
        assert np.array_equal(a, b)
    if which == "coincident":  # and the ties resolve as the reference's own hierarchy would
        o = po.Oracle(scene.view_ptr)
        rid, rt_, rn = o.first_hit(w, h)
        assert np.array_equal(out["1"][0], rid) and np.array_equal(out["1"][1], rt_)


@pytest.mark.parametrize("steps", ["1", "2", "3"])
def test_failed_device_build_falls_back_to_the_host(gpu_ctx, monkeypatch, steps):
    """Either device step of the upload may fail (bit 0: the hierarchy, bit 1: the reference's
    order); the host then does that step and the upload succeeds with the same results."""
    scene = HostScene.generate_cover(30, scene_seed=9, width=128, height=72, spp=2, max_depth=8)
    w, h = scene.width, scene.height
    monkeypatch.setenv("MI355RT_DEVICE_BUILD", "0")
    gpu_ctx.upload(scene)
    want = (gpu_ctx.first_hit(w, h)[0], gpu_ctx.render(gpu_ctx.params(w, h, 0, 2, 8, seed=3), dtype=np.float64))
    monkeypatch.setenv("MI355RT_DEVICE_BUILD", "1")
    monkeypatch.setenv("MI355RT_FAIL_DEVICE_BUILD", steps)
    gpu_ctx.upload(scene)
    assert np.array_equal(gpu_ctx.first_hit(w, h)[0], want[0])
    assert np.array_equal(gpu_ctx.render(gpu_ctx.params(w, h, 0, 2, 8, seed=3), dtype=np.float64), want[1])


def test_stats_match_flat_oracle(gpu_ctx, golden_dir):
    """The device walks the flattened BVH exactly as the instrumented CPU traversal does."""
    from raytracer_amd.render import build_bvh
    scene, g = load(golden_dir, "book_cover")
    gpu_ctx.upload(scene)
    window = (40, 30, 32, 16)
    from raytracer_amd import abi
    p = gpu_ctx.params(scene.width, scene.height, 0, 4, 50, seed=9, window=window)
    img, st = gpu_ctx.render(p, dtype=np.float64, stats=True)
    p2 = gpu_ctx.params(scene.width, scene.height, 0, 4, 50, seed=9, window=window,
                        flags=abi.RT_FLAG_BINNED)
    img2, st2 = gpu_ctx.render(p2, dtype=np.float64, stats=True)
    assert np.array_equal(img, img2) and list(st[:20]) == list(st2[:20])
    nodes, items, _ = build_bvh(scene)
    o = po.Oracle(scene.view_ptr)
    pi = po.make_params(scene.width, scene.height, 0, 4, 50, 9, window=window)
    ref, rs = o.render(pi, po.MODE_FLAT, bvh=(nodes, items), threads=2, stats=True)
    assert np.array_equal(img, ref)
    assert list(st[:16]) == list(rs[:16])
    assert st[abi.STAT_NAMES.index("node_visits")] == rs[abi.STAT_NAMES.index("node_visits")] > 0


@pytest.mark.parametrize("name", ["book2", "perlin", "globe", "checker_mirror", "emissive", "medium_demo"])
def test_deferred_texture_hits_give_the_binned_image(gpu_ctx, golden_dir, name):
    """Default pipeline (fused, hits on procedural / image textures deferred to their own shade
    kernel) against the binned one: same image bit for bit, same operation counts."""
    from raytracer_amd import abi
    scene, g = load(golden_dir, name)
    w, h = scene.width, scene.height
    gpu_ctx.upload(scene)
    depth = scene.max_depth
    a, sa = gpu_ctx.render(gpu_ctx.params(w, h, 0, 3, depth, seed=4), dtype=np.float64, stats=True)
    b, sb = gpu_ctx.render(gpu_ctx.params(w, h, 0, 3, depth, seed=4, flags=abi.RT_FLAG_BINNED),
                           dtype=np.float64, stats=True)
    assert np.array_equal(a, b)
    assert list(sa[:20]) == list(sb[:20])
    # one sample at the last allowed bounce only: the depth budget ends deferred paths too
    a1 = gpu_ctx.render(gpu_ctx.params(w, h, 0, 2, 1, seed=4), dtype=np.float64)
    b1 = gpu_ctx.render(gpu_ctx.params(w, h, 0, 2, 1, seed=4, flags=abi.RT_FLAG_BINNED), dtype=np.float64)
    assert np.array_equal(a1, b1)


@pytest.mark.parametrize("block,per_cu", [("256", "1"), ("256", "4"), ("256", "16"), ("512", "1"), ("512", "2"),
                                          ("512", "8")])
def test_grid_shape_does_not_change_the_image(gpu_ctx, golden_dir, monkeypatch, block, per_cu):
    """Workgroup size and grid size of the fused kernel only change which wave carries which path:
    image and operation counts stay those of the binned pipeline (mixed scenes: some hits shaded
    in place, some deferred, with and without media)."""
    from raytracer_amd import abi
    for name in ("checker_mirror", "book2", "book_cover"):
        scene, g = load(golden_dir, name)
        w, h = scene.width, scene.height
        depth = scene.max_depth
        gpu_ctx.upload(scene)
        want, sw = gpu_ctx.render(gpu_ctx.params(w, h, 0, 3, depth, seed=4, flags=abi.RT_FLAG_BINNED),
                                  dtype=np.float64, stats=True)
        monkeypatch.setenv("MI355RT_TRACE_BLOCK", block)
        monkeypatch.setenv("MI355RT_TRACE_BLOCKS_PER_CU", per_cu)
        gpu_ctx.upload(scene)
        got, sg = gpu_ctx.render(gpu_ctx.params(w, h, 0, 3, depth, seed=4), dtype=np.float64, stats=True)
        monkeypatch.delenv("MI355RT_TRACE_BLOCK")
        monkeypatch.delenv("MI355RT_TRACE_BLOCKS_PER_CU")
        assert list(sw[:20]) == list(sg[:20]), name
        assert np.array_equal(want, got), name


@pytest.mark.parametrize("name", ["book_cover", "book2"])
def test_stack_overflow_area_gives_the_same_image(gpu_ctx, golden_dir, name, monkeypatch):
    """With almost no LDS stack, nearly every stacked entry goes through the global overflow area."""
    scene, g = load(golden_dir, name)
    w, h = scene.width, scene.height
    gpu_ctx.upload(scene)
    want = gpu_ctx.render(gpu_ctx.params(w, h, 0, 3, 50, seed=2), dtype=np.float64)
    want_ids = gpu_ctx.first_hit(w, h)[0]
    monkeypatch.setenv("MI355RT_STACK_CAP", "3")
    gpu_ctx.upload(scene)
    got = gpu_ctx.render(gpu_ctx.params(w, h, 0, 3, 50, seed=2), dtype=np.float64)
    got_ids = gpu_ctx.first_hit(w, h)[0]
    monkeypatch.delenv("MI355RT_STACK_CAP")
    gpu_ctx.upload(scene)
    assert np.array_equal(want, got) and np.array_equal(want_ids, got_ids)


@pytest.mark.parametrize("name", ["book_cover", "material_demo"])
def test_sphere_only_kernel_gives_the_plain_kernels_image(gpu_ctx, golden_dir, name, monkeypatch):
    """Scenes of plain spheres run a trace kernel of their own (32-byte leaf slots in LDS, one-case
    leaf test and hit record); MI355RT_SPHERES_KERNEL=0 sends them through the plain one.  Same
    image, same operation counts, and both are the oracle's image."""
    import pyoracle as po
    scene, g = load(golden_dir, name)
    w, h = scene.width, scene.height
    depth = scene.max_depth
    gpu_ctx.upload(scene)
    a, sa = gpu_ctx.render(gpu_ctx.params(w, h, 0, 5, depth, seed=6), dtype=np.float64, stats=True)
    monkeypatch.setenv("MI355RT_SPHERES_KERNEL", "0")
    gpu_ctx.upload(scene)
    b, sb = gpu_ctx.render(gpu_ctx.params(w, h, 0, 5, depth, seed=6), dtype=np.float64, stats=True)
    monkeypatch.delenv("MI355RT_SPHERES_KERNEL")
    gpu_ctx.upload(scene)
    assert np.array_equal(a, b)
    assert list(sa[:20]) == list(sb[:20])
    ref = po.Oracle(scene.view_ptr).render(po.make_params(w, h, 0, 5, depth, 6), po.MODE_ITERATIVE, threads=8)
    assert np.array_equal(a, ref)


@pytest.mark.parametrize("name,tail", [("book_cover", None), ("book_cover", "0"), ("material_demo", None),
                                       ("cornell", None), ("cornell", "0"), ("emissive", None),
                                       ("medium_demo", None), ("motion_blur", None), ("refraction", "64"),
                                       ("book2", None), ("book2", "0"), ("perlin", None), ("globe", "0"),
                                       ("checker_mirror", None)])
def test_streamed_path_state_gives_the_indexed_image(gpu_ctx, golden_dir, name, tail, monkeypatch):
    """The fused kernels keep a path's ray and attenuation at its position in the live queue (read
    and written densely, two sets of arrays taking turns; a deferred textured hit is shaded from the
    position it was found at); MI355RT_STREAM=0 keeps them at the path's index.  Same paths, same
    draws, the same sums: the images are equal bit for bit, with the tail launch, without it
    (every bounce a launch of its own) and with a tail of one wave."""
    scene, g = load(golden_dir, name)
    w, h = scene.width, scene.height
    depth = scene.max_depth
    if tail is not None:
        monkeypatch.setenv("MI355RT_TAIL_PATHS", tail)
    monkeypatch.setenv("MI355RT_STREAM_MIN_PATHS", "0")  # (by default only passes of 2^24 paths and more stream)
    gpu_ctx.upload(scene)
    a, sa = gpu_ctx.render(gpu_ctx.params(w, h, 0, 6, depth, seed=8), dtype=np.float64, stats=True)
    c = gpu_ctx.render(gpu_ctx.params(w, h, 0, 6, depth, seed=8), dtype=np.float64)  # (the timed kernels)
    monkeypatch.setenv("MI355RT_STREAM", "0")
    gpu_ctx.upload(scene)
    b, sb = gpu_ctx.render(gpu_ctx.params(w, h, 0, 6, depth, seed=8), dtype=np.float64, stats=True)
    monkeypatch.delenv("MI355RT_STREAM")
    monkeypatch.delenv("MI355RT_STREAM_MIN_PATHS")
    if tail is not None:
        monkeypatch.delenv("MI355RT_TAIL_PATHS")
    gpu_ctx.upload(scene)
    assert np.array_equal(a, b) and np.array_equal(a, c)
    assert list(sa[:20]) == list(sb[:20])


def test_streamed_and_indexed_renders_share_a_context(gpu_ctx, golden_dir, monkeypatch):
    """One context renders a large streamed pass, then a smaller pass of a scene whose textured hits
    are deferred (state by path index, eight sub-queues as far apart as the LARGEST pass so far),
    then the binned pipeline on the first scene: the buffers grow to what each layout needs."""
    import pyoracle as po
    cover, _ = load(golden_dir, "book_cover")
    perlin, _ = load(golden_dir, "perlin")
    cover.set_image(512, 512)
    gpu_ctx.upload(cover)
    big = gpu_ctx.render(gpu_ctx.params(512, 512, 0, 64, 50, seed=2), dtype=np.float64)  # (2^24 paths: streamed)
    w, h = perlin.width, perlin.height
    monkeypatch.setenv("MI355RT_STREAM", "0")
    gpu_ctx.upload(perlin)
    monkeypatch.delenv("MI355RT_STREAM")
    small = gpu_ctx.render(gpu_ctx.params(w, h, 0, 4, perlin.max_depth, seed=2), dtype=np.float64)
    ref = po.Oracle(perlin.view_ptr).render(po.make_params(w, h, 0, 4, perlin.max_depth, 2), po.MODE_ITERATIVE, threads=8)
    assert np.allclose(small, ref, rtol=1e-9, atol=1e-12)
    gpu_ctx.upload(cover)
    binned = gpu_ctx.render(gpu_ctx.params(512, 512, 0, 64, 50, seed=2, flags=4), dtype=np.float64)
    assert np.array_equal(big, binned)


@pytest.mark.parametrize("streamed", [False, True])
@pytest.mark.parametrize("w,h,spp", [(2, 2, 1), (3, 5, 2), (9, 7, 3), (8, 8, 1), (64, 1 + 1, 5), (65, 3, 1)])
def test_renders_smaller_than_a_wave(gpu_ctx, golden_dir, w, h, spp, streamed, monkeypatch):
    """Images of a few pixels: fewer paths than a wave has lanes, queues shorter than the eight
    stretches the work counters split them into, chunk reservations that are mostly empty; windows
    that do and do not divide into 8x8 tiles.  The oracle's image, with the path state by index
    and at queue positions."""
    import pyoracle as po
    scene, g = load(golden_dir, "material_demo")
    scene.set_image(w, h)
    if streamed:
        monkeypatch.setenv("MI355RT_STREAM_MIN_PATHS", "0")
    gpu_ctx.upload(scene)
    got = gpu_ctx.render(gpu_ctx.params(w, h, 0, spp, scene.max_depth, seed=4), dtype=np.float64)
    ref = po.Oracle(scene.view_ptr).render(po.make_params(w, h, 0, spp, scene.max_depth, 4), po.MODE_ITERATIVE, threads=2)
    assert np.array_equal(got, ref)
    ids, t, n = gpu_ctx.first_hit(w, h)
    assert ids.shape == (h, w)


def test_a_tuning_variable_that_is_unset_no_longer_applies(gpu_ctx, golden_dir, monkeypatch):
    """The tuning variables are read at every upload, each from its default: MI355RT_TAIL_PATHS=0
    makes every bounce a launch of its own, and an upload after the variable is gone brings the
    tail launch (and the launch count) back."""
    from raytracer_amd import abi
    scene, g = load(golden_dir, "material_demo")
    w, h = scene.width, scene.height
    at = abi.STAT_NAMES.index("extend_launches")

    def launches():
        gpu_ctx.upload(scene)
        img, st = gpu_ctx.render(gpu_ctx.params(w, h, 0, 4, scene.max_depth, seed=3), dtype=np.float64, timing=True)
        return img, int(st[at])

    a, default = launches()
    monkeypatch.setenv("MI355RT_TAIL_PATHS", "0")
    b, without_tail = launches()
    monkeypatch.delenv("MI355RT_TAIL_PATHS")
    c, again = launches()
    assert without_tail > default and again == default
    assert np.array_equal(a, b) and np.array_equal(a, c)


@pytest.mark.parametrize("name", ["perlin", "globe", "book2", "checker_mirror"])
def test_textured_tail_gives_the_image_of_the_deferred_kernel(gpu_ctx, golden_dir, name, monkeypatch):
    """Once few paths are left, a scene with procedural or image textures finishes them in one
    launch that shades such hits in place; without it (MI355RT_TEXTURED_TAIL=0) every bounce runs
    the trace kernel and the deferred shade kernel.  Same paths, same draws: the same numbers."""
    scene, g = load(golden_dir, name)
    w, h = scene.width, scene.height
    depth = scene.max_depth
    assert depth > 9  # (the live count is looked at after the eighth bounce)
    gpu_ctx.upload(scene)
    a = gpu_ctx.render(gpu_ctx.params(w, h, 0, 4, depth, seed=8), dtype=np.float64)
    monkeypatch.setenv("MI355RT_TEXTURED_TAIL", "0")
    gpu_ctx.upload(scene)
    b = gpu_ctx.render(gpu_ctx.params(w, h, 0, 4, depth, seed=8), dtype=np.float64)
    monkeypatch.delenv("MI355RT_TEXTURED_TAIL")
    gpu_ctx.upload(scene)
    assert np.array_equal(a, b)
    assert a.max() > 0


@pytest.mark.parametrize("name,w,h,window", [("book_cover", 320, 184, None), ("book_cover", 640, 360, (200, 96, 64, 40)),
                                              ("material_demo", 160, 96, None), ("motion_blur", 160, 56, None),
                                              ("cornell", 96, 96, None), ("refraction", 160, 56, (8, 8, 48, 40)),
                                              ("checker_mirror", 160, 120, None), ("book2", 160, 96, None),
                                              ("medium_demo", 160, 96, (16, 8, 96, 64))])
def test_camera_ray_tile_lists_give_the_walks_image(gpu_ctx, golden_dir, name, w, h, window, monkeypatch):
    """The first bounce of a plain scene tests the leaf slots its 8x8 tile of pixels can meet (made
    once per call from a conservative beam) instead of walking the hierarchy per ray;
    MI355RT_TILE_LISTS=0 walks.  Same closest hits, so the same numbers; and both are the oracle's."""
    import pyoracle as po
    scene, g = load(golden_dir, name)
    scene.set_image(w, h)
    depth = min(scene.max_depth, 12)
    gpu_ctx.upload(scene)
    a = gpu_ctx.render(gpu_ctx.params(w, h, 0, 6, depth, seed=12, window=window), dtype=np.float64)
    monkeypatch.setenv("MI355RT_TILE_LISTS", "0")
    gpu_ctx.upload(scene)
    b = gpu_ctx.render(gpu_ctx.params(w, h, 0, 6, depth, seed=12, window=window), dtype=np.float64)
    monkeypatch.delenv("MI355RT_TILE_LISTS")
    gpu_ctx.upload(scene)
    assert np.array_equal(a, b)
    ref = po.Oracle(scene.view_ptr).render(po.make_params(w, h, 0, 6, depth, 12, window=window),
                                           po.MODE_ITERATIVE, threads=8)
    if name in ("checker_mirror", "book2", "medium_demo"):  # (sin, log10 and friends: libm's last place)
        assert np.allclose(a, ref, rtol=1e-9, atol=1e-12)
    else:
        assert np.array_equal(a, ref)


@pytest.mark.parametrize("aperture,focus,fov,look_from", [(2.0, 2.0, 80.0, (0.3, 0.4, 3.0)), (0.8, 12.0, 35.0, (6.0, 1.0, 2.0)),
                                                         (4.0, 0.7, 100.0, (0.0, 0.2, 1.5)), (0.0, 5.0, 20.0, (0.0, 9.0, 0.1))])
def test_tile_lists_under_wide_lenses(gpu_ctx, aperture, focus, fov, look_from):
    """Cameras that strain the beam bound: lenses wider than the objects, a focal plane in front of,
    inside and behind them, steep view angles.  Spheres and a rotated, translated box; the image
    must be the oracle's, bit for bit (a leaf slot missing from a tile's list would show)."""
    import random
    import pyoracle as po
    from raytracer_amd.scene import HostScene
    r = random.Random(7)
    objs = ['Sphere(center: (0, -100.5, 0), radius: 100, material: Shared("grey"))']
    for _ in range(40):
        objs.append('Sphere(center: (%.3f, %.3f, %.3f), radius: %.3f, material: Shared("grey"))'
                    % (r.uniform(-3, 3), r.uniform(-0.4, 1.5), r.uniform(-4, 2), r.uniform(0.05, 0.5)))
    objs.append('Translate(offset: (1.0, 0.2, -1.0), hittable: RotateY(angle: 30, hittable: '
                'Cuboid(size: (0.6, 0.8, 0.6), material: Shared("grey"))))')
    text = """Scene(
  image: (width: 96, height: 64, samples_per_pixel: 2, max_depth: 6),
  camera: (look_from: (%s, %s, %s), look_at: (0, 0, -1), vertical_fov: %s, aperture: %s, focus_distance: %s),
  background: (0.6, 0.7, 1.0),
  materials: { "grey": Lambertian(albedo: Solid(0.5, 0.5, 0.5)) },
  objects: [
    %s
  ]
)""" % (*look_from, fov, aperture, focus, ",\n    ".join(objs))
    scene = HostScene.load_ron_text(text)
    gpu_ctx.upload(scene)
    w, h = scene.width, scene.height
    img = gpu_ctx.render(gpu_ctx.params(w, h, 0, 8, scene.max_depth, seed=3), dtype=np.float64)
    ref = po.Oracle(scene.view_ptr).render(po.make_params(w, h, 0, 8, scene.max_depth, 3), po.MODE_ITERATIVE, threads=8)
    assert np.array_equal(img, ref)
    assert img.max() > 0


def test_headline_size_properties(gpu_ctx, golden_dir):
    """book_cover at 1920x1080, 1024 spp (the benchmark's workload): the sample range splits
    additively, a window is the same numbers as that region of the frame, and the window equals
    the CPU oracle bit for bit."""
    scene, g = load(golden_dir, "book_cover")
    w, h, spp = 1920, 1080, 1024
    scene.set_image(w, h)
    gpu_ctx.upload(scene)
    depth = scene.max_depth
    whole = gpu_ctx.render(gpu_ctx.params(w, h, 0, spp, depth, seed=1), dtype=np.float64)
    lo = gpu_ctx.render(gpu_ctx.params(w, h, 0, spp // 2, depth, seed=1), dtype=np.float64)
    hi = gpu_ctx.render(gpu_ctx.params(w, h, spp // 2, spp, depth, seed=1), dtype=np.float64)
    assert np.abs(whole - (lo + hi)).max() <= 1e-12 * np.abs(whole).max()
    assert np.isfinite(whole).all() and whole.min() >= 0.0
    window = (928, 522, 64, 36)
    part = gpu_ctx.render(gpu_ctx.params(w, h, 0, spp, depth, seed=1, window=window), dtype=np.float64)
    x0, y0, ww, wh = window
    assert np.array_equal(part, whole[y0:y0 + wh, x0:x0 + ww])
    o = po.Oracle(scene.view_ptr)
    ref = o.render(po.make_params(w, h, 0, spp, depth, 1, window=window), po.MODE_ITERATIVE, threads=8)
    assert np.array_equal(part, ref)


@pytest.mark.parametrize("name,w,h,spp,window", [("cornell", 1024, 1024, 64, (496, 600, 32, 24)),
                                                   ("book2", 1920, 1080, 32, (900, 500, 48, 27))])
def test_other_named_configurations_at_full_size(gpu_ctx, golden_dir, name, w, h, spp, window):
    """cornell at 1024x1024 (depth 500) and book2 at 1920x1080 (depth 32, shutter open): a window of
    the full-size frame is the same numbers as that region of the frame, sample ranges add up, and the
    window matches the CPU oracle (bit for bit without libm calls in the scene, 1e-9 with)."""
    scene, g = load(golden_dir, name)
    scene.set_image(w, h)
    gpu_ctx.upload(scene)
    depth = scene.max_depth
    whole = gpu_ctx.render(gpu_ctx.params(w, h, 0, spp, depth, seed=3), dtype=np.float64)
    lo = gpu_ctx.render(gpu_ctx.params(w, h, 0, spp // 4, depth, seed=3), dtype=np.float64)
    hi = gpu_ctx.render(gpu_ctx.params(w, h, spp // 4, spp, depth, seed=3), dtype=np.float64)
    assert np.abs(whole - (lo + hi)).max() <= 1e-12 * np.abs(whole).max()
    assert np.isfinite(whole).all() and whole.min() >= 0.0
    part = gpu_ctx.render(gpu_ctx.params(w, h, 0, spp, depth, seed=3, window=window), dtype=np.float64)
    x0, y0, ww, wh = window
    assert np.array_equal(part, whole[y0:y0 + wh, x0:x0 + ww])
    o = po.Oracle(scene.view_ptr)
    ref = o.render(po.make_params(w, h, 0, spp, depth, 3, window=window), po.MODE_ITERATIVE, threads=8)
    if name == "cornell":
        assert np.array_equal(part, ref)
    else:
        close = np.isclose(part, ref, rtol=1e-9, atol=1e-12)
        assert close.mean() > 0.995


def test_command_line_renderer_end_to_end(gpu_ctx, tmp_path):
    """scene.ron in, image.png and 'Done in' out (main.rs:49-115), against the library + host resolve."""
    import io
    import subprocess
    from PIL import Image
    from raytracer_amd import abi
    from raytracer_amd.scene import HostScene, resolve_srgb8
    import test_oracle_kat as kat
    text = kat.SCENE_HEAD + ('Sphere(center: (0, 0, 0), radius: 1, material: Shared("mirror")),'
                             'Sphere(center: (0, -101, 0), radius: 100, material: Shared("grey"))') + "\n  ]\n)"
    ron = tmp_path / "scene.ron"
    ron.write_text(text)
    exe = os.path.join(os.path.dirname(abi.device_lib()._name), "raytracer")
    out = tmp_path / "image.png"
    r = subprocess.run([exe, str(ron), "--spp", "5", "--seed", "7", "--out", str(out)], capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "Done in" in r.stdout + r.stderr
    assert r.stderr == ""  # (no progress line unless stderr is a terminal or --progress asks)
    bar = subprocess.run([exe, str(ron), "--spp", "5", "--seed", "7", "--out", str(tmp_path / "again.png"),
                          "--progress"], capture_output=True, timeout=120)
    shown = HostScene.load_ron_text(text)
    n_pixels = format(shown.width * shown.height, ",")
    last = bar.stderr.decode("utf-8", "replace").rstrip("\n").split("\r")[-1]
    assert bar.returncode == 0 and last.split(" ", 1)[1].startswith(
        f"{n_pixels}/{n_pixels} (100%) 00:00:0"), last  # main.rs:59-61
    assert (tmp_path / "again.png").read_bytes() == out.read_bytes()
    got = np.asarray(Image.open(io.BytesIO(out.read_bytes())).convert("RGB"))
    scene = HostScene.load_ron_text(text)
    gpu_ctx.upload(scene)
    sums = gpu_ctx.render(gpu_ctx.params(scene.width, scene.height, 0, 5, scene.max_depth, seed=7), dtype=np.float64)
    assert np.array_equal(got, resolve_srgb8(sums, 5))
    missing = subprocess.run([exe, str(tmp_path / "nope.ron")], capture_output=True, text=True, timeout=60)
    assert missing.returncode != 0 and "Error" in missing.stderr


@pytest.mark.parametrize("first_seed,textured", [(0, False), (30, False), (60, True)])
def test_random_scenes_match_the_oracle(gpu_ctx, first_seed, textured):
    """30 seeded random scenes per case against the reference-topology oracle: first hits exactly;
    radiance bit for bit where no libm call is involved (no media, no checker/Perlin), 1e-9 else."""
    from random_scenes import random_scene
    from raytracer_amd.scene import HostScene
    for seed in range(first_seed, first_seed + 30):
        text = random_scene(seed, textured=textured)
        scene = HostScene.load_ron_text(text)
        gpu_ctx.upload(scene)
        w, h = scene.width, scene.height
        o = po.Oracle(scene.view_ptr)
        ids, t, n = gpu_ctx.first_hit(w, h)
        rid, rt_, rn = o.first_hit(w, h)
        assert np.array_equal(ids, rid) and np.array_equal(t, rt_) and np.array_equal(n, rn), f"seed {seed}"
        img = gpu_ctx.render(gpu_ctx.params(w, h, 0, 2, scene.max_depth, seed=5), dtype=np.float64)
        ref = o.render(po.make_params(w, h, 0, 2, scene.max_depth, 5), po.MODE_ITERATIVE, threads=4)
        if textured:  # the pipelines agree exactly, libm or not
            from raytracer_amd import abi
            binned = gpu_ctx.render(gpu_ctx.params(w, h, 0, 2, scene.max_depth, seed=5, flags=abi.RT_FLAG_BINNED),
                                    dtype=np.float64)
            assert np.array_equal(img, binned), f"seed {seed}"
        if textured or "ConstantMedium" in text:
            close = np.isclose(img, ref, rtol=1e-9, atol=1e-12)
            assert close.mean() > 0.995, f"seed {seed}: {(~close).sum()} values differ"
        else:
            assert np.array_equal(img, ref), f"seed {seed}"


def test_degenerate_requests(gpu_ctx, golden_dir):
    """No objects, no samples, no depth: what main.rs:82-101 would produce for them."""
    from raytracer_amd.scene import HostScene
    import test_oracle_kat as kat
    empty = HostScene.load_ron_text(kat.SCENE_HEAD + "\n  ]\n)")
    gpu_ctx.upload(empty)
    w, h = 64, 64
    img = gpu_ctx.render(gpu_ctx.params(w, h, 0, 3, 5, seed=1), dtype=np.float64)
    assert np.array_equal(img, np.broadcast_to(3 * np.array([0.5, 0.7, 1.0]), (h, w, 3)))  # background x 3
    ids, t, n = gpu_ctx.first_hit(w, h)
    assert (ids == 0xFFFFFFFF).all() and not t.any() and not n.any()
    scene, g = load(golden_dir, "material_demo")
    gpu_ctx.upload(scene)
    w, h = scene.width, scene.height
    assert not gpu_ctx.render(gpu_ctx.params(w, h, 5, 5, 50, seed=1), dtype=np.float64).any()  # no samples
    assert not gpu_ctx.render(gpu_ctx.params(w, h, 0, 2, 0, seed=1), dtype=np.float64).any()   # ray_color(depth 0)
    one = gpu_ctx.render(gpu_ctx.params(w, h, 0, 2, 1, seed=1), dtype=np.float64)  # depth 1: emitted or sky only
    o = po.Oracle(scene.view_ptr)
    assert np.array_equal(one, o.render(po.make_params(w, h, 0, 2, 1, 1), po.MODE_ITERATIVE, threads=2))
    edge = (w - 3, h - 2, 3, 2)  # a window in the last rows and columns
    part = gpu_ctx.render(gpu_ctx.params(w, h, 0, 2, 50, seed=1, window=edge), dtype=np.float64)
    full = gpu_ctx.render(gpu_ctx.params(w, h, 0, 2, 50, seed=1), dtype=np.float64)
    assert np.array_equal(part, full[h - 2:, w - 3:])


def test_errors_are_reported(gpu_ctx, golden_dir):
    from raytracer_amd.abi import RtError
    scene, g = load(golden_dir, "material_demo")
    gpu_ctx.upload(scene)
    with pytest.raises(RtError):
        gpu_ctx.render(gpu_ctx.params(64, 36, 0, 1, 5, window=(60, 30, 10, 10)))
    with pytest.raises(RtError):
        gpu_ctx.render(gpu_ctx.params(1, 1, 0, 1, 5))
