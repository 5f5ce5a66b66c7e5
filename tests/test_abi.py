"""The C-ABI libraries: exported symbols, loud failure without a device, BVH invariants."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from raytracer_amd import abi
from raytracer_amd.render import Context, build_bvh
from raytracer_amd.scene import HostScene
import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "mi355rt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_(?:host_)?[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    names = declared_functions()
    assert len(names) >= 25
    host, dev = abi.host_lib(), abi.device_lib()
    for n in names:
        lib = host if n.startswith("rt_host_") else dev
        assert hasattr(lib, n), f"{n} is declared in include/mi355rt.h but not exported"
    assert set(abi.HOST_SYMBOLS + abi.DEVICE_SYMBOLS) == set(names)


def test_struct_sizes_match_the_header():
    # sizes the frozen format and the device upload rely on
    assert C.sizeof(abi.Prim) == 104 and C.sizeof(abi.Op) == 32 and C.sizeof(abi.Object) == 96
    assert C.sizeof(abi.Material) == 40 and C.sizeof(abi.Texture) == 48
    assert C.sizeof(abi.Perlin) == 256 * 24 + 3 * 1024 and C.sizeof(abi.BvhNode) == 32
    assert C.sizeof(abi.RenderParams) == 48 and C.sizeof(abi.Camera) == 168


def test_no_device_means_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present")
    with pytest.raises(abi.RtError):
        Context(device=0)


def test_upload_rejects_malformed_scenes(golden_dir):
    s = HostScene.load_frozen(os.path.join(golden_dir, "material_demo.rtscene"))
    v = s.view
    saved = v.prims[0].material
    v.prims[0].material = 9999
    h = C.c_void_p()
    assert abi.device_lib().rt_bvh_build(s.view_ptr, C.byref(h)) == -1
    v.prims[0].material = saved
    assert abi.device_lib().rt_bvh_build(s.view_ptr, C.byref(h)) == 0
    abi.device_lib().rt_bvh_free(h)


@pytest.mark.parametrize("name", ["material_demo", "book_cover", "cornell", "book2", "motion_blur"])
def test_bvh_invariants(golden_dir, name):
    s = HostScene.load_frozen(os.path.join(golden_dir, name + ".rtscene"))
    nodes, items, depth = build_bvh(s)
    v = s.view
    assert depth <= 32
    # every primitive of every non-medium object, and every medium object, appears exactly once
    expect = set()
    for i in range(v.n_objects):
        o = v.objects[i]
        if o.is_medium:
            expect.add((i, 0xFFFFFFFF))
        else:
            expect.update((i, o.first_prim + k) for k in range(o.prim_count))
    got = [(int(it["object"]), int(it["prim"])) for it in items]
    assert len(got) == len(set(got)) and set(got) == expect
    # ranks follow the oracle's reference-topology preference order
    ranks = po.Oracle(s.view_ptr).ranks(v.n_objects)
    order = np.argsort(ranks)
    base, run = {}, 0
    for obj in order:
        base[int(obj)] = run
        run += v.objects[int(obj)].prim_count
    for it in items:
        o = v.objects[int(it["object"])]
        k = 0 if it["prim"] == 0xFFFFFFFF else int(it["prim"]) - o.first_prim
        assert int(it["rank"]) == base[int(it["object"])] + k
    # children lie inside their parent and are consecutive, numbering is breadth-first,
    # leaves tile the item array
    seen_items = np.zeros(len(items), dtype=bool)
    stack = [(0, 0)]
    max_seen = 0
    while stack:
        n, d = stack.pop()
        max_seen = max(max_seen, d)
        node = nodes[n]
        if node["b"] & 0x80000000:
            kids = int(node["b"] & 3) + 1
            assert 2 <= kids <= 4
            for c in range(int(node["a"]), int(node["a"]) + kids):
                assert c > n
                assert np.all(nodes[c]["lo"] >= node["lo"]) and np.all(nodes[c]["hi"] <= node["hi"])
                stack.append((c, d + 1))
        else:
            a, b = int(node["a"]), int(node["b"])
            assert 1 <= b <= 8 and not seen_items[a:a + b].any()
            seen_items[a:a + b] = True
    assert seen_items.all() and max_seen == depth


@pytest.mark.parametrize("name,size,spp", [("book_cover", (96, 54), 2), ("cornell", (40, 40), 4),
                                           ("book2", (64, 36), 1), ("medium_demo", (64, 36), 2),
                                           ("motion_blur", (80, 25), 2)])
def test_flat_traversal_equals_reference_topology(golden_dir, name, size, spp):
    """Conservative culling + rank tie-breaks give the reference's answer (checked on the CPU)."""
    s = HostScene.load_frozen(os.path.join(golden_dir, name + ".rtscene"))
    s.set_image(*size)
    nodes, items, _ = build_bvh(s)
    o = po.Oracle(s.view_ptr)
    p = po.make_params(size[0], size[1], 0, spp, s.max_depth, 3)
    ref = o.render(p, po.MODE_ITERATIVE, threads=4)
    flat = o.render(p, po.MODE_FLAT, bvh=(nodes, items), threads=4)
    assert np.array_equal(ref, flat)


OVERHANG_SCENE = """Scene(
  image: (width: 64, height: 48, samples_per_pixel: 2, max_depth: 6),
  camera: (look_from: (3, 9, 7), look_at: (0, 0, 0), vertical_fov: 55, aperture: 0),
  background: (0.5, 0.7, 1.0),
  materials: { "grey": Lambertian(albedo: Solid(0.5, 0.5, 0.5)),
               "red": Lambertian(albedo: Solid(0.7, 0.2, 0.2)) },
  objects: [
    Cuboid(size: (2, 1, 6), material: Shared("grey")),
    RotateY(angle: 30, hittable: Translate(offset: (0, -2, 0),
        hittable: Cuboid(size: (5, 1, 1), material: Shared("red")))),
    ConstantMedium(boundary: Translate(offset: (0, 3, 0), hittable: Cuboid(size: (1, 1, 4), material: Shared("grey"))),
                   density: 0.8, texture: Solid(1, 1, 1)),
  ]
)"""


def test_flat_traversal_answers_to_the_reference_boxes_for_overhanging_cuboid_sides():
    """Cuboid Y sides reach outside Cuboid::bounding_box when size.x != size.z (cuboid.rs:21-30,
    :45-47); whether the reference sees such a hit depends on its own hierarchy's boxes, which
    the flat traversal therefore consults for exactly those sides."""
    s = HostScene.load_ron_text(OVERHANG_SCENE)
    nodes, items, _ = build_bvh(s)
    o = po.Oracle(s.view_ptr)
    p = po.make_params(64, 48, 0, 2, 6, 3)
    ref = o.render(p, po.MODE_ITERATIVE, threads=4)
    flat = o.render(p, po.MODE_FLAT, bvh=(nodes, items), threads=4)
    assert np.array_equal(ref, flat)
    # the case is real: a ray straight down onto the overhang of the first cuboid's top side
    # (x = 2.5 is outside the box's x extent of +-1, inside the side's swapped extent of +-3)
    # is culled by the root list's box, one that drifts towards the box in x is not
    assert not o.world_hit([2.5, 10, 0], [0, -1, 0]).hit or o.world_hit([2.5, 10, 0], [0, -1, 0]).object != 0
    h = o.world_hit([2.5, 10, 0], [-0.01, -1, 0])
    assert h.hit and h.object == 0 and h.t == 9.5


@pytest.mark.parametrize("first_seed", [0, 40, 80])
def test_flat_traversal_equals_reference_topology_on_random_scenes(first_seed):
    """40 seeded random scenes per case: every primitive kind, nested wrappers, overhanging cuboid
    sides, media, all materials; 1 to 300 objects (one or two leaf lists in the reference)."""
    from random_scenes import random_scene
    for seed in range(first_seed, first_seed + 40):
        s = HostScene.load_ron_text(random_scene(seed))
        nodes, items, _ = build_bvh(s)
        o = po.Oracle(s.view_ptr)
        p = po.make_params(s.width, s.height, 0, 2, s.max_depth, 5)
        ref = o.render(p, po.MODE_ITERATIVE, threads=4)
        flat = o.render(p, po.MODE_FLAT, bvh=(nodes, items), threads=4)
        assert np.array_equal(ref, flat), f"seed {seed}"


def test_threaded_build_equals_the_serial_one(monkeypatch):
    """Ranges of 32768 items and more build their two sides on two threads: same records, same
    leaf order as the one-thread build (40 000 spheres: two levels of splits run threaded)."""
    from raytracer_amd.render import build_bvh
    from raytracer_amd.scene import HostScene
    scene = HostScene.generate_cover(100, scene_seed=3, width=64, height=36, spp=1, max_depth=5)
    monkeypatch.setenv("MI355RT_BUILD_THREADS", "1")
    nodes1, items1, depth1 = build_bvh(scene)
    monkeypatch.delenv("MI355RT_BUILD_THREADS")
    nodes2, items2, depth2 = build_bvh(scene)
    assert len(items1) == 40004 and depth1 == depth2
    assert np.array_equal(nodes1, nodes2) and np.array_equal(items1, items2)


def test_rust_mirror_follows_the_header():
    """rust/mi355rt-sys cannot be compiled here; its structs, sizes and symbols are held to the ctypes mirror."""
    import re
    text = open(os.path.join(ROOT, "rust", "mi355rt-sys", "src", "lib.rs")).read()
    mirror = {"rt_camera": abi.Camera, "rt_camera_desc": abi.CameraDesc, "rt_prim": abi.Prim, "rt_op": abi.Op,
              "rt_object": abi.Object, "rt_material": abi.Material, "rt_texture": abi.Texture,
              "rt_perlin": abi.Perlin, "rt_image": abi.Image, "rt_scene": abi.Scene,
              "rt_host_options": abi.HostOptions, "rt_config": abi.Config, "rt_window": abi.Window,
              "rt_render_params": abi.RenderParams, "rt_bvh_node": abi.BvhNode, "rt_bvh_item": abi.BvhItem,
              "rt_group_report": abi.GroupReport}
    for name, cls in mirror.items():
        body = re.search(r"pub struct %s \{(.*?)\n\}" % name, text, re.S).group(1)
        fields = re.findall(r"pub (\w+):", body)
        assert fields == [f[0] for f in cls._fields_], name
        size = re.search(r"size_of::<%s>\(\) == ([^)]+)\)" % name, text).group(1)
        assert eval(size) == C.sizeof(cls), name
    fns = set(re.findall(r"pub fn (rt_\w+)\(", text))
    assert fns == set(abi.HOST_SYMBOLS + abi.DEVICE_SYMBOLS)


def levelwise_reference_order(centres):
    """The reference's resolution order by the scheme the device uses (bvh_device.hip): level l
    sorts every range of 256 objects or more stably by box centre along axis l mod 3 (NaN last,
    -0 = +0) and halves it; the upper half of every split comes first in the order."""
    n = centres.shape[1]
    perm = np.arange(n)
    ranges = [(0, n)]
    level = 0
    while any(ln >= 256 for _, ln in ranges):
        key = centres[level % 3][perm]
        nxt = []
        for lo, ln in ranges:
            if ln < 256:
                nxt.append((lo, ln))
                continue
            k = key[lo:lo + ln] + 0.0  # (-0 + 0 = +0)
            order = np.lexsort((np.arange(ln), np.where(np.isnan(k), np.inf, k), np.isnan(k)))
            perm[lo:lo + ln] = perm[lo:lo + ln][order]
            half = ln // 2
            nxt += [(lo, half), (lo + half, ln - half)]
        ranges = nxt
        level += 1

    def emit(lo, ln):  # upper half first, all the way down
        if ln < 256:
            return list(perm[lo:lo + ln])
        half = ln // 2
        return emit(lo + half, ln - half) + emit(lo, half)
    return np.array(emit(0, n))


@pytest.mark.parametrize("half", [9, 20])
def test_levelwise_sorts_give_the_reference_order(half):
    """The device makes the tie-break order by one stable sort per level of the median-split
    recursion instead of recursing (bvh.rs:33-77).  Same scheme in numpy against the ranks the host's
    recursive sorts give: 328 spheres (one split) and 1604 (three levels; every y centre of the
    small spheres equal, so the y level is all ties and only stability orders it)."""
    from raytracer_amd.render import build_bvh
    from raytracer_amd.scene import HostScene
    scene = HostScene.generate_cover(half, scene_seed=7, width=64, height=36, spp=1, max_depth=5)
    v = scene.view
    n = v.n_objects
    c = np.array([[v.prims[i].center[a] for i in range(n)] for a in range(3)])
    r = np.array([v.prims[i].radius for i in range(n)])
    centres = ((c + -r) + (c + r)) / 2.0  # sphere.rs:67-72, then the sort key of bvh.rs:47-53
    _, items, _ = build_bvh(scene)
    host_order = np.empty(n, dtype=np.int64)
    host_order[items["rank"]] = items["object"]
    assert np.array_equal(levelwise_reference_order(centres), host_order)
