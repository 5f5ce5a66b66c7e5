"""Known answers for the CPU oracle.

The reference's own tests pin only Vec3::project and Aabb::surrounding
(src/geom/vec3.rs:299-309, src/geom/aabb.rs:87-104); both are repeated here on
the oracle's restatement.  Everything else is hand-derived from the reference's
source and written down next to the lines it follows: parity for the render
path is otherwise UNPINNED (no reference binary can be built in this image).
"""
import math
import os

import numpy as np
import pytest

from raytracer_amd.scene import HostScene
import pyoracle as po

SCENE_HEAD = """Scene(
  image: (width: 64, height: 64, samples_per_pixel: 1, max_depth: 5),
  camera: (look_from: (0, 0, 10), look_at: (0, 0, 0), vertical_fov: 20, aperture: 0),
  background: (0.5, 0.7, 1.0),
  materials: {
    "grey": Lambertian(albedo: Solid(0.5, 0.5, 0.5)),
    "glass": Dielectric(index_of_refraction: 1.5),
    "mirror": Metal(albedo: (0.9, 0.8, 0.7), fuzziness: 0.0),
    "checker": Lambertian(albedo: Checkerboard(Solid(1, 1, 1), Solid(0, 0, 0))),
  },
  objects: [
"""


def scene_of(objects, shutter=None):
    head = SCENE_HEAD
    if shutter is not None:
        head = head.replace("aperture: 0)", f"aperture: 0, shutter_time: ({shutter[0]}, {shutter[1]}))")
    return HostScene.load_ron_text(head + objects + "\n  ]\n)")


def oracle_of(objects, shutter=None):
    s = scene_of(objects, shutter)
    o = po.Oracle(s.view_ptr)
    o._scene = s  # keep the arrays alive
    return o


# ---------------------------------------------------------------- reference's own tests
def test_reference_unit_test_project():
    assert po.vec3_project([1, 2, 3], [2, 0, 0]) == [1, 0, 0]
    assert po.vec3_project([1, 2, 3], [0, 2, 0]) == [0, 2, 0]
    assert po.vec3_project([1, 2, 3], [0, 0, 2]) == [0, 0, 3]


def test_reference_unit_test_surrounding():
    out = po.aabb_surrounding([[1, 2, 3, 4, 5, 6], [7, 8, 9, 10, 11, 12]])
    assert out == [1, 2, 3, 10, 11, 12]
    assert po.aabb_surrounding(np.zeros((0, 6))) is None


# ---------------------------------------------------------------- Philox4x32-10 (Random123 vectors)
def test_philox_known_answers():
    assert po.philox([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert po.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6,
                                                             0x6D5451FD]
    assert po.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344],
                     [0xA4093822, 0x299F31D0]) == [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


# ---------------------------------------------------------------- primitives
def test_sphere_front_hit_and_uv():
    o = oracle_of('Sphere(center: (0, 0, -1), radius: 0.5, material: Shared("grey"))')
    h = o.world_hit([0, 0, 0], [0, 0, -1])
    assert h.hit and h.t == 0.5 and list(h.p) == [0, 0, -0.5] and list(h.normal) == [0, 0, 1]
    assert h.front_face == 1 and h.object == 0
    # get_uv((0,0,1)): theta = acos(0), phi = atan2(-1, 0) + pi (sphere.rs:20-24)
    assert h.v == pytest.approx(0.5, abs=1e-15) and h.u == pytest.approx(0.25, abs=1e-15)


def test_sphere_from_inside_takes_second_root():
    o = oracle_of('Sphere(center: (0, 0, -1), radius: 0.5, material: Shared("grey"))')
    h = o.world_hit([0, 0, -1], [0, 0, -1])
    assert h.hit and h.t == 0.5 and h.front_face == 0 and list(h.normal) == [0, 0, 1]


def test_sphere_range_is_half_open():
    o = oracle_of('Sphere(center: (0, 0, -1), radius: 0.5, material: Shared("grey"))')
    assert o.world_hit([0, 0, 0], [0, 0, -1], t_start=0.5, t_end=2.0).t == 0.5  # start included
    assert not o.world_hit([0, 0, 0], [0, 0, -1], t_start=0.001, t_end=0.5).hit  # end excluded
    assert o.world_hit([0, 0, 0], [0, 0, -1], t_start=0.6, t_end=2.0).t == 1.5  # falls to the far root


def test_rect_z():
    o = oracle_of('AARect(center: (0, 0, -2), width: 2, height: 2, axis: Z, material: Shared("grey"))')
    h = o.world_hit([0, 0, 0], [0, 0, -1])
    assert h.hit and h.t == 2 and list(h.normal) == [0, 0, 1] and h.front_face == 1
    assert h.u == 0.5 and h.v == 0.5
    # the edge itself is inside (RangeInclusive, rect.rs:50)
    assert o.world_hit([1, 0, 0], [0, 0, -1]).hit
    assert not o.world_hit([1.0000001, 0, 0], [0, 0, -1]).hit


def test_rect_y_width_runs_along_x():
    # Axis::Y => offset (w2, thickness, h2): X +- w/2, Z +- h/2 (rect.rs:28-32)
    o = oracle_of('AARect(center: (0, 0, 0), width: 4, height: 2, axis: Y, material: Shared("grey"))')
    assert o.world_hit([1.9, 1, 0.9], [0, -1, 0]).hit
    assert not o.world_hit([2.1, 1, 0], [0, -1, 0]).hit
    assert not o.world_hit([0, 1, 1.1], [0, -1, 0]).hit


def test_cuboid_y_faces_swap_extents():
    # width = size[axis.next()], height = size[axis.prev()] (cuboid.rs:26-27) + the Y-rect rule:
    # a (2,4,6) cuboid has Y faces with X in [-3,3] and Z in [-1,1]
    # (a second object widens the leaf's bounding box: Cuboid::bounding_box is center +- half and
    # does not cover the overhang, cuboid.rs:45-47)
    o = oracle_of('Cuboid(size: (2, 4, 6), material: Shared("grey")),'
                  'Sphere(center: (0, -100, 0), radius: 50, material: Shared("grey"))')
    h = o.world_hit([2.5, 10, 0], [0, -1, 0])  # outside the box's x extent, still hits the top face
    assert h.hit and h.t == 8 and h.object == 0
    h = o.world_hit([0, 10, 2.0], [0, -1, 0])  # inside the box's z extent, misses the top face
    assert not (h.hit and h.t == 8)
    # alone in the scene, the leaf box IS the cuboid's box and culls the overhang for this ray
    alone = oracle_of('Cuboid(size: (2, 4, 6), material: Shared("grey"))')
    assert not alone.world_hit([2.5, 10, 0], [0, -1, 0]).hit


def test_moving_sphere_moves_backwards():
    # time_portion = (t0 - time) / (t1 - t0) (moving_sphere.rs:21)
    assert po.moving_center([0, 0, 0], [1, 0, 0], 0.0, 1.0, 1.0) == [-1, 0, 0]
    assert po.moving_center([0, 0, 0], [1, 0, 0], 0.0, 1.0, 0.0) == [0, 0, 0]
    # the world's boxes cover the camera's shutter range (loader.rs:86), so give it one
    o = oracle_of('MovingSphere(center: ((0, 0, 0), (1, 0, 0)), time: (0, 1), radius: 0.5, '
                  'material: Shared("grey"))', shutter=(0, 1))
    h = o.world_hit([-1, 0, 5], [0, 0, -1], time=1.0)
    assert h.hit and h.t == 4.5


def test_rotate_y_keeps_the_x_for_z_slip():
    # theta = 90 deg: object-space hit p=(1,y,0), n=(1,0,0) comes back as p=(c+s, y, -s+c)
    # (instance.rs:67-73); with sin=1, cos~6e-17 that is (1, y, -1) and n=(1,0,-1) up to 1e-16
    o = oracle_of('RotateY(angle: 90, hittable: AARect(center: (1, 0, 0), width: 2, height: 2, '
                  'axis: X, material: Shared("grey")))')
    # object-space ray (5,0.25,0)+t(-1,0,0) is world-space origin (0,0.25,-5)... use the inverse map:
    # o'.x = c*ox - s*oz = 5  =>  oz = -5 ; o'.z = s*ox + c*oz = 0  =>  ox = 0 (to 1e-16)
    h = o.world_hit([0, 0.25, -5], [0, 0, 1])
    assert h.hit and h.t == pytest.approx(4.0, abs=1e-12)
    assert list(h.p) == pytest.approx([1, 0.25, -1], abs=1e-12)
    assert list(h.normal) == pytest.approx([1, 0, -1], abs=1e-12)  # not unit length


def test_translate_reports_front_face_for_inside_hits():
    # set_face_normal on the already flipped normal (instance.rs:16)
    o = oracle_of('Translate(offset: (0, 0, -1), hittable: Sphere(center: (0, 0, 0), radius: 0.5, '
                  'material: Shared("glass")))')
    h = o.world_hit([0, 0, -1], [0, 0, -1])
    assert h.hit and h.t == 0.5 and h.front_face == 1
    assert list(h.p) == [0, 0, -1.5] and list(h.normal) == [0, 0, 1]


def test_constant_medium_uses_log10():
    o = oracle_of('ConstantMedium(boundary: Sphere(center: (0, 0, 0), radius: 100, '
                  'material: Shared("grey")), density: 0.2, texture: Solid(1, 1, 1))')
    seed, pixel, sample, bounce = 5, 3, 2, 1
    blk = po.philox([pixel, sample, bounce, (5 << 24) | 0], [seed, 0])  # DRAW_MEDIUM, object 0
    u = ((blk[1] << 32 | blk[0]) >> 11) / 2.0 ** 53
    h = o.world_hit([0, 0, 0], [0, 0, -2], seed=seed, pixel=pixel, sample=sample, bounce=bounce,
                    with_media=True)
    dist = (-1 / 0.2) * math.log10(u)  # constant_medium.rs:54
    assert h.hit and h.t == pytest.approx(0.001 + dist / 2.0, rel=1e-12)
    # d.x == 0: `d . (1,0,0) < 0` is false, so the record is a back face and the normal flips
    assert list(h.normal) == [-1, 0, 0] and h.front_face == 0 and h.u == 0 and h.v == 0
    assert not o.world_hit([0, 0, 0], [0, 0, -2]).hit  # media are left out without a draw context


def test_list_returns_first_of_equal_minima_and_bvh_prefers_right():
    two = ('Sphere(center: (0, 0, -1), radius: 0.5, material: Shared("grey")),'
           'Sphere(center: (0, 0, -1), radius: 0.5, material: Shared("glass"))')
    assert oracle_of(two).world_hit([0, 0, 0], [0, 0, -1]).object == 0  # hittable_list.rs:31
    # 256 objects make an interior node (bvh.rs:9,38); equal keys keep file order, the upper half
    # goes right, and equal t resolves to the right child (bvh.rs:93-97)
    many = ",".join(['Sphere(center: (0, 0, -1), radius: 0.5, material: Shared("grey"))'] * 256)
    o = oracle_of(many)
    assert o.bvh_shape() == (3, 2, 128)
    assert o.world_hit([0, 0, 0], [0, 0, -1]).object == 128
    ranks = o.ranks(256)
    assert ranks[128] == 0 and ranks[0] == 128


def test_aabb_intersect_is_per_axis_only():
    # misses the unit box diagonally, yet each axis interval alone overlaps (0.001, inf)
    assert po.aabb_intersect([0, 0, 0], [1, 1, 1], [-1, 2.5, 0.5], [1, -1, 0.0], 0.001, math.inf)
    assert not po.aabb_intersect([0, 0, 0], [1, 1, 1], [-1, 2.5, 0.5], [-1, -1, 0.0], 0.001, math.inf)


# ---------------------------------------------------------------- materials and textures
def test_reflectance_and_refraction():
    assert po.reflectance(1.0, 1.5) == pytest.approx(0.04, abs=1e-15)
    assert po.reflectance(0.0, 1.5) == 1.0
    assert po.refract([0, 0, -1], [0, 0, 1], 1 / 1.5) == pytest.approx([0, 0, -1], abs=1e-15)
    r = po.reflect([1 / math.sqrt(2), -1 / math.sqrt(2), 0], [0, 1, 0])
    assert r == pytest.approx([1 / math.sqrt(2), 1 / math.sqrt(2), 0], abs=1e-15)


def test_metal_with_zero_fuzz_reflects_exactly():
    o = oracle_of('Sphere(center: (0, 0, 0), radius: 1, material: Shared("mirror"))')
    h = o.world_hit([0, 5, 0], [0, -1, 0])
    s = o.scatter(h.material, [0, 5, 0], [0, -1, 0], 0.0, h)
    assert s["scattered"] and s["direction"] == [0, 1, 0] and s["attenuation"] == [0.9, 0.8, 0.7]
    assert s["emitted"] == [0, 0, 0]


def test_checkerboard_sign_rule():
    s = scene_of('Sphere(center: (0, 0, 0), radius: 1, material: Shared("checker"))')
    o = po.Oracle(s.view_ptr)
    tex = s.view.materials[s.view.prims[0].material].texture
    assert o.texture_value(tex, 0, 0, [0.1, 0.1, 0.1]) == [1, 1, 1]    # sin(1)^3 > 0: even
    assert o.texture_value(tex, 0, 0, [0.1, 0.1, -0.1]) == [0, 0, 0]   # negative: odd


def test_image_texel_indexing(golden_dir):
    s = HostScene.load_frozen(os.path.join(golden_dir, "globe.rtscene"))
    v = s.view
    assert (v.images[0].width, v.images[0].height) == (1024, 512)
    o = po.Oracle(s.view_ptr)
    tex = [i for i in range(v.n_textures) if v.textures[i].kind == 3][0]
    texels = np.ctypeslib.as_array(v.texels, shape=(v.n_texel_bytes,)).reshape(512, 1024, 3)

    def expect(i, j):
        return [float(c) * (1.0 / 255.0) for c in texels[j, i]]
    # i = (u*w) as u32, j = h - (v*h) as u32, clamped (image.rs:30-31)
    assert o.texture_value(tex, 0.5, 0.5, [0, 0, 0]) == expect(512, 256)
    assert o.texture_value(tex, 0.5, 0.0, [0, 0, 0]) == expect(512, 511)
    assert o.texture_value(tex, 0.5, 1.0, [0, 0, 0]) == expect(512, 0)
    assert o.texture_value(tex, 1.5, 0.5, [0, 0, 0]) == expect(1023, 256)


def test_into_srgb8():
    assert po.into_srgb8([64, 32, 0], 64) == [255, 128, 0]
    assert po.into_srgb8([0.999 * 64, 0.5 * 64, -1], 64) == [255, 128, 0]
    assert po.into_srgb8([float("nan"), 1e9, 0.25 * 64], 64) == [0, 255, 64]


def test_cornell_camera_constants(golden_dir):
    s = HostScene.load_frozen(os.path.join(golden_dir, "cornell.rtscene"))
    cam = po.camera_build(s.camera_desc, 500, 500)
    vh = 2 * math.tan(math.radians(40) / 2)
    assert list(cam.u) == [-1, 0, 0] and list(cam.v) == [0, 1, 0]
    assert list(cam.horizontal) == pytest.approx([-800 * vh, 0, 0], rel=1e-15)
    assert list(cam.lower_left) == pytest.approx([277.5 + 400 * vh, 277.5 - 400 * vh, 0], rel=1e-14)
    assert cam.lens_radius == 0 and cam.time0 == 0 and cam.time1 == 0


# ---------------------------------------------------------------- the oracle against its fixtures
@pytest.mark.parametrize("name", ["material_demo", "book_cover", "cornell", "medium_demo",
                                  "motion_blur", "refraction", "checker_mirror", "emissive"])
def test_oracle_reproduces_golden_fixtures(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".golden.npz"))
    s = HostScene.load_frozen(os.path.join(golden_dir, name + ".rtscene"))
    w, h = int(g["width"]), int(g["height"])
    s.set_image(w, h)
    o = po.Oracle(s.view_ptr)
    ids, t, n = o.first_hit(w, h)
    assert np.array_equal(ids, g["first_id"]) and np.array_equal(t, g["first_t"])
    assert np.array_equal(n, g["first_normal"])
    window = tuple(int(x) for x in g["window"])
    p = po.make_params(w, h, 0, int(g["spp"]), int(g["max_depth"]), int(g["seed"]), window=window)
    img = o.render(p, po.MODE_REFERENCE, threads=4)
    assert np.array_equal(img, g["render_sum"])
    # the iterative estimator differs from the literal recursion only by association
    it = o.render(p, po.MODE_ITERATIVE, threads=4)
    assert np.abs(it - img).max() <= 1e-12 * max(1.0, np.abs(img).max())


def test_reference_bvh_shapes(golden_dir):
    # leaves hold fewer than 256 objects (bvh.rs:9,38): material_demo/cornell are one leaf,
    # book_cover (488 spheres) 2 x 244, book2 (10,370 objects) 64 leaves
    shapes = {"material_demo": (1, 1, 4), "cornell": (1, 1, 8), "book_cover": (3, 2, 244),
              "book2": (127, 64, 163)}
    for name, shape in shapes.items():
        s = HostScene.load_frozen(os.path.join(golden_dir, name + ".rtscene"))
        assert po.Oracle(s.view_ptr).bvh_shape() == shape
