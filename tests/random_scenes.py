"""Seeded random scenes in the reference's RON dialect, for differential tests.

Every primitive kind, wrappers nested both ways, cuboids with unequal x/z sizes (sides
outside their bounding boxes), media, all material kinds; only SolidColor textures unless
asked, so the device arithmetic is bit-comparable with the CPU oracle.
"""
import random


def _v(r, lo, hi):
    return "(%s, %s, %s)" % tuple(round(r.uniform(lo, hi), 3) for _ in range(3))


def _material(r, textured):
    k = r.randrange(6 if textured else 5)
    if k == 0:
        return "Lambertian(albedo: Solid(%s, %s, %s))" % tuple(round(r.uniform(0.1, 0.9), 3) for _ in range(3))
    if k == 1:
        return "Metal(albedo: %s, fuzziness: %s)" % (_v(r, 0.3, 0.95), round(r.choice([0.0, r.uniform(0, 0.6)]), 3))
    if k == 2:
        return "Dielectric(index_of_refraction: %s)" % round(r.uniform(1.1, 2.0), 3)
    if k == 3:
        return "DiffuseLight(color: Solid(%s, %s, %s))" % tuple(round(r.uniform(1, 6), 2) for _ in range(3))
    if k == 4:
        return 'Shared("grey")'
    return r.choice(["Lambertian(albedo: Checkerboard(Solid(1, 1, 1), Solid(0.1, 0.2, 0.1)))",
                     "Lambertian(albedo: Perlin)"])


def _base(r, textured):
    k = r.randrange(5)
    m = _material(r, textured)
    if k == 0:
        return "Sphere(center: %s, radius: %s, material: %s)" % (_v(r, -4, 4), round(r.uniform(0.2, 1.5), 3), m)
    if k == 1:
        c0 = [round(r.uniform(-4, 4), 3) for _ in range(3)]
        c1 = [round(c + r.uniform(-1, 1), 3) for c in c0]
        return ("MovingSphere(center: ((%s, %s, %s), (%s, %s, %s)), time: (0, 1), radius: %s, material: %s)"
                % (*c0, *c1, round(r.uniform(0.2, 1.0), 3), m))
    if k == 2:
        return ("AARect(center: %s, width: %s, height: %s, axis: %s, material: %s)"
                % (_v(r, -4, 4), round(r.uniform(0.5, 4), 3), round(r.uniform(0.5, 4), 3), r.choice("XYZ"), m))
    # cuboids: often with size.x != size.z
    size = [round(r.uniform(0.4, 3), 3) for _ in range(3)]
    if r.random() < 0.3:
        size[2] = size[0]
    return "Cuboid(size: (%s, %s, %s), material: %s)" % (*size, m)


def _wrapped(r, textured, depth=0):
    inner = _base(r, textured)
    while depth < 3 and r.random() < 0.45:
        if r.random() < 0.5:
            inner = "Translate(offset: %s, hittable: %s)" % (_v(r, -3, 3), inner)
        else:
            inner = "RotateY(angle: %s, hittable: %s)" % (round(r.uniform(-170, 170), 2), inner)
        depth += 1
    return inner


def random_scene(seed, n_objects=None, textured=False, width=48, height=32, wide_lens=False):
    """wide_lens: lenses up to several units across and focal planes anywhere from in front of the
    objects to far behind them (drawn from a generator of their own: the scene is the same otherwise)."""
    r = random.Random(seed)
    n = n_objects if n_objects is not None else r.choice([1, 3, 8, 20, 60, 300])
    objs = []
    for _ in range(n):
        if r.random() < 0.08:
            objs.append("ConstantMedium(boundary: %s, density: %s, texture: Solid(%s, %s, %s))"
                        % (_wrapped(r, False), round(r.uniform(0.1, 1.5), 3),
                           *(round(r.uniform(0.2, 1), 2) for _ in range(3))))
        else:
            objs.append(_wrapped(r, textured))
    cam = "look_from: %s, look_at: %s, vertical_fov: %s, aperture: %s" % (
        _v(r, -9, 9), _v(r, -1, 1), round(r.uniform(25, 70), 1), r.choice([0, 0, round(r.uniform(0, 0.3), 3)]))
    if r.random() < 0.5:
        cam += ", shutter_time: (0, 1)"
    if wide_lens:
        rl = random.Random(seed * 7919 + 13)
        cam = cam[:cam.rindex("aperture:")] + "aperture: %s, focus_distance: %s" % (
            round(rl.choice([rl.uniform(0.3, 1.0), rl.uniform(1.0, 4.0)]), 3), round(rl.uniform(0.4, 15.0), 3)) + (
            ", shutter_time: (0, 1)" if "shutter_time" in cam else "")
    return """Scene(
  image: (width: %d, height: %d, samples_per_pixel: 2, max_depth: %d),
  camera: (%s),
  background: %s,
  materials: { "grey": Lambertian(albedo: Solid(0.5, 0.5, 0.5)) },
  objects: [
    %s
  ]
)""" % (width, height, r.choice([1, 4, 12]), cam, _v(r, 0, 1), ",\n    ".join(objs))
