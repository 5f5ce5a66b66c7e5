"""A second, independent restatement of the reference's scene realisation, in plain Python,
for scenes without load-time randomness.  TEST INFRASTRUCTURE: it pins host/loader.cpp, whose
output both the CPU oracle and the device consume, from outside that code.

Follows src/scene/desc.rs:12-177 (the description types and Value::eval, :124-147),
src/scene/loader.rs:47-93 (load), :95-189 (realize_hittable), :192-311 (realize_pattern),
:313-377 (materials, textures), src/hittable/cuboid.rs:18-37 (the six sides),
src/hittable/instance.rs:36-48 (RotateY::new), src/hittable/constant_medium.rs:17-24 and
src/camera.rs:52-86 (Builder::done).  Python floats are IEEE doubles and the arithmetic is
written in the reference's operation order, so every number is expected bit for bit.

The output is a list of top-level objects in the loader's order, each a dict:
  wrappers   outermost first: ("translate", (x, y, z)) | ("rotate_y", sin, cos) | ("medium", -1/density, texture)
  prims      [("sphere", c, r, material) | ("moving_sphere", c0, c1, t0, t1, r, material) |
              ("rect", axis, centre, width, height, material)]  (a Cuboid: its six rects, cuboid = (centre, half))
with materials and textures as nested tuples of their contents (no indices).
"""
import math


# ------------------------------------------------------------------ RON (the subset the scenes use)
class _Reader:
    def __init__(self, text):
        self.s, self.i = text, 0

    def ws(self):
        s = self.s
        while self.i < len(s):
            if s[self.i].isspace():
                self.i += 1
            elif s.startswith("//", self.i):
                while self.i < len(s) and s[self.i] != "\n":
                    self.i += 1
            else:
                break

    def peek(self):
        self.ws()
        return self.s[self.i] if self.i < len(self.s) else ""

    def eat(self, c):
        self.ws()
        assert self.s.startswith(c, self.i), f"expected {c!r} at {self.i}: {self.s[self.i:self.i + 30]!r}"
        self.i += len(c)

    def value(self):
        c = self.peek()
        if c == '"':
            j = self.s.index('"', self.i + 1)
            out = self.s[self.i + 1:j]
            self.i = j + 1
            return ("str", out)
        if c == "(":
            return self.parens(None)
        if c == "[":
            self.eat("[")
            items = []
            while self.peek() != "]":
                items.append(self.value())
                if self.peek() == ",":
                    self.eat(",")
            self.eat("]")
            return ("list", items)
        if c == "{":
            self.eat("{")
            items = []
            while self.peek() != "}":
                k = self.value()
                self.eat(":")
                items.append((k[1], self.value()))
                if self.peek() == ",":
                    self.eat(",")
            self.eat("}")
            return ("map", items)
        if c.isdigit() or c in "-+.":
            j = self.i + 1
            while j < len(self.s) and (self.s[j].isalnum() or self.s[j] in "._+-") and not (
                    self.s[j] in "+-" and self.s[j - 1] not in "eE"):
                j += 1
            tok = self.s[self.i:j].replace("_", "")
            self.i = j
            return ("num", float(tok))
        j = self.i
        while j < len(self.s) and (self.s[j].isalnum() or self.s[j] == "_"):
            j += 1
        name = self.s[self.i:j]
        assert name, f"unexpected {c!r} at {self.i}"
        self.i = j
        if self.peek() == "(":
            return self.parens(name)
        return ("unit", name)

    def parens(self, name):
        """`(a, b)` tuple, `(k: v, ...)` struct, each optionally named."""
        self.eat("(")
        fields, items = {}, []
        while self.peek() != ")":
            save = self.i
            v = self.value()
            if v[0] == "unit" and self.peek() == ":":
                self.eat(":")
                fields[v[1]] = self.value()
            else:
                self.i = save
                items.append(self.value())
            if self.peek() == ",":
                self.eat(",")
        self.eat(")")
        if fields:
            return ("struct", name, fields)
        return ("tuple", name, items)


def parse_ron(text):
    r = _Reader(text)
    v = r.value()
    r.ws()
    assert r.i == len(text), "trailing text"
    return v


# ------------------------------------------------------------------ Vec3, in the reference's operation order
def vadd(a, b):
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


def vneg(a):
    return (-a[0], -a[1], -a[2])


def vsub(a, b):  # vec3.rs:163-168: self + (-rhs)
    return vadd(a, vneg(b))


def vscale(a, s):
    return (a[0] * s, a[1] * s, a[2] * s)


def vdiv(a, s):  # vec3.rs:214-219: self * rhs.recip()
    return vscale(a, 1.0 / s)


def vdot(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def vlen(a):
    return math.sqrt(vdot(a, a))


def vunit(a):
    return vdiv(a, vlen(a))


def vcross(a, b):  # vec3.rs:74-80
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])


def vproject(a, onto):  # vec3.rs:135-138
    u = vunit(onto)
    return vscale(u, vdot(a, u))


AXES = {"X": 0, "Y": 1, "Z": 2}
UNIT = [(1.0, 0.0, 0.0), (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)]


# ------------------------------------------------------------------ realisation
class Loader:
    def __init__(self, text):
        root = parse_ron(text)
        assert root[0] == "struct"
        self.desc = root[2]
        self.vars = {}
        self.materials = {}
        for key, m in self.desc["materials"][1]:  # (a HashMap in the reference: the order carries no meaning)
            self.materials[key] = self.material(m)
        self.objects = []
        for o in self.desc["objects"][1]:
            self.hittable(o, self.objects)

    # desc.rs:124-147
    def eval(self, v):
        if v[0] == "num":
            return v[1]
        if v[0] == "str":
            return float(self.vars[v[1]])
        assert v[0] == "tuple" and len(v[2]) == 3, v
        op, a, b = v[2]
        a, b = self.eval(a), self.eval(b)
        name = op[1].lower()
        if name == "add":
            return a + b
        if name == "mult":
            return a * b
        raise ValueError("load-time randomness is outside this restatement")

    def vec3(self, v):
        assert v[0] == "tuple" and len(v[2]) == 3, v
        return tuple(self.eval(x) for x in v[2])

    # loader.rs:356-377
    def texture(self, t):
        if t[0] == "tuple" and t[1] == "Solid":
            return ("solid",) + tuple(x[1] for x in t[2])
        if t[0] == "tuple" and t[1] == "Checkerboard":
            return ("checker", self.texture(t[2][0]), self.texture(t[2][1]))
        raise ValueError(f"texture {t[1]} is outside this restatement")

    # loader.rs:313-354
    def material(self, m):
        kind = m[1]
        if kind == "Shared":
            return self.materials[m[2][0][1]]
        f = m[2]
        if kind == "Lambertian":
            return ("lambertian", self.texture(f["albedo"]))
        if kind == "Metal":
            return ("metal", self.vec3(f["albedo"]), self.eval(f["fuzziness"]))
        if kind == "Dielectric":
            return ("dielectric", self.eval(f["index_of_refraction"]))
        if kind == "DiffuseLight":
            return ("diffuse_light", self.texture(f["color"]))
        raise ValueError(f"material {kind} is outside this restatement")

    # loader.rs:95-189 (and :192-311, which repeats it per pattern value)
    def hittable(self, h, out):
        kind, f = h[1], h[2]
        if kind == "Sphere":
            out.append({"wrappers": [], "cuboid": None, "prims": [
                ("sphere", self.vec3(f["center"]), self.eval(f["radius"]), self.material(f["material"]))]})
        elif kind == "MovingSphere":
            c, t = f["center"][2], f["time"][2]
            out.append({"wrappers": [], "cuboid": None, "prims": [
                ("moving_sphere", self.vec3(c[0]), self.vec3(c[1]), self.eval(t[0]), self.eval(t[1]),
                 self.eval(f["radius"]), self.material(f["material"]))]})
        elif kind == "AARect":
            out.append({"wrappers": [], "cuboid": None, "prims": [
                ("rect", AXES[f["axis"][1]], self.vec3(f["center"]), self.eval(f["width"]),
                 self.eval(f["height"]), self.material(f["material"]))]})
        elif kind == "Cuboid":  # cuboid.rs:18-37
            center = self.vec3(f["center"]) if "center" in f else (0.0, 0.0, 0.0)
            size = self.vec3(f["size"])
            mat = self.material(f["material"])
            half = vdiv(size, 2.0)
            sides = []
            for axis in range(3):  # iproduct!([X, Y, Z], [-1.0, 1.0])
                for mult in (-1.0, 1.0):
                    sides.append(("rect", axis, vadd(center, vscale(vproject(half, UNIT[axis]), mult)),
                                  size[(axis + 1) % 3], size[(axis + 2) % 3], mat))  # width next(), height prev()
            out.append({"wrappers": [], "cuboid": (center, half), "prims": sides})
        elif kind == "Pattern":
            r = [int(x[1]) for x in f["range"][1]]
            rng = (range(0, r[0]) if len(r) == 1 else range(r[0], r[1]) if len(r) == 2
                   else range(r[0], r[1], abs(r[2])))
            for val in rng:
                self.vars[f["var"][1]] = val
                self.hittable(f["object"], out)
        elif kind in ("Translate", "RotateY", "ConstantMedium"):
            if kind == "Translate":
                wrap = ("translate", self.vec3(f["offset"]))
                inner_desc = f["hittable"]
            elif kind == "RotateY":
                theta = self.eval(f["angle"]) * (math.pi / 180.0)  # f64::to_radians
                wrap = ("rotate_y", math.sin(theta), math.cos(theta))  # instance.rs:38
                inner_desc = f["hittable"]
            else:
                texture = self.texture(f["texture"])  # evaluated before the density, loader.rs:177-178
                density = self.eval(f["density"])
                wrap = ("medium", -(1.0 / density), texture)  # constant_medium.rs:20
                inner_desc = f["boundary"]
            inner = []
            self.hittable(inner_desc, inner)
            for o in inner:  # each inner object is wrapped on its own, loader.rs:155-160
                out.append({"wrappers": [wrap] + o["wrappers"], "cuboid": o["cuboid"], "prims": o["prims"]})
        else:
            raise ValueError(kind)

    # camera.rs:52-86 with the loader's arguments, loader.rs:64-83
    def camera(self, width=None, height=None):
        c = self.desc["camera"][2]
        im = self.desc["image"][2]
        width = width or int(im["width"][1])
        height = height or int(im["height"][1])
        pt = lambda v: tuple(x[1] for x in v[2])  # noqa: E731  (plain f64 triples, no expressions)
        look_from = pt(c["look_from"])
        look_at = pt(c["look_at"]) if "look_at" in c else (0.0, 0.0, 0.0)
        look_vector = vsub(look_from, look_at)
        focus = c["focus_distance"][1] if "focus_distance" in c else vlen(vsub(look_at, look_from))
        aspect = float(width) / float(height)
        theta = c["vertical_fov"][1] * (math.pi / 180.0)
        h = math.tan(theta / 2.0)
        vh = 2.0 * h
        vw = aspect * vh
        v_up = pt(c["v_up"]) if "v_up" in c else (0.0, 1.0, 0.0)
        w = vunit(look_vector)
        u = vunit(vcross(v_up, w))
        v = vcross(w, u)
        horizontal = vscale(u, focus * vw)   # (focus_dist * viewport_width) * u
        vertical = vscale(v, focus * vh)
        llc = vsub(vsub(vsub(look_from, vdiv(horizontal, 2.0)), vdiv(vertical, 2.0)), vscale(w, focus))
        shutter = tuple(x[1] for x in c["shutter_time"][2]) if "shutter_time" in c else (0.0, 0.0)
        return {"origin": look_from, "lower_left": llc, "horizontal": horizontal, "vertical": vertical,
                "u": u, "v": v, "lens_radius": c["aperture"][1] / 2.0, "time0": shutter[0], "time1": shutter[1]}

    def image(self):
        im = self.desc["image"][2]
        return tuple(int(im[k][1]) for k in ("width", "height", "samples_per_pixel", "max_depth"))

    def background(self):
        return self.vec3(self.desc["background"]) if "background" in self.desc else (0.0, 0.0, 0.0)
