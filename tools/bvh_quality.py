"""Traversal work per ray of the hierarchy the builder makes, measured on the CPU by the oracle's
instrumented flat traversal (no GPU): node visits, box tests, primitive tests, roots.

usage: python tools/bvh_quality.py [scene ...]      (builder knobs come from the environment)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402
from raytracer_amd import abi  # noqa: E402
from raytracer_amd.render import build_bvh  # noqa: E402
from raytracer_amd.scene import HostScene  # noqa: E402


def measure(scene, w=160, h=90, spp=2):
    scene.set_image(w, h)
    nodes, items, depth = build_bvh(scene)
    o = po.Oracle(scene.view_ptr)
    _, st = o.render(po.make_params(w, h, 0, spp, scene.max_depth, 1), po.MODE_FLAT, bvh=(nodes, items),
                     threads=8, stats=True)
    g = lambda n: float(st[abi.STAT_NAMES.index(n)])  # noqa: E731
    rays = g("rays")
    prim = g("sphere_tests") + g("msphere_tests") + g("rect_tests") + g("medium_tests")
    # rough instruction weights of the device kernel: a visit ~100, a primitive test ~70, a root ~45
    cost = (g("node_visits") * 100 + prim * 70 + g("sphere_roots") * 45) / rays
    return dict(records=len(nodes), depth=depth, visits=g("node_visits") / rays, boxes=g("box_tests") / rays,
                prims=prim / rays, roots=g("sphere_roots") / rays, cost=cost)


if __name__ == "__main__":
    names = sys.argv[1:] or ["book_cover", "book2", "cornell", "material_demo"]
    for name in names:
        if name.startswith("cover"):
            scene = HostScene.generate_cover(int(name[5:]), scene_seed=1, width=160, height=90, spp=2, max_depth=50)
        else:
            scene = HostScene.load_frozen(os.path.join(ROOT, "tests", "golden", name + ".rtscene"))
        m = measure(scene)
        print(f"{name:14s} records {m['records']:7d} depth {m['depth']:2d}  visits/ray {m['visits']:6.2f}  "
              f"boxes/ray {m['boxes']:6.2f}  prims/ray {m['prims']:5.2f}  roots/ray {m['roots']:5.2f}  cost {m['cost']:7.1f}")
