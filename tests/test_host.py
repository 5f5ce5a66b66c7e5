"""Host front-end: RON subset, loader semantics, camera, frozen files, resolve, PNG."""
import ctypes as C
import glob
import io
import os
import zlib

import numpy as np
import pytest

from raytracer_amd import abi
from raytracer_amd.scene import (HostScene, human_duration, resolve_srgb8, write_png)
import pyoracle as po

HEAD = """Scene(
  image: (width: 40, height: 20, samples_per_pixel: 3, max_depth: 7),
  camera: (look_from: (13, 2, 3), v_up: (0, 1, 0), vertical_fov: 18, aperture: 0.02, focus_distance: 10),
"""


def test_ron_syntax_features():
    text = HEAD + """
  // line comment
  /* block /* nested */ comment */
  background: (0.7, 0.8, 1.0),
  materials: {
    "a": Lambertian(albedo: Solid(0.1, 0.2, 0.3)),
    "b": Metal (albedo: (0.5, 0.5, 0.5), fuzziness: 0.25,),
    "c": Dielectric(index_of_refraction: 1.5),
    "d": DiffuseLight(color: Checkerboard(Solid(1, 1, 1), Perlin)),
  },
  objects: [
    Sphere (center: (0, -1_000, 0), radius: 1_000, material: Shared("a"),),
    Pattern(var: "i", range: [2], object: Pattern(var: "j", range: [1, 3], object:
      Sphere(center: (("Add", "i", 0.5), 0.2, ("mult", "j", 2)), radius: 0.2, material: Shared("b")))),
    Pattern(var: "k", range: [0, 10, 4], object:
      AARect(center: ("k", 1, 0), width: 1, height: 2, axis: Y, material: Shared("c"))),
    Translate(offset: (1, 2, 3), hittable: RotateY(angle: 30, hittable:
      Cuboid(size: (1, 2, 3), material: Shared("d")))),
    ConstantMedium(boundary: Sphere(center: (0, 0, 0), radius: 2, material: Shared("a")),
      density: 0.5, texture: Solid(1, 1, 1)),
    MovingSphere(center: ((0, 0, 0), (0, 1, 0)), time: (0, 1), radius: 0.3,
      material: Lambertian(albedo: (0.3, 0.2, 0.1))),
  ],
)
"""
    s = HostScene.load_ron_text(text)
    v = s.view
    assert (v.width, v.height, v.samples_per_pixel, v.max_depth) == (40, 20, 3, 7)
    # 1 + 2*2 + 3 (k = 0, 4, 8) + 1 cuboid + 1 medium + 1 moving sphere
    assert v.n_objects == 11
    assert v.n_prims == 1 + 4 + 3 + 6 + 1 + 1
    centers = [tuple(v.prims[i].center) for i in range(1, 5)]
    # outer variable major (loader.rs:208-209): i=0: j=1,2 ; i=1: j=1,2
    assert centers == [(0.5, 0.2, 2.0), (0.5, 0.2, 4.0), (1.5, 0.2, 2.0), (1.5, 0.2, 4.0)]
    assert [v.prims[i].center[0] for i in range(5, 8)] == [0.0, 4.0, 8.0]
    cub = v.objects[8]
    assert cub.is_cuboid and cub.prim_count == 6 and cub.op_count == 2
    assert v.ops[cub.first_op].kind == 0 and v.ops[cub.first_op + 1].kind == 1  # translate outside rotate
    assert v.ops[cub.first_op + 1].a[0] == pytest.approx(0.5) and v.n_perlin == 1
    sides = [(v.prims[cub.first_prim + k].axis, tuple(v.prims[cub.first_prim + k].center),
              v.prims[cub.first_prim + k].width, v.prims[cub.first_prim + k].height) for k in range(6)]
    # cuboid.rs:21-28: (X,-1) (X,+1) (Y,-1) (Y,+1) (Z,-1) (Z,+1); width=size[next], height=size[prev]
    assert sides == [(0, (-0.5, 0, 0), 2, 3), (0, (0.5, 0, 0), 2, 3), (1, (0, -1, 0), 3, 1),
                     (1, (0, 1, 0), 3, 1), (2, (0, 0, -1.5), 1, 2), (2, (0, 0, 1.5), 1, 2)]
    med = v.objects[9]
    assert med.is_medium and med.neg_inv_density == -2.0
    assert v.materials[med.medium_material].kind == 4
    assert list(v.background) == [0.7, 0.8, 1.0]


def test_loader_errors_keep_the_reference_wording():
    with pytest.raises(abi.RtError, match="Material nope not defined"):
        HostScene.load_ron_text(HEAD + 'materials: {}, objects: [Sphere(center: (0,0,0), radius: 1, '
                                       'material: Shared("nope"))])')
    with pytest.raises(abi.RtError, match="Variable q not found"):
        HostScene.load_ron_text(HEAD + 'materials: {}, objects: [Sphere(center: ("q",0,0), radius: 1, '
                                       'material: Dielectric(index_of_refraction: 1.5))])')
    with pytest.raises(abi.RtError, match="loading scene file: line"):
        HostScene.load_ron_text(HEAD + "materials: {}, objects: [Sphere(center: (0,0,0)")
    with pytest.raises(abi.RtError, match="opening scene file"):
        HostScene.load_ron("/nonexistent/scene.ron")
    with pytest.raises(abi.RtError, match="unsupported"):
        HostScene.load_ron_text(HEAD + 'materials: {}, objects: [ConstantMedium(density: 1, texture: '
                                'Solid(1,1,1), boundary: ConstantMedium(density: 1, texture: Solid(1,1,1), '
                                'boundary: Sphere(center: (0,0,0), radius: 1, material: '
                                'Dielectric(index_of_refraction: 1.5))))])')


def test_scene_seed_makes_expansion_reproducible(golden_dir):
    text = HEAD + """materials: {}, objects: [Pattern(var: "a", range: [-3, 3], object:
      Sphere(center: (("add", "a", ("rand", -0.4, 0.4)), 0.2, 0), radius: 0.2,
             material: RandomChoiceWeighted([(16, Lambertian(albedo: (("rand", 0, 1), 0.5, 0.5))),
                                             (3, Metal(albedo: (0.5, 0.5, 0.5), fuzziness: ("rand", 0, 0.5))),
                                             (1, Dielectric(index_of_refraction: 1.5))])))])"""
    a = HostScene.load_ron_text(text, scene_seed=1)
    b = HostScene.load_ron_text(text, scene_seed=1)
    c = HostScene.load_ron_text(text, scene_seed=2)
    ca = [a.view.prims[i].center[0] for i in range(6)]
    assert ca == [b.view.prims[i].center[0] for i in range(6)]
    assert ca != [c.view.prims[i].center[0] for i in range(6)]
    for i, x in enumerate(ca):
        assert -3 + i - 0.4 <= x < -3 + i + 0.4


def test_frozen_round_trip(tmp_path, golden_dir):
    s = HostScene.load_frozen(os.path.join(golden_dir, "book2.rtscene"))
    out = tmp_path / "copy.rtscene"
    s.save_frozen(out)
    t = HostScene.load_frozen(out)
    a, b = s.view, t.view
    assert (a.n_objects, a.n_prims, a.n_ops, a.n_materials, a.n_textures, a.n_perlin, a.n_images,
            a.n_texel_bytes) == (b.n_objects, b.n_prims, b.n_ops, b.n_materials, b.n_textures,
                                 b.n_perlin, b.n_images, b.n_texel_bytes) == (10370, 12175, 10000, 11, 8,
                                                                             1, 1, 1024 * 512 * 3)
    assert bytes(C.string_at(a.prims, a.n_prims * C.sizeof(abi.Prim))) == \
        bytes(C.string_at(b.prims, b.n_prims * C.sizeof(abi.Prim)))
    assert bytes(C.string_at(a.texels, 4096)) == bytes(C.string_at(b.texels, 4096))
    with pytest.raises(abi.RtError):
        bad = tmp_path / "bad.rtscene"
        bad.write_bytes(b"not a scene")
        HostScene.load_frozen(bad)


def test_camera_builder_matches_oracle(golden_dir):
    for name in ["book_cover", "material_demo", "cornell", "book2"]:
        s = HostScene.load_frozen(os.path.join(golden_dir, name + ".rtscene"))
        for (w, h) in [(1920, 1080), (1024, 1024), (400, 225)]:
            s.set_image(w, h)
            mine, ref = s.view.camera, po.camera_build(s.camera_desc, w, h)
            assert bytes(mine) == bytes(ref)


def test_resolve_matches_oracle():
    rng = np.random.default_rng(0)
    sums = rng.uniform(-2, 70, size=(16, 16, 3))
    sums[0, 0] = [np.nan, np.inf, -np.inf]
    sums[0, 1] = [64.0, 32.0, 0.0]
    got = resolve_srgb8(sums, 64)
    want = np.array([po.into_srgb8(px, 64) for px in sums.reshape(-1, 3)], dtype=np.uint8)
    assert np.array_equal(got.reshape(-1, 3), want)
    assert list(got[0, 1]) == [255, 128, 0]
    # float sums resolve as the doubles they convert to exactly
    sums32 = sums.astype(np.float32)
    assert np.array_equal(resolve_srgb8(sums32, 64), resolve_srgb8(sums32.astype(np.float64), 64))


def test_png_is_valid_rgb8(tmp_path):
    img = (np.arange(20 * 10 * 3) % 251).astype(np.uint8).reshape(10, 20, 3)
    path = tmp_path / "image.png"
    write_png(path, img)
    data = path.read_bytes()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    from PIL import Image
    back = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
    assert np.array_equal(back, img)
    # every chunk's CRC is right
    pos = 8
    while pos < len(data):
        n = int.from_bytes(data[pos:pos + 4], "big")
        crc = int.from_bytes(data[pos + 8 + n:pos + 12 + n], "big")
        assert zlib.crc32(data[pos + 4:pos + 8 + n]) == crc
        pos += 12 + n


def test_human_duration_format():
    # src/main.rs:146-165
    assert human_duration(1234) == "1.23s"
    assert human_duration(59_990) == "59.99s"
    assert human_duration(61_000) == "1m1s"
    assert human_duration(3_600_000 + 61_000) == "1h1m1s"
    assert human_duration(60_000) == "60.00s"  # `>` not `>=`: exactly one minute stays in seconds


@pytest.mark.skipif(not os.path.isdir("/root/reference/scenes"), reason="reference tree not present")
def test_every_reference_scene_loads():
    """(globe.ron and book2.ron read scenes/earthmap.jpg through the built-in baseline JPEG decoder)"""
    counts = {}
    for path in sorted(glob.glob("/root/reference/scenes/*.ron")):
        s = HostScene.load_ron(path, scene_seed=1)
        counts[os.path.basename(path)] = s.view.n_objects
    assert counts["book_cover.ron"] == 488 and counts["book2.ron"] == 10370
    assert counts["cornell.ron"] == 8 and counts["bvh_test.ron"] == 29791 and len(counts) == 12


def _jpeg_bytes(img, **kw):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, format="JPEG", **kw)
    return buf.getvalue()


def _smooth_image(h, w, grey=False):
    y, x = np.mgrid[0:h, 0:w]
    r = 128 + 100 * np.sin(x / 9.0) * np.cos(y / 7.0)
    g = 128 + 90 * np.cos(x / 5.0 + y / 11.0)
    b = 255.0 * x / w
    a = np.stack([r, g, b], axis=-1).clip(0, 255).astype(np.uint8)
    return a[..., 0].copy() if grey else a


@pytest.mark.parametrize("shape,kw,tol", [
    ((40, 56), dict(quality=92, subsampling=0), 3),            # 4:4:4, the sampling of earthmap.jpg
    ((33, 47), dict(quality=85, subsampling=0), 3),            # sizes that are no multiple of 8
    ((48, 64), dict(quality=90, subsampling=2), 4),            # 4:2:0: triangle-filtered chroma
    ((48, 64), dict(quality=90, subsampling=1), 4),            # 4:2:2
    ((45, 61), dict(quality=90, subsampling=2), 4),            # 4:2:0, odd size
    ((37, 29), dict(quality=95), 3),                           # grey
])
def test_jpeg_decoder_agrees_with_another_decoder(shape, kw, tol):
    """Decoders differ in IDCT rounding and chroma upsampling (src/texture/image.rs:12-20 uses the
    image crate's own); against PIL's: a few levels, and the same picture on average."""
    from PIL import Image
    from raytracer_amd.scene import decode_jpeg
    grey = "subsampling" not in kw
    data = _jpeg_bytes(_smooth_image(*shape, grey=grey), **kw)
    want = np.asarray(Image.open(io.BytesIO(data)).convert("RGB")).astype(int)
    got = decode_jpeg(data).astype(int)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= tol
    assert np.abs(got - want).mean() < 1.0


def test_jpeg_decoder_restart_intervals_and_refusals():
    from PIL import Image
    from raytracer_amd.abi import RtError
    from raytracer_amd.scene import decode_jpeg
    img = _smooth_image(64, 80)
    plain = _jpeg_bytes(img, quality=90, subsampling=0)
    try:
        with_rst = _jpeg_bytes(img, quality=90, subsampling=0, restart_marker_blocks=3)
    except TypeError:
        with_rst = None
    if with_rst is not None and b"\xff\xdd" in with_rst:
        assert np.array_equal(decode_jpeg(with_rst), decode_jpeg(plain))
    with pytest.raises(RtError):
        decode_jpeg(_jpeg_bytes(img, quality=90, progressive=True))
    with pytest.raises(RtError):
        decode_jpeg(plain[:200])
    with pytest.raises(RtError):
        decode_jpeg(b"not a jpeg at all")


def test_image_texture_loads_the_jpeg_itself(tmp_path):
    """Image("x.jpg") reads x.jpg when it is a baseline JPEG (else x.ppm next to it)."""
    from raytracer_amd.scene import decode_jpeg
    img = _smooth_image(32, 64)
    data = _jpeg_bytes(img, quality=92, subsampling=0)
    (tmp_path / "tex.jpg").write_bytes(data)
    (tmp_path / "scene.ron").write_text("""Scene(
  image: (width: 16, height: 16, samples_per_pixel: 1, max_depth: 2),
  camera: (look_from: (0, 0, 5), look_at: (0, 0, 0), vertical_fov: 30, aperture: 0),
  background: (1, 1, 1),
  materials: {},
  objects: [ Sphere(center: (0, 0, 0), radius: 1, material: Lambertian(albedo: Image("tex.jpg"))) ]
)""")
    s = HostScene.load_ron(str(tmp_path / "scene.ron"))
    v = s.view
    assert v.n_images == 1 and v.images[0].width == 64 and v.images[0].height == 32
    texels = np.ctypeslib.as_array(v.texels, shape=(v.n_texel_bytes,))
    assert np.array_equal(texels.reshape(32, 64, 3), decode_jpeg(data))


def test_frozen_scene_with_a_corrupted_header_is_refused_cleanly(tmp_path):
    """Counts in the header that the file does not back (0xFFFFFFFF in any header word) must end in an
    error code: no exception across the C boundary, no allocation of what the header claims."""
    import gzip
    import resource
    from raytracer_amd.scene import HostScene
    golden = os.path.join(os.path.dirname(__file__), "golden", "material_demo.rtscene")
    raw = gzip.open(golden, "rb").read()
    soft, hard = resource.getrlimit(resource.RLIMIT_AS)
    refused = loaded = 0
    try:
        resource.setrlimit(resource.RLIMIT_AS, (8 << 30, hard))  # a 4 G-record allocation would fail loudly
        for at in range(0, min(len(raw) - 4, 800), 4):
            bad = bytearray(raw)
            bad[at:at + 4] = b"\xff\xff\xff\xff"
            path = tmp_path / "bad.rtscene"
            path.write_bytes(bytes(bad))  # (plain bytes: the reader takes uncompressed files too)
            try:
                HostScene.load_frozen(path).close()
                loaded += 1
            except RuntimeError:
                refused += 1
    finally:
        resource.setrlimit(resource.RLIMIT_AS, (soft, hard))
    assert refused >= 8 and refused + loaded == len(range(0, min(len(raw) - 4, 800), 4))


@pytest.mark.parametrize("args", [["--devices", "0,x"], ["--devices", "a"], ["--devices", "0,"], ["--devices", ""],
                                  ["--gpus", "0"], ["--gpus", "-3"], ["--gpus", "two"]])
def test_cli_rejects_malformed_device_lists(args):
    """A device list that is not ordinals separated by commas, or a count below one, ends with the
    usage error (exit status 2) before anything is loaded."""
    import subprocess
    from raytracer_amd import abi
    exe = os.path.join(abi.LIB_DIR, "raytracer")
    r = subprocess.run([exe, "no_such_scene.ron"] + args, capture_output=True, text=True, timeout=20)
    assert r.returncode == 2 and "error:" in r.stderr
