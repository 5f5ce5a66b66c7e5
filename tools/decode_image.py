"""Decode an image file to a binary PPM (P6) once, here, with PIL.

The render path links no JPEG decoder; `Image("x.jpg")` textures are read from
`x.ppm`.  Usage: python tools/decode_image.py in.jpg out.ppm
"""
import hashlib
import sys

from PIL import Image


def main(src, dst):
    im = Image.open(src).convert("RGB")
    w, h = im.size
    data = im.tobytes()
    with open(dst, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(data)
    print(f"{dst}: {w}x{h}, sha256(rgb8)={hashlib.sha256(data).hexdigest()}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
