"""tga2png.py — convert the 24-bit bottom-up TGA that ptss_write_tga / saveScreenshot emits into a PNG (stdlib only)."""
import struct
import sys
import zlib

import numpy as np


def read_tga(path):
    b = open(path, "rb").read()
    w = b[12] | (b[13] << 8)
    h = b[14] | (b[15] << 8)
    assert b[2] == 2 and b[16] == 24
    a = np.frombuffer(b, np.uint8, w * h * 3, 18).reshape(h, w, 3)
    return a[::-1, :, ::-1]  # bottom-up BGR -> top-down RGB


def write_png(path, rgb):
    h, w, _ = rgb.shape
    raw = b"".join(b"\x00" + rgb[y].tobytes() for y in range(h))

    def chunk(t, d):
        c = struct.pack(">I", len(d)) + t + d
        return c + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


if __name__ == "__main__":
    write_png(sys.argv[2], np.ascontiguousarray(read_tga(sys.argv[1])))
