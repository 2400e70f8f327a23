"""Screenshot of the renderer output: the caller side of GetOutputTexturePixels.

Follows the Sandbox's OutputLayer::MakeScreenshot (Sandbox/src/OutputLayer.cpp:882-896): fetch the RGBA8 output, apply the
display gamma per colour channel (pow(c/255, 1/gamma)*255, truncated to a byte; alpha is left alone; default gamma 2.2,
OutputLayer.h:73), create the parent directory and write an 8-bit RGBA PNG.  The reference writes the PNG with
stb_image_write; the writer here emits a plain zlib-compressed, filter-0 PNG, which decodes to the same pixels.
"""
import os
import struct
import zlib

import numpy as np


def apply_gamma(pixels_rgba8, gamma=2.2):
    """Per-channel display gamma of OutputLayer.cpp:886-891 on an (H, W, 4) uint8 array; returns a new array."""
    px = np.ascontiguousarray(pixels_rgba8, np.uint8)
    if px.ndim != 3 or px.shape[2] != 4:
        raise ValueError("expected an (H, W, 4) uint8 image")
    # 256-entry table: float32 arithmetic like the reference's powf, the conversion to a byte truncates
    lut = (np.power(np.arange(256, dtype=np.float32) / np.float32(255.0), np.float32(1.0) / np.float32(gamma)) * np.float32(255.0)).astype(np.uint8)
    out = px.copy()
    out[..., :3] = lut[px[..., :3]]
    return out


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def write_png(path, pixels_rgba8, level=6):
    """Write an (H, W, 4) uint8 array as an 8-bit RGBA PNG (colour type 6, no interlace, filter 0 on every row)."""
    px = np.ascontiguousarray(pixels_rgba8, np.uint8)
    if px.ndim != 3 or px.shape[2] != 4:
        raise ValueError("expected an (H, W, 4) uint8 image")
    h, w = px.shape[:2]
    if h == 0 or w == 0:
        raise ValueError("empty image")
    rows = np.zeros((h, 1 + w * 4), np.uint8)
    rows[:, 1:] = px.reshape(h, w * 4)
    parent = os.path.dirname(os.path.abspath(path))
    os.makedirs(parent, exist_ok=True)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)))
        f.write(_chunk(b"IDAT", zlib.compress(rows.tobytes(), level)))
        f.write(_chunk(b"IEND", b""))


def read_png_rgba8(path):
    """Minimal reader for files written by write_png (used by the tests to check the file without a third-party decoder)."""
    data = open(path, "rb").read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG file")
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        if crc != (zlib.crc32(tag + body) & 0xFFFFFFFF):
            raise ValueError("bad chunk checksum")
        if tag == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
            if (depth, ctype, interlace) != (8, 6, 0):
                raise ValueError("only 8-bit RGBA, non-interlaced files are supported")
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * 4)
    if rows[:, 0].any():
        raise ValueError("only filter 0 rows are supported")
    return rows[:, 1:].reshape(h, w, 4).copy()


def make_screenshot(renderer, path, gamma=2.2):
    """OutputLayer::MakeScreenshot for a LumenRendererMI: returns the pixels that were written."""
    px = apply_gamma(renderer.GetOutputTexturePixels(), gamma)
    write_png(path, px)
    return px
