"""The linear-filtering rule of the CUDA C Programming Guide (appendix "Texture Fetching": "Linear Filtering"), restated INDEPENDENTLY of oracle/ and csrc/ in
float64 numpy straight from the Guide's text — what the reference's tex2D<float4> fetches go through (PTTexture.cpp:57-73: cudaFilterModeLinear, cudaAddressModeWrap,
normalizedCoords, cudaReadModeNormalizedFloat, sRGB):

    wrap, normalised coordinates:   x = N frac(u)                                  ("x is replaced by frac(x)")
    linear filter:                  xB = x - 0.5,  i = floor(xB),  alpha = frac(xB)   (same for y: j, beta)
                                    tex = (1-a)(1-b) T[i,j] + a(1-b) T[i+1,j] + (1-a) b T[i,j+1] + a b T[i+1,j+1]
    "alpha, beta ... are stored in 9-bit fixed point format with 8 bits of fractional value (so 1.0 is exactly represented)"

The Guide does not say how alpha is rounded into that format; the oracle and the product round to nearest (decision D6) and so does this restatement.  Texel values:
byte / 255, or the sRGB decode of the byte (per texel, before filtering) for textures created with normalize."""
import numpy as np


def srgb_decode(b):
    c = np.asarray(b, np.float64) / 255.0
    return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)


def guide_tex2d(pixels, srgb, uv, quantise=True):
    """pixels [h][w][4] uint8, uv [n][2] float32 -> [n][4] float64 by the Guide's four-term formula."""
    px = np.asarray(pixels, np.uint8); h, w = px.shape[:2]
    T = np.empty((h, w, 4), np.float64)
    T[..., :3] = srgb_decode(px[..., :3]) if srgb else px[..., :3] / 255.0
    T[..., 3] = px[..., 3] / 255.0
    uv = np.asarray(uv, np.float32)
    fu = (uv[:, 0] - np.floor(uv[:, 0])).astype(np.float32); fv = (uv[:, 1] - np.floor(uv[:, 1])).astype(np.float32)
    xb = (fu * np.float32(w) - np.float32(0.5)).astype(np.float64); yb = (fv * np.float32(h) - np.float32(0.5)).astype(np.float64)
    i = np.floor(xb); j = np.floor(yb)
    a = xb - i; b = yb - j
    if quantise:
        a = np.floor(a * 256.0 + 0.5) / 256.0; b = np.floor(b * 256.0 + 0.5) / 256.0
    i0 = i.astype(np.int64) % w; i1 = (i0 + 1) % w; j0 = j.astype(np.int64) % h; j1 = (j0 + 1) % h
    a = a[:, None]; b = b[:, None]
    return (1 - a) * (1 - b) * T[j0, i0] + a * (1 - b) * T[j0, i1] + (1 - a) * b * T[j1, i0] + a * b * T[j1, i1]


def test_textures(seed=5):
    """(pixels, srgb) pairs: a ragged high-contrast map, a power-of-two noise map, a 2 x 3 map, a one-row and a one-column map."""
    rng = np.random.default_rng(seed)
    out = []
    for (h, w), srgb in (((37, 53), True), ((64, 64), False), ((3, 2), True), ((1, 17), False), ((9, 1), True)):
        px = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        px[rng.random((h, w)) < 0.3] = (255, 0, 255, 0)          # hard edges: where a weight error shows
        out.append((px, srgb))
    return out


def test_coordinates(n=4000, seed=6):
    rng = np.random.default_rng(seed)
    uv = rng.uniform(-3.0, 4.0, (n, 2)).astype(np.float32)
    uv[:64] = rng.integers(-2, 3, (64, 2)).astype(np.float32)                       # exact integers: frac = 0, the wrap seam
    uv[64:128, 0] = (rng.integers(0, 53, 64) + 0.5) / np.float32(53.0)              # texel centres of the first map
    return uv
