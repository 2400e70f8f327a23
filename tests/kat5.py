"""Rows of tests/golden/ref_kat5.npz — what the reference's own __global__ kernel bodies computed, thread by thread, on a 64 x 48 synthetic image
over three frames (generator: oracle/ref_kat/gen_kat5.cpp + make_kat.py; layouts in the generator's header) — as arrays both sides of the parity
tests consume: 32-bit words, floats by bit pattern, integers as they are."""
import os
import numpy as np

W, H = 64, 48
N = W * H
FRAMES = 3
STAGES = ("pick", "temporal", "spatial1", "spatial2", "combine")
_G = None


def gold():
    global _G
    if _G is None:
        _G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_kat5.npz"))
    return _G


def u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def as_f32(words):
    return u32(words).view(np.float32)


def lights():
    g = gold()
    return u32(g["light"]), u32(g["cdf"][:, 0]), u32(g["cdfw"][:, 0])


def frame(f):
    """Inputs and reference outputs of frame f of the ReSTIR chain (Framework/ReSTIR.cpp:65-233)."""
    g = gold()
    seed_row = g["seed"][g["seed"][:, 0] == f][0]
    surf = g["surf"]; surf = surf[surf[:, 0] == f]
    assert np.array_equal(surf[:, 1], np.arange(N))
    mot = g["mot"]; mot = mot[mot[:, 0] == f]
    occ = g["occ"]; occ = occ[occ[:, 0] == f]
    res = g["res"]; res = res[res[:, 0] == f]
    ray = g["ray"]; ray = ray[ray[:, 0] == f]
    shd = g["shd"]; shd = shd[shd[:, 0] == f]
    out = {
        "seed": int(seed_row[1]), "current": int(seed_row[2]),
        "surf": u32(surf[:, 2:]), "motion": u32(mot[:, 2:]),
        "occ": [np.ascontiguousarray(occ[occ[:, 1] == p][:, 3], dtype=np.uint8) for p in (0, 1)],
        "stages": np.stack([u32(res[res[:, 1] == s][:, 3:]) for s in range(5)]),          # [5][N][17]
        "rays": [u32(ray[ray[:, 1] == p][:, 2:]) for p in (0, 1)],                         # (index, origin, direction, distance), append order
        "shade_from": np.zeros((3, N), np.uint32),
    }
    for site in range(3):
        rows = shd[shd[:, 1] == site]
        to = rows[:, 5] * W + rows[:, 4]
        assert len(np.unique(to)) == len(to)                                              # at most one ShadeReservoirs call per pixel and call site
        out["shade_from"][site, to] = rows[:, 3] * W + rows[:, 2] + 1
    if f == 0:
        b = g["bags"]
        assert np.array_equal(b[:, 1], np.arange(50000))
        out["bags"] = u32(b[:, 2:])
    return out


def reservoirs_before(f):
    """The four reservoir buffers as frame f finds them: zero before the first frame (ResetReservoirs on fresh memory); afterwards what the kernels of the
    earlier frames left — the swap-chain buffer each frame's combine wrote, and buffers 2 / 3 from the last spatial passes."""
    res4 = np.zeros((4, N, 17), np.uint32)
    for k in range(f):
        fr = frame(k)
        res4[fr["current"]] = fr["stages"][4]
        res4[2] = fr["stages"][2]; res4[3] = fr["stages"][3]
    return res4


def expected_direct(f):
    """DIRECT channel after frame f under decision D1 (fp32 accumulation): every recorded ShadeReservoirs call adds contribution * (weight / 3) of the reservoir
    it names, if that reservoir's weight is > 0 (ReSTIRKernels.cu:619-665), in call-site order.  Site 0 reads the picked reservoirs after visibility pass 1,
    site 1 the PREVIOUS frame's buffer, site 2 the temporal result after visibility pass 2."""
    fr = frame(f); before = reservoirs_before(f)
    cur = fr["current"]
    direct = np.zeros((N, 3), np.float32)

    def after_visibility(stage_words, p):
        r = stage_words.copy()
        idx = fr["rays"][p][:, 0]
        blocked = idx[fr["occ"][p][idx] != 0]
        r[blocked, 2] = 0                                                                  # weight = 0.f
        return r

    sources = [after_visibility(fr["stages"][0], 0), before[cur ^ 1], after_visibility(fr["stages"][1], 1)]
    for site in range(3):
        frm = fr["shade_from"][site]
        to = np.nonzero(frm)[0]
        r = sources[site][frm[to] - 1]
        wgt = as_f32(r[:, 2]); contrib = as_f32(r[:, 13:16])
        add = contrib * (wgt / np.float32(3.0))[:, None]
        live = wgt > 0
        direct[to[live]] = direct[to[live]] + add[live]
    return direct


def shade_rows(tag):
    g = gold()[tag]
    return u32(g[:, :43]), u32(g[:, 43:])


def primary():
    g = gold()
    return u32(g["camr"][0]), g["prim"]
