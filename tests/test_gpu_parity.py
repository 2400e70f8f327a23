"""GPU parity tests (run with -m gpu on an MI355X): every call goes through the C ABI of liblumen_mi.so and is compared
with the CPU oracle on the same seeded inputs.  Integer/index work and — because every stage is specified as exact
fp32 arithmetic — the floating-point results are held to bit equality; the north-star tolerance (1e-3 relative L2 on
radiance) is asserted as the outer bar."""
import os
import re
import numpy as np
import pytest

from helpers import GOLDEN, cornell, oracle_from, product_from, random_soup, rel_l2
from oracle_lib import lib as orc_lib, fptr, f32

pytestmark = pytest.mark.gpu
KAT = np.load(os.path.join(GOLDEN, "ref_kat.npz"))
RADIANCE_TOL = 1e-3           # BASELINE.json north_star: 1e-3 relative L2


@pytest.fixture(scope="module")
def bare():
    from lumenrenderer_amd import LumenRendererMI
    r = LumenRendererMI(); r.Init(depth=2, render_resolution=(16, 16))
    yield r
    r.close()


def test_library_is_the_hip_one(bare):
    import lumenrenderer_amd
    assert os.path.exists(lumenrenderer_amd.library_path())
    maps = open("/proc/self/maps").read()
    assert "liblumen_mi.so" in maps and "libamdhip64" in maps


@pytest.mark.parametrize("fn,lo,hi", [(0, -2.0, 8.0), (1, -2.0, 8.0), (2, 1e-7, 100.0), (3, -90.0, 30.0)])
def test_device_transcendentals_bit_exact(bare, fn, lo, hi):
    x = np.linspace(lo, hi, 200003).astype(np.float32)
    got = bare.TestMath(fn, x)
    want = np.zeros_like(x); orc_lib().orc_det_math(x.size, fn, fptr(x), fptr(x), fptr(want))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_device_pow_halton_hash_half_bit_exact(bare):
    rng = np.random.default_rng(3); L = orc_lib()
    a, b = rng.uniform(0, 1, 50000).astype(np.float32), rng.uniform(0, 3, 50000).astype(np.float32)
    want = np.zeros_like(a); L.orc_det_math(a.size, 4, fptr(a), fptr(b), fptr(want))
    assert np.array_equal(bare.TestMath(4, a, b).view(np.uint32), want.view(np.uint32))
    idx = rng.integers(0, 2 ** 31, 20000, dtype=np.uint32)
    for base in (2, 3):
        got = bare.TestMath(5, idx.view(np.float32), np.full(idx.size, base, np.uint32).view(np.float32))
        want = np.array([L.orc_halton(int(i), base) for i in idx[:4000]], np.float32)
        assert np.array_equal(got[:4000].view(np.uint32), want.view(np.uint32))
    # ... and against the REFERENCE's HaltonSequence compiled from its own text (ref_kat.npz rows `halt`, incl. the index that wraps in `++index`)
    hr = KAT["halt"]
    for base in (2, 3):
        rows = hr[hr[:, 1] == base]
        got = bare.TestMath(5, rows[:, 0].astype(np.uint32).view(np.float32), np.full(len(rows), base, np.uint32).view(np.float32))
        assert np.array_equal(got.view(np.uint32), rows[:, 2].astype(np.uint32)), base
    big = np.concatenate([rng.integers(2 ** 24 - 64, 2 ** 24 + 64, 500, dtype=np.uint32), rng.integers(0, 2 ** 32, 4000, dtype=np.uint64).astype(np.uint32)])
    for base in (2, 3):
        got = bare.TestMath(5, big.view(np.float32), np.full(big.size, base, np.uint32).view(np.float32))
        want = np.array([L.orc_halton(int(i), base) for i in big], np.float32)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), base
    got = bare.TestMath(6, idx.view(np.float32)).view(np.uint32)
    assert got[:2000].tolist() == [L.orc_wang_hash(int(i)) for i in idx[:2000]]
    xs = np.concatenate([rng.uniform(-2, 2, 20000), rng.uniform(-7e4, 7e4, 2000), rng.uniform(-1e-5, 1e-5, 2000)]).astype(np.float32)
    with np.errstate(over="ignore"):
        want16 = xs.astype(np.float16).view(np.uint16)
    assert np.array_equal(bare.TestMath(7, xs).view(np.uint32).astype(np.uint16), want16)
    hs = np.arange(0, 0x7c00, dtype=np.uint32)
    assert np.array_equal(bare.TestMath(8, hs.view(np.float32)), hs.astype(np.uint16).view(np.float16).astype(np.float32))


def test_device_bsdf_matches_oracle_bit_exact_and_reference_goldens(bare):
    L = orc_lib()
    e = KAT["eval"]; n = e.shape[0]
    mat, N, T, wo, wi = f32(e[:, :23]), f32(e[:, 26:29]), f32(e[:, 29:32]), f32(e[:, 32:35]), f32(e[:, 35:38])
    got = bare.TestBsdf(0, mat, N, T, wo, wi)[:, :4]
    want = np.zeros((n, 4), np.float32); L.orc_eval_bsdf(n, fptr(mat), fptr(N), fptr(T), fptr(wo), fptr(wi), fptr(want))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    ref = e[:, 38:42]
    assert np.nanmax(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3)) < 2e-4          # the reference's own headers
    s = KAT["samp"]
    mat, N, T, wo, r3 = f32(s[:, :23]), f32(s[:, 26:29]), f32(s[:, 29:32]), f32(s[:, 32:35]), f32(s[:, 35:38])
    got = bare.TestBsdf(1, mat, N, T, wo, r3)
    want = np.zeros((n, 8), np.float32); L.orc_sample_bsdf(n, fptr(mat), fptr(N), fptr(T), fptr(wo), fptr(r3), fptr(want))
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), np.argwhere(~same)[:5]


def test_device_reservoir_cdf_and_output_quantisation_match_reference_header_vectors(bare):
    """The device functions themselves against vectors generated from the reference's own headers (tests/golden/ref_kat.npz rows
    resv / cdfq / color, oracle/ref_kat/gen_kat2.cpp): Reservoir::Update / UpdateWeight (ReSTIRData.h:115-163) through lm_res_update /
    lm_res_update_weight, CDF::Get (ReSTIRData.h:230-306) through lm_cdf_get, make_color (cuda/helpers.h:35-66) through lm_srgb8 —
    bit for bit, except make_color whose libm powf the fixed polynomial pow replaces (one quantisation step at most, rarely)."""
    from test_oracle_kat import kat_reservoir_rows, kat_cdf_cases, _bits_to_f32
    w, pdf, seeds, ws, cnt, held, took, weight, _ = kat_reservoir_rows()
    out = bare.TestRestir(0, w, pdf, seeds).reshape(-1, 33)
    per = out[:, :32].reshape(-1, 8, 4)
    assert np.array_equal(per[:, :, 0].copy().view(np.uint32), ws)
    assert np.array_equal(per[:, :, 1].astype(np.int64), cnt) and np.array_equal(per[:, :, 2].astype(np.int32), held)
    assert np.array_equal(per[:, :, 3].astype(np.int32), took)
    assert np.array_equal(out[:, 32].copy().view(np.uint32), weight)
    for cid, data, values, idx, pdfbits in kat_cdf_cases():
        o = bare.TestRestir(1, data, values).reshape(-1, 2)
        assert np.array_equal(o[:, 0].copy().view(np.uint32), idx), cid
        assert np.array_equal(o[:, 1].copy().view(np.uint32), pdfbits), cid
    g = KAT["color"]
    rgb = np.stack([_bits_to_f32(g[:, k]) for k in range(3)], axis=1).ravel()
    q = bare.TestRestir(2, rgb).reshape(-1, 3).astype(np.int32)
    diff = np.abs(q - g[:, 3:6].astype(np.int32))
    assert diff.max() <= 1 and (diff > 0).mean() < 2e-3
    # and the device agrees with the oracle on the same inputs exactly
    L = orc_lib()
    oq = np.zeros((g.shape[0], 4), np.uint8)
    L.orc_make_color(g.shape[0], fptr(np.ascontiguousarray(rgb)), oq.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_uint8)))
    assert np.array_equal(q, oq[:, :3].astype(np.int32))


def _rel_floor(got, ref, floor=1e-3):
    got = got.astype(np.float64)
    both_nan = np.isnan(got) & np.isnan(ref)
    return np.nan_to_num(np.where(both_nan, 0.0, np.abs(got - ref) / np.maximum(np.abs(ref), floor)), nan=1.0)


def test_device_resample_and_combine_match_the_reference_functions(bare):
    """Round 3: the DEVICE functions lm_resample / lm_combine2 (lm_restir.h) against rows produced by the reference's own text of Resample
    and CombineBiased (ReSTIRKernels.cu:1259-1325, :1200-1257; oracle/ref_kat/gen_kat4.cpp), in both arithmetic policies.
    Exact policy: bit-identical to the oracle, and to the reference up to the libm-vs-polynomial difference inside EvaluateBSDF.
    Fast policy (what bench.py's headline runs), on the surfaces the contracted evaluation covers: same early-outs, same held sample."""
    from test_oracle_kat import resample_rows
    L = orc_lib()
    surf, smp, ref = resample_rows(); n = surf.shape[0]
    want = np.zeros((n, 4), np.float32); L.orc_resample(n, fptr(surf), fptr(smp), fptr(want))
    got = bare.TestRestir(3, surf, smp)
    assert np.all(got[:, 4] == 1)
    same = (got[:, :4].copy().view(np.uint32) == want.view(np.uint32)) | (np.isnan(got[:, :4]) & np.isnan(want))
    assert same.all(), np.argwhere(~same)[:5]
    assert np.array_equal(got[:, 3] == 0, ref[:, 3] == 0)
    err = _rel_floor(got[:, :4], ref).max()
    assert err < 2e-5, err
    fast = bare.TestRestir(5, surf, smp)
    ok = fast[:, 4] == 1
    assert 0.3 < ok.mean() < 0.9                                  # the contracted evaluation covers the opaque isotropic stack only
    # an early-out decided by a comparison within rounding of its threshold may differ: bound how many and how large
    flip = ok & ((fast[:, 3] == 0) != (ref[:, 3] == 0))
    assert flip.sum() <= 3 and np.all(np.maximum(fast[flip, 3], ref[flip, 3]) < 1e-3), (flip.sum(), ref[flip, 3])
    live = ok & ~flip & (ref[:, 3] != 0)
    ferr = _rel_floor(fast[live, :4], ref[live], floor=1e-4)
    rowerr = ferr.max(axis=1); worst = np.flatnonzero(live)[np.argsort(-rowerr)[:3]]
    print(f"fast Resample vs reference: max rel {ferr.max():.3e}, p99 {np.quantile(rowerr, 0.99):.3e} over {live.sum()} rows; exact vs reference {err:.3e}; "
          f"worst rows {worst.tolist()} roughness {surf[worst, 12 + 15].tolist()} got {fast[worst, :4].tolist()} ref {ref[worst].tolist()}")
    # Conditioning.  Two correct binary32 evaluations of this function differ by more than rounding where the function amplifies the 1e-7
    # rounding of the unit light direction: (a) an emitter or receiver seen edge-on — cosOut or cosIn is a difference of products, relative
    # error 1e-7 / cos; (b) the specular peak of a smooth surface (alpha^2 = roughness^4 ~ 1e-4): the half vector's error is divided by
    # alpha^2 in ANY formulation, the reference's tangent-frame form included.  Away from both the agreement is 1e-6.
    to = smp[:, 6:9].astype(np.float64) - surf[:, 0:3]; to /= np.linalg.norm(to, axis=1, keepdims=True)
    cos = np.minimum(np.einsum("ij,ij->i", to, surf[:, 3:6].astype(np.float64)), -np.einsum("ij,ij->i", to, smp[:, 3:6].astype(np.float64)))[live]
    smooth = surf[live, 12 + 15] < 0.2
    bound = 2e-5 + 1e-6 / np.maximum(cos, 1e-6) + np.where(smooth, 1e-3, 0.0)
    assert np.all(rowerr <= bound), (np.flatnonzero(live)[rowerr > bound][:5], rowerr[rowerr > bound][:5])
    well = (cos > 0.05) & ~smooth
    print(f"  well-conditioned rows (cosines > 0.05, roughness >= 0.2): {well.sum()}, max rel {rowerr[well].max():.3e}")
    assert well.sum() > 300 and rowerr[well].max() < 4e-5
    stale = ok & ~flip & (ref[:, 3] == 0) & np.all(ref[:, :3].astype(np.float32) == smp[:, 10:13], axis=1)
    assert np.array_equal(fast[stale, :3], smp[stale, 10:13])     # the geometric early-out keeps the stale contribution

    g = KAT["cmbb2"]; n = g.shape[0]
    surf = f32(g[:, :35]); seeds = np.ascontiguousarray(g[:, 36], dtype=np.uint32); res = f32(g[:, 37:71]); ref = g[:, 71:]
    want = np.zeros((n, 17), np.float32); L.orc_combine_biased(n, 2, fptr(surf), seeds.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_uint32)), fptr(res), fptr(want))
    got = bare.TestRestir(4, surf, res, seeds)
    same = (got[:, :17].copy().view(np.uint32) == want.view(np.uint32)) | (np.isnan(got[:, :17]) & np.isnan(want))
    assert same.all(), np.argwhere(~same)[:5]
    assert np.array_equal(got[:, 9:13], ref[:, 9:13].astype(np.float32))          # the held sample: same choice as the reference
    from test_oracle_kat import _strict                                           # no floor, zeros must match zeros (round 4: no masked column)
    assert _strict(got[:, :17], ref).max() < 5e-7
    fast = bare.TestRestir(6, surf, res, seeds)
    ok = fast[:, 17] == 1
    held_same = np.all(fast[:, 9:13] == ref[:, 9:13].astype(np.float32), axis=1)
    assert (ok & ~held_same).sum() <= 2, np.flatnonzero(ok & ~held_same)        # rnd <= w / weightSum decided within an ulp
    live = ok & held_same
    ferr = _rel_floor(fast[live, :17], ref[live], floor=1e-4)
    rowerr = ferr.max(axis=1)
    print(f"fast CombineBiased vs reference: max rel {ferr.max():.3e}, p99 {np.quantile(rowerr, 0.99):.3e} over {live.sum()} rows")
    assert np.quantile(rowerr, 0.9) < 1e-5 and rowerr.max() < 5e-3, ferr.max()       # (the same conditioning as above, through both resampled inputs)


# ---- round 4: the KERNELS of the hot path against what the reference's own __global__ bodies computed (tests/golden/ref_kat5.npz, oracle/ref_kat/gen_kat5.cpp)
def _device_restir_frame(bare, f, fast=0, res4=None):
    import kat5
    lt, cdf, _ = kat5.lights()
    fr = kat5.frame(f)
    prev = kat5.frame(f - 1)["surf"] if f else None
    out = bare.TestRestirFrame(kat5.W, kat5.H, fr["surf"], prev, fr["motion"], lt, cdf, fr["seed"], fr["current"], fr["occ"][0], fr["occ"][1],
                               kat5.reservoirs_before(f) if res4 is None else res4, fast=fast)
    return fr, out


def _rays_by_pixel(rows):
    order = np.argsort(rows[:, 0], kind="stable")
    return rows[order]


@pytest.mark.parametrize("f", [0, 1, 2])
def test_device_restir_kernels_match_the_reference_kernels(bare, f):
    """lm_k_fill_bags, lm_k_pick_primary (+ GenerateShadowRay fused), lm_k_restir_temporal (+ the second ray generation, + ShadeReservoirs of the previous
    reservoir), lm_k_restir_spatial x 2, lm_k_restir_combine — launched with frame.cpp's arguments on the synthetic frames — against the rows the reference's own
    kernel text produced, and bit for bit against the oracle on the same rows.  Exact arithmetic policy."""
    import kat5
    from test_oracle_kat import _run_restir_frame, assert_reservoirs_match, KAT5_FLOAT_TOL
    fr, out = _device_restir_frame(bare, f)
    _, orc = _run_restir_frame(f)
    if f == 0:
        assert np.array_equal(out["bags"], fr["bags"])
    for p in (0, 1):
        got = _rays_by_pixel(out["rays"][p]); want = _rays_by_pixel(fr["rays"][p])          # the device queue is appended block by block: compare by pixel
        assert np.array_equal(got, want), p
    for s, name in enumerate(kat5.STAGES):
        assert_reservoirs_match(out["stages"][s], fr["stages"][s], f"frame {f} after {name}")
        assert np.array_equal(out["stages"][s], orc["stages"][s]), f"device vs oracle, frame {f} after {name}"
    got = kat5.as_f32(out["direct"])[:, :3]; want = kat5.expected_direct(f)
    assert np.array_equal(got == 0, want == 0)
    assert (np.abs(got - want) / np.maximum(np.abs(want), 1e-30)).max() <= KAT5_FLOAT_TOL
    assert np.array_equal(out["direct"], orc["direct"])
    assert np.array_equal(out["res4"], orc["res4"])


def test_device_restir_kernels_fast_policy_track_the_reference_kernels(bare):
    """The same kernels in the fast arithmetic policy (both launches: contracted evaluation + the exact launch for the surfaces it does not cover), each frame started
    from the reference's own state: decisions may flip where a comparison falls within rounding of its threshold — bounded in number — and what is held
    agrees to the tolerance of the fast policy."""
    import kat5
    flips = 0; total = 0
    for f in range(kat5.FRAMES):
        fr, out = _device_restir_frame(bare, f, fast=2)
        for p in (0, 1):
            a = set(out["rays"][p][:, 0].tolist()); b = set(fr["rays"][p][:, 0].tolist())
            assert len(a ^ b) <= 8, (f, p, len(a ^ b))
        for s, name in enumerate(kat5.STAGES):
            got = out["stages"][s]; ref = fr["stages"][s]
            assert np.array_equal(got[:, 1], ref[:, 1]), (f, name)                            # sample counts never depend on the arithmetic
            held = (got[:, 3:13] == ref[:, 3:13]).all(axis=1)
            flips += int((~held).sum()); total += len(held)
            cols = [0, 2, 16]
            a = kat5.as_f32(got)[held][:, cols].astype(np.float64); b = kat5.as_f32(ref)[held][:, cols].astype(np.float64)
            live = (b != 0) & (a != 0)
            err = np.abs(a - b)[live] / np.abs(b)[live]
            assert np.quantile(err, 0.99) < 2e-4 and err.max() < 5e-2, (f, name, float(np.quantile(err, 0.99)), float(err.max()))
    print(f"fast policy: {flips} of {total} reservoir decisions differ from the reference rows")
    assert flips <= total // 500


def test_device_primary_ray_kernel_matches_the_reference_kernel(bare):
    import kat5
    cam, prim = kat5.primary()
    for fc in np.unique(prim[:, 1]):
        rows = prim[prim[:, 1] == fc]
        out = bare.TestPrimaryRays(kat5.W, kat5.H, int(fc), cam)
        assert np.array_equal(out, rows[:, 2:].astype(np.uint32)), int(fc)


def test_device_shade_direct_and_indirect_match_the_reference_kernels(bare):
    """lm_shade_direct / lm_shade_indirect (what lm_k_extract0, lm_k_shade_wave and lm_k_path_tail call per surface) on the rows of the reference's ShadeDirect /
    ShadeIndirect: bit-identical to the oracle, and to the reference as the oracle is (tests/test_oracle_kat.py)."""
    import kat5
    from oracle_lib import u32ptr
    L = orc_lib(); lt, cdf, _ = kat5.lights()
    rin, ref = kat5.shade_rows("sdir"); n = len(rin)
    want = np.zeros((n, 12), np.uint32)
    L.orc_kat_shade(n, kat5.W, kat5.H, u32ptr(rin), len(lt), u32ptr(lt), u32ptr(cdf), u32ptr(want), None)
    got, _ = bare.TestShade(kat5.W, kat5.H, rin, lt, cdf, indirect=False)
    assert np.array_equal(got, want)
    assert np.array_equal(got[:, 0], ref[:, 0]) and np.array_equal(got[ref[:, 0] == 1][:, 1:8], ref[ref[:, 0] == 1][:, 1:8])
    em = ref[:, 0] == 1
    a = kat5.as_f32(got[em][:, 8:11]).astype(np.float64); b = kat5.as_f32(ref[em][:, 8:11]).astype(np.float64)
    assert (np.abs(a - b) / np.maximum(np.abs(b), 1e-30)).max() <= 2e-6
    fastd, _ = bare.TestShade(kat5.W, kat5.H, rin, lt, cdf, fast=1, indirect=False)         # tuning key fast_shade: the contribution in hardware rcp / rsq
    assert (fastd[:, 0] != ref[:, 0]).sum() <= 4
    both = (fastd[:, 0] == 1) & em
    a = kat5.as_f32(fastd[both][:, 8:11]).astype(np.float64); b = kat5.as_f32(ref[both][:, 8:11]).astype(np.float64)
    assert np.quantile(np.abs(a - b) / np.maximum(np.abs(b), 1e-30), 0.99) < 1e-4
    rin, ref = kat5.shade_rows("sind")
    want = np.zeros((n, 10), np.uint32)
    L.orc_kat_shade(n, kat5.W, kat5.H, u32ptr(rin), len(lt), u32ptr(lt), u32ptr(cdf), None, u32ptr(want))
    _, got = bare.TestShade(kat5.W, kat5.H, rin, lt, cdf, direct=False)
    assert np.array_equal(got, want)
    assert np.array_equal(got[:, 0], ref[:, 0])


@pytest.fixture(scope="module")
def kat6_pair():
    import kat6
    d = kat6.scene()
    r = product_from(d, kat6.W, kat6.H, 3); o = oracle_from(d, kat6.W, kat6.H, 3)
    yield r, o
    r.close(); o.close()


KEPT35 = [c for c in range(35) if not 8 <= c < 11]      # geomNormal is not kept by this build (no reference kernel reads it: DESIGN.md)


@pytest.mark.parametrize("which", [0, 1])
def test_device_surface_extraction_matches_the_reference_kernel(kat6_pair, which):
    """lm_extract — what lm_k_extract0, lm_k_shade_wave, lm_k_path_tail run per hit — on the hit / ray rows of the reference's own ExtractSurfaceDataGpu text, against
    the scene loaded through the ordinary API: every kept word bit for bit, for both ray sets (the primary wave; a deeper wave with its own origins and transport)."""
    import kat6
    from oracle_lib import u32ptr
    r, o = kat6_pair
    h9, r9, want = kat6.hits(which)
    got = r.TestExtract(h9, r9)
    orc = np.zeros((kat6.N, 35), np.uint32)
    o.L.orc_kat_extract(o.h, kat6.N, u32ptr(h9), u32ptr(r9), u32ptr(orc))
    assert np.array_equal(orc, want)
    bad = np.flatnonzero((got[:, KEPT35] != want[:, KEPT35]).any(axis=1))
    assert len(bad) == 0, (len(bad), int(bad[0]), int(want[bad[0], 0]), np.flatnonzero(got[bad[0]] != want[bad[0]]).tolist())
    assert not got[:, 8:11].any()


def test_device_depth0_kernel_matches_the_reference_extraction_and_motion_vector_kernels(kat6_pair):
    """lm_k_extract0 itself, launched as frame.cpp launches it, on the primary-wave rows: the G-buffer record it stores is the reference's SurfaceData, the motion vector
    it stores is GenerateMotionVector's (MotionVectors.cu:8-55), and the DIRECT channel starts at the emitter's radiance where one is seen directly
    (ResolveDirectLightHits, GPUShadeDirect.cu:11-40) and 0 elsewhere."""
    import kat6
    r, o = kat6_pair
    h9, r9, want = kat6.hits(0)
    M, mv = kat6.motion()
    eye = np.ascontiguousarray(kat6.gold()["xeye"][0], dtype=np.uint32)
    assert np.array_equal(r9[:, :3], np.broadcast_to(eye, (kat6.N, 3)))
    g, got_mv, direct = r.TestExtract0(h9, r9[:, 3:6], eye, M)
    g = g.view(np.uint32)
    rows = np.zeros((kat6.N, 35), np.uint32)
    rows[:, 0] = g[:, 1, 3]; rows[:, 1] = g[:, 0, 3]; rows[:, 2:5] = g[:, 0, :3]; rows[:, 5:8] = g[:, 1, :3]; rows[:, 11:14] = g[:, 2, :3]; rows[:, 14:17] = g[:, 3, :3]
    rows[:, 17:20] = np.float32(1.0).view(np.uint32)                                          # transport of a primary ray (GPUGeneratePrimRay.cu:77)
    rows[:, 20:24] = g[:, 4]; rows[:, 24:28] = g[:, 5]; rows[:, 28:32] = g[:, 6]; rows[:, 32:35] = g[:, 7, :3]
    keep = [c for c in KEPT35 if not 17 <= c < 20]
    bad = np.flatnonzero((rows[:, keep] != want[:, keep]).any(axis=1))
    assert len(bad) == 0, (len(bad), int(bad[0]), int(want[bad[0], 0]), np.flatnonzero(rows[bad[0]] != want[bad[0]]).tolist())
    assert np.array_equal(np.stack([got_mv & 0xffff, got_mv >> 16], axis=1), mv)
    emit = (want[:, 0] & 1) != 0
    assert emit.sum() > 400
    assert np.array_equal(direct[emit, :3].view(np.uint32), want[emit, 20:23]) and not direct[~emit, :3].any()
    # ... and, rounded once to the binary16 the reference stores, it is what the reference's own ResolveDirectLightHits text wrote (rows xres)
    assert np.array_equal(direct[:, :3].astype(np.float16).view(np.uint16).astype(np.uint32), kat6.resolved()[:, :3])


def test_device_light_list_matches_the_reference_kernel(kat6_pair):
    """lm_k_build_lights on the scene of the rows against BuildLightDataBufferGPU's list (an atomic append on both sides: compared as sets).  Bit for bit with the oracle;
    against the rows everything but the area bit for bit, the area within 1 ulp (the rows' host compiler evaluated pow(float, int) in double: test_oracle_kat.py)."""
    import kat6
    from oracle_lib import u32ptr
    from test_oracle_kat import _sorted_rows
    r, o = kat6_pair
    want, n, _ = kat6.lights()
    assert r.TraceFrame() is True
    got = np.ascontiguousarray(r.GetLights()[0]).view(np.uint32).reshape(-1, 16)
    orc = np.zeros((n + 8, 16), np.uint32)
    assert o.L.orc_kat_light_slots(o.h, u32ptr(orc), n + 8) == n and len(got) == n
    a = _sorted_rows(got); b = _sorted_rows(want)
    assert np.array_equal(a, _sorted_rows(orc[:n]))
    assert np.array_equal(a[:, :15], b[:, :15])
    assert np.abs(a[:, 15].astype(np.int64) - b[:, 15].astype(np.int64)).max() <= 1


def test_device_contracted_bsdf_matches_the_reference_evaluate_bsdf(bare):
    """The contracted evaluation of the fast policy (lm_quick_setup + lm_quick_eval, lm_bsdf.h) against the reference's EvaluateBSDF
    (disney.cuh:320-405) on the reference-header rows of ref_kat.npz it covers, plus the depth-0 surfaces of the Resample rows
    (sheen, subsurface, grazing views) with the light direction of each row."""
    e = KAT["eval"]
    mat, N, T, wo, wi = f32(e[:, :23]), f32(e[:, 26:29]), f32(e[:, 29:32]), f32(e[:, 32:35]), f32(e[:, 35:38])
    ref = e[:, 38:42]
    g = KAT["rsmp"]                                                # surface(35) sample(14): wo = -incoming, wi = towards the light point
    d = g[:, 35 + 6:35 + 9] - g[:, 0:3]; dist = np.linalg.norm(d, axis=1); keep = dist > 0.05
    mat2, N2, T2, wo2 = f32(g[keep, 12:35]), f32(g[keep, 3:6]), f32(g[keep, 6:9]), f32(-g[keep, 9:12])
    wi2 = f32(d[keep] / dist[keep, None]); wi2 /= np.linalg.norm(wi2, axis=1, keepdims=True).astype(np.float32)
    exact2 = bare.TestBsdf(0, mat2, N2, T2, wo2, wi2)[:, :4]      # = the oracle = the reference to 2e-7 (test above)
    total = 0
    for name, (m_, n_, t_, wo_, wi_, ref_) in {"eval rows": (mat, N, T, wo, wi, ref), "resample surfaces": (mat2, N2, T2, wo2, wi2, exact2.astype(np.float64))}.items():
        q = bare.TestBsdf(2, m_, n_, t_, wo_, wi_)
        ok = q[:, 4] == 1
        # the contracted form is specified for a view direction on the lit side (a depth-0 surface seen by the camera)
        front = ok & (np.einsum("ij,ij->i", n_, wo_) > 1e-3) & np.isfinite(ref_).all(axis=1)
        err = _rel_floor(q[front, :4], ref_[front], floor=1e-4)
        rowerr = err.max(axis=1); smooth = m_[front, 15] < 0.2
        print(f"contracted EvaluateBSDF vs reference, {name}: {front.sum()} rows, max rel {err.max():.3e}, p99 {np.quantile(rowerr, 0.99):.3e}, "
              f"max over roughness >= 0.2: {rowerr[~smooth].max():.3e}")
        # (conditioning of the microfacet term on smooth surfaces: see test_device_resample_and_combine_match_the_reference_functions)
        assert front.sum() > 150 and np.quantile(rowerr, 0.99) < 2e-5 and rowerr[~smooth].max() < 1e-4 and rowerr.max() < 5e-4, (name, err.max(), int(rowerr.argmax()))
        total += front.sum()
        sheen = front & (m_[:, 18] > 0); sub = front & (m_[:, 13] > 0)
        if name == "resample surfaces":
            graz = front & (np.einsum("ij,ij->i", n_, wo_) < 0.05)
            assert sheen.sum() > 50 and sub.sum() > 50 and graz.sum() > 5, (sheen.sum(), sub.sum(), graz.sum())
    assert total > 800


def _moller_trumbore_f64(tris, org, dr, tmin, tmax, chunk=512):
    """Brute-force closest hit in float64 with the textbook Moeller-Trumbore test — shares no code and no formulation with the
    product (Woop unit-triangle packets, fp32) or the oracle.  Returns per ray (nearest t, triangle, second-nearest t)."""
    v0, e1, e2 = tris[:, 0], tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0]
    best_t = np.full(len(org), np.inf); best_i = np.full(len(org), -1, np.int64); second_t = np.full(len(org), np.inf)
    for a in range(0, len(org), chunk):
        o, d = org[a:a + chunk, None, :], dr[a:a + chunk, None, :]
        p = np.cross(d, e2[None]); det = np.einsum("rtk,tk->rt", p, e1)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / det
            tv = o - v0[None]
            u = np.einsum("rtk,rtk->rt", tv, p) * inv
            q = np.cross(tv, e1[None]); v = np.einsum("rtk,rtk->rt", np.broadcast_to(d, q.shape), q) * inv
            t = np.einsum("rtk,tk->rt", q, e2) * inv
        ok = (det != 0) & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > tmin) & (t < tmax)
        t = np.where(ok, t, np.inf)
        order = np.argsort(t, axis=1)[:, :2]
        rows = np.arange(t.shape[0])
        best_t[a:a + chunk] = t[rows, order[:, 0]]; best_i[a:a + chunk] = np.where(np.isfinite(t[rows, order[:, 0]]), order[:, 0], -1)
        if t.shape[1] > 1:
            second_t[a:a + chunk] = t[rows, order[:, 1]]
    return best_t, best_i, second_t


@pytest.mark.parametrize("n_tris,seed,extent", [(300, 11, 4.0), (6000, 12, 10.0)])
def test_closest_hit_against_an_independent_float64_moller_trumbore(n_tris, seed, extent):
    """An independent geometric check of the hit rule (not the oracle): lumen_mi_query_closest against a float64 Moeller-Trumbore brute
    force written in numpy.  Wherever the two nearest candidate hits of a ray are separated by more than 1e-6 * t the product must
    report the same triangle, t within the binary32 error bound of the plane equation (below; 95 % of the rays within 1e-5 relative, the median below 1e-6) and
    barycentrics within 1e-4; rays that miss in float64 must miss."""
    d = random_soup(n_tris, seed, extent=extent, size=1.0)
    r = product_from(d, 16, 16, 2)
    rng = np.random.default_rng(seed)
    n_rays = 6000
    org = rng.uniform(-extent, extent, (n_rays, 3)).astype(np.float32)
    dr = rng.normal(size=(n_rays, 3)).astype(np.float32); dr /= np.linalg.norm(dr, axis=1, keepdims=True)
    tmin, tmax = 0.01, 5000.0
    ip, uvt = r.QueryClosest(org, dr, tmin, tmax)
    wt = r.GetWorldTriangles().astype(np.float64).reshape(-1, 3, 3)
    assert len(wt) == d.triangle_count()
    bt, bi, st = _moller_trumbore_f64(wt, org.astype(np.float64), dr.astype(np.float64), tmin, tmax)
    hit64 = bi >= 0
    got_hit = uvt[:, 2] > 0
    with np.errstate(invalid="ignore"):
        clear = hit64 & ((st - bt) > 1e-6 * bt) & (bt > tmin * (1 + 1e-5)) & (bt < tmax * (1 - 1e-5))
    assert clear.sum() > 0.3 * n_rays                                   # most rays hit something, unambiguously
    assert got_hit[clear].all()
    # the product reports (table entry, primitive-local triangle); one primitive per instance in random_soup, the light quad comes last
    entry, prim = ip[:, 0].astype(np.int64), ip[:, 1].astype(np.int64)
    counts = np.array([len(d.primitives[pi]["indices"]) // 3 for inst in d.instances for pi in d.meshes[inst["mesh"]]])
    base = np.concatenate([[0], np.cumsum(counts)])[:-1]
    gidx = base[entry] + prim
    assert np.array_equal(gidx[clear], bi[clear])
    rel = np.abs(uvt[clear, 2].astype(np.float64) - bt[clear]) / bt[clear]
    tri = wt[bi[clear]]; o, dd = org[clear].astype(np.float64), dr[clear].astype(np.float64)
    e1, e2 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
    nrm = np.cross(e1, e2); cosang = np.abs(np.einsum("ij,ij->i", nrm, dd)) / np.linalg.norm(nrm, axis=1)
    # t = -(n.o + c) / (n.d) evaluated in binary32: the ABSOLUTE error is a few ulp of the coordinate magnitudes that enter n.o + c, divided
    # by |cos| of the incidence angle (measured: 3.3 ulp at most, tools/mt_stats.py) — a relative bound on t alone cannot hold for an origin
    # close to the plane.  6.7 ulp (4e-7) is the bar; the typical ray is far inside 1e-5 relative.
    dt = np.abs(uvt[clear, 2].astype(np.float64) - bt[clear])
    scale = np.abs(o).max(axis=1) + np.abs(tri).max(axis=(1, 2)) + bt[clear]
    assert (dt * cosang / scale).max() <= 4e-7, (dt * cosang / scale).max()
    assert np.median(rel) < 1e-6 and np.quantile(rel, 0.95) <= 1e-5, (np.median(rel), np.quantile(rel, 0.95))
    # barycentrics from float64 for the agreed triangle
    p = np.cross(dd, e2); inv = 1.0 / np.einsum("ij,ij->i", p, e1); tv = o - tri[:, 0]
    u64 = np.einsum("ij,ij->i", tv, p) * inv; v64 = np.einsum("ij,ij->i", dd, np.cross(tv, e1)) * inv
    assert np.abs(uvt[clear, 0] - u64).max() < 1e-4 and np.abs(uvt[clear, 1] - v64).max() < 1e-4
    # clear misses: no candidate within the interval in float64 and none grazing an edge
    miss64 = ~hit64
    assert (~got_hit[miss64]).mean() > 0.999                            # an fp32 edge-graze may differ for a ray in a few thousand
    r.close()


@pytest.mark.parametrize("n_tris,seed", [(1, 1), (7, 2), (500, 3), (20000, 4)])
def test_closest_and_any_hit_match_oracle(n_tris, seed):
    d = random_soup(n_tris, seed)
    r = product_from(d, 16, 16, 2); o = oracle_from(d, 16, 16, 2)
    rng = np.random.default_rng(seed + 100)
    n = 60000
    org = rng.uniform(-12, 12, (n, 3)).astype(np.float32)
    dr = rng.normal(size=(n, 3)); dr[: n // 50, rng.integers(0, 3)] = 0.0            # some axis-aligned rays
    dr = (dr / np.linalg.norm(dr, axis=1, keepdims=True)).astype(np.float32)
    ip, uvt = r.QueryClosest(org, dr)
    oip, ouvt = o.trace_closest(org, dr, use_bvh=(n_tris > 500))
    assert np.array_equal(uvt.view(np.uint32), ouvt.view(np.uint32))
    assert np.array_equal(ip, oip)
    if n_tris <= 500:                                                                 # brute force == oracle BVH == product BVH
        bip, buvt = o.trace_closest(org, dr, use_bvh=True)
        assert np.array_equal(buvt.view(np.uint32), ouvt.view(np.uint32)) and np.array_equal(bip, oip)
    tmax = rng.uniform(0.5, 30, n).astype(np.float32)
    assert np.array_equal(r.QueryAny(org, dr, tmax), o.trace_any(org, dr, tmax, use_bvh=(n_tris > 500)))
    hit = uvt[:, 2] > 0
    assert hit.any() if n_tris > 100 else True
    r.close(); o.close()


def test_empty_and_degenerate_geometry():
    from lumenrenderer_amd.scenes import SceneDescription, interleave
    d = SceneDescription()
    m = d.add_material(metallic_factor=0.0)
    pos = np.float32([[0, 0, 0], [1, 0, 0], [2, 0, 0], [0, 0, 1], [1, 0, 1], [0, 1, 1]])      # first triangle is degenerate (collinear)
    v = interleave(pos, None, np.tile(np.float32([0, 1, 0]), (6, 1)), np.tile(np.float32([1, 0, 0, 1]), (6, 1)))
    d.add_instance(d.add_mesh([d.add_primitive(v, np.uint32([0, 1, 2, 3, 4, 5]), m)]))
    r = product_from(d, 16, 16, 2); o = oracle_from(d, 16, 16, 2)
    org = np.float32([[0.3, 0.2, -5], [1.0, 0.0, -5], [5, 5, 5]]); dr = np.float32([[0, 0, 1], [0, 0, 1], [0, 0, 1]])
    ip, uvt = r.QueryClosest(org, dr); oip, ouvt = o.trace_closest(org, dr, use_bvh=False)
    assert np.array_equal(uvt.view(np.uint32), ouvt.view(np.uint32)) and np.array_equal(ip, oip)
    assert uvt[0, 2] > 0 and ip[0, 1] == 1 and uvt[2, 2] == -1.0
    assert r.TraceFrame() is False                     # no emissive triangle: frame skipped like the reference (WaveFrontRenderer.cpp:456-464)
    assert o.trace_frame() == 1
    r.close(); o.close()


def test_library_and_torch_share_one_hip_runtime_in_either_import_order():
    """torch bundles its own HIP runtime; loaded after liblumen_mi.so it used to become a second runtime that finds no GPU.
    capi.load_library binds both to one copy: a torch tensor's device pointer is usable by the library in both import orders."""
    import subprocess, sys
    body = ("r = LumenRendererMI(); r.Init(depth=2, render_resolution=(32, 32)); r.LoadSceneDescription(cornell()); assert r.TraceFrame();"
            "t = torch.zeros((32, 32, 4), dtype=torch.float32, device='cuda:0'); r.CopyRadianceToDevice(t.data_ptr()); r.Synchronize(); torch.cuda.synchronize();"
            "import numpy as np; assert np.array_equal(t.cpu().numpy().view(np.uint32), r.GetRadiance().view(np.uint32)) and float(t.abs().sum()) > 0; print('shared ok')")
    pre = "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); from helpers import cornell;" % (os.path.dirname(GOLDEN), os.path.dirname(os.path.dirname(GOLDEN)))
    for order in ("from lumenrenderer_amd import LumenRendererMI; _r0 = LumenRendererMI(); _r0.Init(depth=1, render_resolution=(8, 8)); import torch;",
                  "import torch; torch.zeros(1, device='cuda:0'); from lumenrenderer_amd import LumenRendererMI;"):
        run = subprocess.run([sys.executable, "-c", pre + order + body], capture_output=True, text=True, timeout=600)
        assert run.returncode == 0 and "shared ok" in run.stdout, (order, run.stdout[-500:], run.stderr[-2000:])


def test_c_abi_reports_bad_arguments_and_call_order():
    """Error behaviour at the boundary: what the reference asserts on (or dereferences) comes back as a status + message, and a
    failed call leaves the renderer usable.  Codes: 1 INVALID, 3 STATE (include/lumen_mi.h)."""
    import ctypes as C
    from lumenrenderer_amd import LumenRendererMI
    from lumenrenderer_amd.capi import LumenMIError

    def fails(code, fn, *a, **k):
        with pytest.raises(LumenMIError) as e:
            fn(*a, **k)
        assert e.value.code == code, (e.value.code, str(e.value))
        assert len(str(e.value)) > len("lumen_mi error 1: ")      # lumen_mi_last_error() carries a description
        return str(e.value)

    r = LumenRendererMI()
    fails(1, r.Init, depth=2, render_resolution=(0, 16))
    fails(1, r.Init, depth=2, render_resolution=(70000, 16))
    fails(1, r.Init, depth=2, render_resolution=(16, 16), device=4096)
    fails(1, r.Init, depth=17, render_resolution=(16, 16))
    r.Init(depth=2, render_resolution=(32, 32))
    fails(3, r.TraceFrame)                                             # no scene set
    fails(3, r.GetRadiance)                                            # nothing traced yet
    fails(1, r.SetDepth, 0); fails(1, r.SetDepth, 17)
    fails(1, r.SetRenderResolution, 0, 4); fails(1, r.SetOutputResolution, 4, 0)
    fails(1, r.SetWindow, 8, 8, 8, 16)
    assert "unknown tuning key" in fails(1, r.SetTuning, "no_such_key", 1)
    assert "pick_wide" in fails(1, r.SetTuning, "pick_wide", 3)
    assert "trace_blocks" in fails(1, r.SetTuning, "trace_blocks_vis", 9)
    assert "trace_blocks" in fails(1, r.SetTuning, "trace_blocks_aux", 0)                 # the waves' grid has no automatic size
    r.SetTuning("trace_blocks_main", 0); r.SetTuning("trace_blocks_vis", 0)               # 0 = chosen per frame (the default)
    fails(1, r.GetLastFrameStat, "No such stat")
    white, normal, _ = r.CreateDefaultResources()
    texs = dict(diffuse_texture=white, normal_map=normal, metallic_roughness_texture=white, emissive_texture=white, transmission_texture=white,
                clearcoat_texture=white, clearcoat_roughness_texture=white, tint_texture=white)
    fails(1, r.CreateMaterial, roughness_factor=0.0, **texs)           # WaveFrontRenderer.cpp:1283 asserts roughness > 0
    fails(1, r.CreateMaterial, **dict(texs, diffuse_texture=0xdeadbeef))
    fails(1, r.CreateMaterial, metallic_factor=0.0)                    # the eight texture slots must be filled (the reference dereferences them)
    mat = r.CreateMaterial(metallic_factor=0.0, **texs)
    tri = np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0]])
    fails(1, r.CreatePrimitive, mat, np.uint32([0, 1, 3]), positions=tri)        # index past the vertex count
    fails(1, r.CreatePrimitive, mat, np.uint32([0, 1]), positions=tri)           # fewer than three indices
    fails(1, r.CreatePrimitive, mat, np.uint32([0, 1, 2]), positions=tri, index_size=3)
    fails(1, r.CreatePrimitive, mat + 12345, np.uint32([0, 1, 2]), positions=tri)
    fails(1, r.CreateMesh, [])
    fails(1, r.CreateMesh, [mat])                                     # a material handle is not a primitive handle
    fails(1, r.CreateTexture, np.zeros((0, 0, 4), np.uint8), 0, 0)
    lib = r.lib
    assert lib.lumen_mi_trace_frame_async(None) == 1 and lib.lumen_mi_synchronize(None) == 1
    assert lib.lumen_mi_get_counters(r.h, None, 4) == 1
    w = C.c_uint32()
    assert lib.lumen_mi_get_render_resolution(r.h, C.byref(w), None) == 1
    # the handle still works: load a scene, trace, and an undersized read-back buffer is refused without touching it
    d = cornell()
    r.LoadSceneDescription(d)
    r.SetWindow(0, 0, 64, 16); fails(1, r.TraceFrame)                  # a window outside the image is reported by the frame that uses it
    r.SetWindow(0, 0, 0, 0)                                            # (0,0,0,0): the whole image again
    assert r.TraceFrame() is True
    # instances die with ILumenScene::Clear: their handles stop resolving, the slots are reused under a new generation
    old = r.m_Scene.m_MeshInstances[0]
    mesh0 = r.m_Meshes[d.instances[0]["mesh"]]
    r.m_Scene.Clear()
    fails(1, old.SetTransform, np.eye(4, dtype=np.float32))
    again = r.m_Scene.AddMesh(mesh0); again.SetTransform(d.instances[0]["transform"])
    assert again.handle != old.handle and (again.handle & 0xffffffff) <= len(d.instances) and ((again.handle >> 32) & 0xffffff) == 2   # a released slot, generation 2
    fails(1, old.SetEmissiveness, 0)
    for inst in d.instances[1:]:
        mi = r.m_Scene.AddMesh(r.m_Meshes[inst["mesh"]]); mi.SetTransform(inst["transform"]); mi.SetEmissiveness(inst["emission_mode"], inst["override_radiance"], inst["scale"])
    assert r.TraceFrame() is True
    small = np.full(8, 7.0, np.float32)
    assert lib.lumen_mi_get_radiance(r.h, small.ctypes.data_as(C.POINTER(C.c_float)), small.nbytes) == 1
    assert (small == 7.0).all() and b"too small" in lib.lumen_mi_last_error()
    o = oracle_from(d, 32, 32, 2)
    assert o.trace_frame() == 0 and o.trace_frame() == 0          # the product has completed two frames (the refused ones do not count)
    assert np.array_equal(r.GetRadiance().view(np.uint32), o.radiance().view(np.uint32))
    r.close(); o.close()


def _compare_frames(r, o, frames, check_gbuffer=True):
    for f in range(frames):
        assert r.TraceFrame() is True
        assert o.trace_frame() == 0
        got, want = r.GetRadiance(), o.radiance()
        l2 = rel_l2(got, want)
        assert l2 <= RADIANCE_TOL, (f, l2)
        mism = int(np.sum(got.view(np.uint32) != want.view(np.uint32)))
        assert mism == 0, (f, mism, l2)
        for ch in (0, 1):
            assert np.array_equal(r.GetChannel(ch).view(np.uint32), o.channel(ch).view(np.uint32)), (f, ch)
        c, s = r.GetCounters(), o.stats(24)
        assert c[0] == s[0] and c[1] == s[1] and c[2] == s[2] and c[3] == s[3], (c[:8], s[:8])
        assert np.array_equal(r.GetOutputTexturePixels(), o.output_pixels())
    if check_gbuffer:
        g, og = r.GetGBuffer(), o.gbuffer()
        assert np.array_equal(g.view(np.uint32), og.view(np.uint32))


def test_tile_copy_entry_points_move_exactly_the_rectangle_and_refuse_bad_ones():
    """lumen_mi_copy_radiance_rect_device / lumen_mi_copy_rect_device (the tile gather's frame path, kernel lm_k_copy_rect): an image rectangle of the merged radiance lands in
    a pitched device image bit for bit and nothing around it is touched; a pitched rectangle moves between two device images; rectangles outside the render window, pitches
    smaller than the rectangle and NULL pointers come back as status 1 with a message; a window-relative renderer takes IMAGE coordinates."""
    import torch
    from lumenrenderer_amd.capi import LumenMIError
    d = cornell()
    for window in (None, (8, 4, 72, 52)):
        r = product_from(d, 96, 64, 3, window=window)
        r.set_stream(torch.cuda.current_stream().cuda_stream)
        with pytest.raises(LumenMIError) as e:                              # before any frame: nothing to copy
            r.CopyRadianceRectToDevice((0, 0, 8, 8), torch.zeros(64 * 4, device="cuda").data_ptr(), 8)
        assert e.value.code == 3
        assert r.TraceFrame()
        rad = r.GetRadiance()                                               # window-shaped
        x0, y0 = (window[0], window[1]) if window else (0, 0)
        rect = (x0 + 5, y0 + 3, x0 + 37, y0 + 30)                            # image coordinates
        pitch = 40
        dst = torch.full((40, pitch, 4), -7.0, dtype=torch.float32, device="cuda")
        r.CopyRadianceRectToDevice(rect, dst[2, 3].data_ptr(), pitch)       # the rectangle's first pixel sits at row 2, column 3 of the destination
        torch.cuda.synchronize()
        got = dst.cpu().numpy()
        w, h = rect[2] - rect[0], rect[3] - rect[1]
        want = rad[rect[1] - y0: rect[3] - y0, rect[0] - x0: rect[2] - x0]
        assert np.array_equal(got[2:2 + h, 3:3 + w].view(np.uint32), want.view(np.uint32))
        mask = np.ones(got.shape[:2], bool); mask[2:2 + h, 3:3 + w] = False
        assert (got[mask] == -7.0).all()                                    # nothing outside the rectangle was written
        img = torch.zeros((50, 64, 4), dtype=torch.float32, device="cuda")
        r.CopyRectDevice(img[10, 20].data_ptr(), 64, dst[2, 3].data_ptr(), pitch, w, h)
        torch.cuda.synchronize()
        moved = img.cpu().numpy()
        assert np.array_equal(moved[10:10 + h, 20:20 + w].view(np.uint32), want.view(np.uint32))
        moved[10:10 + h, 20:20 + w] = 0.0
        assert not moved.any()
        for bad in ((rect[0], rect[1], rect[0], rect[3]), (x0 + 60, y0 + 10, x0 + 200, y0 + 20), (90, 60, 100, 70)):      # empty; past the window's right edge; past the image
            with pytest.raises(LumenMIError) as e:
                r.CopyRadianceRectToDevice(bad, dst.data_ptr(), pitch)
            assert e.value.code == 1 and "rectangle" in str(e.value)
        with pytest.raises(LumenMIError) as e:
            r.CopyRadianceRectToDevice(rect, dst.data_ptr(), w - 1)
        assert e.value.code == 1 and "pitch" in str(e.value)
        with pytest.raises(LumenMIError):
            r.CopyRectDevice(0, 64, dst.data_ptr(), pitch, w, h)
        with pytest.raises(LumenMIError):
            r.CopyRectDevice(img.data_ptr(), 4, dst.data_ptr(), pitch, w, h)
        assert r.TraceFrame()                                               # the renderer is still usable
        r.close()


def test_cornell_c1_frame_bit_exact():
    """BASELINE config C1: Cornell box 256x256, 1 path/pixel, depth 2."""
    d = cornell()
    r = product_from(d, 256, 256, 2); o = oracle_from(d, 256, 256, 2)
    _compare_frames(r, o, 1)
    rad = r.GetRadiance()
    assert rad[..., :3].max() > 0.1 and np.isfinite(rad).all()
    r.close(); o.close()


def test_plain_c_caller_of_the_abi_renders_the_oracle_pixels(tmp_path):
    """examples/render_scene.c: a C99 program that drives the whole boundary (textures, materials, primitives, meshes, scene,
    instances, camera, blended frames, read-back) without Python in between; its PPM holds the oracle's sRGB8 output."""
    import subprocess
    from helpers import build_c_example
    from lumenrenderer_amd.scenes import write_scene_file
    d = cornell()
    scene = str(tmp_path / "cornell.slm"); out = str(tmp_path / "out.ppm")
    write_scene_file(d, scene)
    exe = build_c_example(tmp_path)
    W, H, depth, frames = 80, 56, 3, 3
    run = subprocess.run([exe, scene, str(W), str(H), str(depth), str(frames), out], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, (run.stdout, run.stderr)
    raw = open(out, "rb").read()
    header = b"P6\n%d %d\n255\n" % (W, H)
    assert raw.startswith(header) and len(raw) == len(header) + W * H * 3
    got = np.frombuffer(raw[len(header):], np.uint8).reshape(H, W, 3)
    o = oracle_from(d, W, H, depth, blend=True)
    for _ in range(frames):
        assert o.trace_frame() == 0
    want = o.output_pixels()
    assert np.array_equal(got, want[..., :3])
    s = o.stats(24)
    assert "%d closest-hit rays, %d shadow rays, %d visibility rays" % (s[0], s[1], s[2]) in run.stdout
    o.close()


def test_reference_shaped_adapter_renders_the_c_example_picture(tmp_path):
    """The reference-side binding executed on the GPU: examples/sandbox_driver.cpp drives include/lumen_mi_renderer.hpp (class
    MI355X::Renderer : public LumenRenderer) through Sandbox's call sequence — Init, CreateDefaultResources, CreateTexture / Material /
    Primitive / Mesh, CreateScene + AddMesh + SetMesh, the scene's camera, StartRendering, PerformDeferredOperations per frame,
    GetOutputTexturePixels — built against the minimal interface headers of examples/sandbox_min/ (the reference tree does not exist on
    this box; the build container links the same two files against the real headers and sources).  The picture must equal, byte for
    byte, the one the plain-C caller of the C ABI renders from the same scene file — and both equal the oracle's pixels."""
    import subprocess
    from helpers import build_c_example, build_sandbox_driver
    from lumenrenderer_amd.scenes import write_scene_file
    d = cornell()
    scene = str(tmp_path / "cornell.slm"); write_scene_file(d, scene)
    cexe, xexe = build_c_example(tmp_path), build_sandbox_driver(tmp_path)
    W, H, D, F = 112, 80, 5, 3
    outs = []
    for exe, name in ((cexe, "c.ppm"), (xexe, "cpp.ppm")):
        out = str(tmp_path / name)
        run = subprocess.run([exe, scene, str(W), str(H), str(D), str(F), out], capture_output=True, text=True, timeout=300)
        assert run.returncode == 0, (run.stdout[-1000:], run.stderr[-2000:])
        outs.append(open(out, "rb").read())
        assert f"{W}x{H}" in run.stdout
    assert outs[0] == outs[1] and len(outs[0]) > W * H * 3
    o = oracle_from(d, W, H, D, blend=True)
    for _ in range(F):
        assert o.trace_frame() == 0
    want = o.output_pixels()[..., :3].tobytes()
    assert outs[1].endswith(want)
    o.close()


def _ollad_test_scene():
    """Cornell box + a UV-mapped quad whose material carries multi-texel base-colour (sRGB), normal and metal-roughness (G = 0 texels: the loader's clamp) maps:
    everything an .ollad file can express (textures typed by use, emissive MATERIALS as lights, one node per instance)."""
    from lumenrenderer_amd.scenes import interleave, generate_tangents_fast
    rng = np.random.default_rng(11)
    d = cornell()
    base = rng.integers(0, 256, (13, 9, 4), dtype=np.uint8); base[..., 3] = 255
    nrm = np.zeros((7, 5, 4), np.uint8); nrm[..., :2] = rng.integers(96, 160, (7, 5, 2)); nrm[..., 2] = 255
    mr = rng.integers(0, 256, (4, 6, 4), dtype=np.uint8); mr[::2, ::2, 1] = 0; mr[..., 3] = 255
    m = d.add_material(diffuse_color=(0.9, 0.8, 0.7, 1), metallic_factor=0.6, roughness_factor=0.7, diffuse_texture=d.add_texture(base, True),
                       normal_map=d.add_texture(nrm, False), metallic_roughness_texture=d.add_texture(mr, False))
    pos = np.float32([[-0.6, 0.4, -0.5], [0.6, 0.4, -0.5], [0.6, 1.5, -0.9], [-0.6, 1.5, -0.9]])
    uv = np.float32([[0, 0], [2.5, 0], [2.5, 1.7], [0, 1.7]])                       # > 1: wrap addressing
    n = np.cross(pos[1] - pos[0], pos[3] - pos[0]); n /= np.linalg.norm(n)
    nr = np.tile(n.astype(np.float32), (4, 1)); idx = np.uint32([[0, 1, 2], [0, 2, 3]])
    t = np.eye(4, dtype=np.float32); t[:3, 3] = (0.1, 0.0, 0.3); t[0, 0] = 0.8
    d.add_instance(d.add_mesh([d.add_primitive(interleave(pos, uv, nr, generate_tangents_fast(pos, nr, uv, idx)), idx.ravel(), m)]), t)
    return d


def test_adapter_opens_an_ollad_model_file_and_renders_the_oracle_picture(tmp_path):
    """Round 4 (VERDICT missing #4): the renderer-side model cache.  SceneManager::LoadGLTF asks the renderer FIRST (OpenCustomFileFormat, SceneManager.cpp:56-64;
    WaveFrontRenderer.cpp:1135-1146 -> LumenPTModelConverter::LoadFile).  An .ollad file written by lumenrenderer_amd/ollad.py is opened through
    MI355X::Renderer::OpenCustomFileFormat by examples/sandbox_driver.cpp — every texture / material / primitive / mesh / scene / instance then arrives through the
    adapter's LumenRenderer virtuals, base-colour map sRGB-flagged, the others not, metal-roughness G clamped, V not flipped — and the picture must be, byte for
    byte, the oracle's picture of the scene ollad.py reads back from the same file.  (This box has no reference tree: the reader is the from-scratch one of
    examples/sandbox_min/Tools/; the build container compiles the adapter against the reference's own converter, tests/test_cpu_host.py.)"""
    import subprocess
    from helpers import build_sandbox_driver
    from lumenrenderer_amd import ollad
    d = _ollad_test_scene()
    path = str(tmp_path / "scene.ollad")
    ollad.write_ollad_from_description(d, path)
    c = d.camera
    np.float32(list(c["position"]) + list(c["right"]) + list(c["up"]) + list(c["forward"]) + [c["fov"]]).tofile(path + ".cam")
    back = ollad.read_ollad(path)
    back.camera = d.camera
    W, H, D, F = 112, 80, 5, 3
    exe = build_sandbox_driver(tmp_path)
    out = str(tmp_path / "ollad.ppm")
    # asked for as the glTF the cache belongs to, as SceneManager does: the renderer replaces the extension
    run = subprocess.run([exe, str(tmp_path / "scene.gltf"), str(W), str(H), str(D), str(F), out], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, (run.stdout[-1000:], run.stderr[-2000:])
    assert f"{len(back.materials)} materials, {len(back.meshes)} meshes, {len(back.instances)} instances" in run.stdout, run.stdout
    o = oracle_from(back, W, H, D, blend=True)
    for _ in range(F):
        assert o.trace_frame() == 0
    want = o.output_pixels()[..., :3].tobytes()
    got = open(out, "rb").read()
    assert got.endswith(want), sum(a != b for a, b in zip(got[-len(want):], want))
    o.close()
    # the file round trip itself: everything comes back as written, except the loader's roughness clamp on the metal-roughness map (G >= 1, LumenPTModelConverter.cpp:121-128)
    assert len(back.textures) == len(d.textures) and len(back.primitives) == len(d.primitives)
    for a, b in zip(d.textures, back.textures):
        want_px = a["pixels"].copy()
        if a is d.textures[-1]:
            assert (want_px[..., 1] == 0).any(); want_px[..., 1] = np.maximum(want_px[..., 1], 1)
        assert np.array_equal(want_px, b["pixels"]) and a["srgb"] == b["srgb"]
    for a, b in zip(d.primitives, back.primitives):
        assert np.array_equal(a["vertices"], b["vertices"]) and np.array_equal(a["indices"], b["indices"])
    # a model without a cache and without a converter in this tree: both calls return an empty resource and the driver says so (SceneManager falls back to its own loader)
    run = subprocess.run([exe, str(tmp_path / "nothing.gltf"), str(W), str(H), str(D), str(F), out], capture_output=True, text=True, timeout=120)
    assert run.returncode == 65 and "could not open" in run.stderr


def test_adapter_start_rendering_runs_the_render_thread_and_picks_up_scene_edits(tmp_path):
    """Round 4 (VERDICT missing #5): MI355X::Renderer::StartRendering starts the render thread of the C ABI (WaveFrontRenderer.cpp:1109-1117) instead of tracing from
    PerformDeferredOperations.  The driver's main loop keeps calling PerformDeferredOperations while the thread free-runs, moves an instance half way (the
    application edits m_Transform directly, PTMeshInstance.cpp:123-178 polls it) and stops after >= 24 frames; the destructor joins the thread.  With blending off
    the last frame shows the MOVED scene: it differs from the same frame count rendered without the move, and both are lit and finite."""
    import subprocess
    from helpers import build_sandbox_driver
    from lumenrenderer_amd.scenes import write_scene_file
    d = cornell()
    scene = str(tmp_path / "cornell.slm"); write_scene_file(d, scene)
    exe = build_sandbox_driver(tmp_path)
    W, H, D, F = 96, 64, 3, 24
    heights = []
    for move in (False, True):
        out = str(tmp_path / f"thr{int(move)}.ppm")
        env = dict(os.environ, SANDBOX_THREADED="1")
        if move:
            env["SANDBOX_MOVE"] = "1"
        run = subprocess.run([exe, scene, str(W), str(H), str(D), str(F), out], capture_output=True, text=True, timeout=300, env=env)
        assert run.returncode == 0, (run.stdout[-1000:], run.stderr[-2000:])
        frames = int(run.stdout.strip().split("frame id")[-1])
        assert frames >= F, run.stdout                              # the THREAD traced them: the main loop only pushed scene state
        px = np.frombuffer(open(out, "rb").read()[-W * H * 3:], np.uint8)
        assert px.max() > 100 and (px > 0).mean() > 0.05            # the lit box (the fixture's emitter has radiance 1: most of the sRGB8 picture is dark)
        m = re.search(r"world triangles (\d+), sum of vertex heights ([-\d.]+)", run.stdout)
        assert m and int(m.group(1)) == d.triangle_count(), run.stdout
        heights.append(float(m.group(2)))
    # the edit made on the application's thread while the render thread ran reached the tracer: instance 0 sits 0.25 higher, every one of its vertices
    moved_vertices = sum(len(d.primitives[p]["indices"]) for p in d.meshes[d.instances[0]["mesh"]])
    assert abs((heights[1] - heights[0]) - 0.25 * moved_vertices) < 1e-3 * moved_vertices, (heights, moved_vertices)


def test_screenshot_of_the_output_matches_the_oracle_pixels(tmp_path):
    """OutputLayer::MakeScreenshot (Sandbox OutputLayer.cpp:882-896) over GetOutputTexturePixels: the PNG holds the oracle's
    sRGB8 output with the display gamma applied."""
    from lumenrenderer_amd import screenshot
    d = cornell()
    r = product_from(d, 96, 64, 3); o = oracle_from(d, 96, 64, 3)
    assert r.TraceFrame() is True and o.trace_frame() == 0
    path = tmp_path / "Screenshots" / "frame.png"
    written = r.MakeScreenshot(str(path))
    want = screenshot.apply_gamma(o.output_pixels(), 2.2)
    assert np.array_equal(written, want)
    assert np.array_equal(screenshot.read_png_rgba8(str(path)), want)
    assert want[..., :3].max() > 32 and (want[..., 3] == o.output_pixels()[..., 3]).all()
    r.close(); o.close()


@pytest.mark.parametrize("w,h,depth", [(37, 23, 1), (1, 1, 3), (250, 3, 2), (61, 67, 16), (128, 128, 7)])
def test_ragged_sizes_and_depth_limits(w, h, depth):
    """Window sizes that are not multiples of the 8x8 / 16x16 tiles, a single pixel, depth 1 (no indirect wave), the deepest
    path the counter block allows (16: the rays run out after 4-5 waves, so the wave loop ends early and the ReSTIR swap chain
    advances by the number of EXECUTED waves, WaveFrontRenderer.cpp:697,827) and an odd depth (the swap chain alternates)."""
    d = cornell()
    r = product_from(d, w, h, depth, blend=True); o = oracle_from(d, w, h, depth, blend=True)
    _compare_frames(r, o, 6)
    c, s = r.GetCounters(), o.stats(24)
    assert list(c[4:4 + min(depth, 16)]) == list(s[4:4 + min(depth, 16)])
    r.close(); o.close()


def test_moving_camera_and_setting_changes_between_frames():
    """Camera motion drives the motion vectors and the temporal reprojection (MotionVectors.cu, ReSTIRKernels.cu:1015-1121);
    depth, blend mode and resolution change between frames as the reference's setters allow (WaveFrontRenderer.cpp:480-505).
    Frames are enqueued asynchronously in groups, so the pipelined schedule sees every change as well."""
    d = cornell()
    r = product_from(d, 112, 80, 4, blend=True); o = oracle_from(d, 112, 80, 4, blend=True)

    def cam(k):
        a = 0.05 * k
        fwd = np.float32([np.sin(a), -0.03 * k, -np.cos(a)]); fwd /= np.linalg.norm(fwd)
        up0 = np.float32([0, 1, 0]); right = np.cross(up0, fwd); right /= np.linalg.norm(right); up = np.cross(fwd, right)
        pos = np.float32([0.08 * k, 1.0 + 0.03 * k, 3.4 - 0.1 * k])
        return pos, right.astype(np.float32), up.astype(np.float32), fwd

    def check(tag):
        r.Synchronize()
        assert np.array_equal(r.GetRadiance().view(np.uint32), o.radiance().view(np.uint32)), tag
        assert np.array_equal(r.GetGBuffer().view(np.uint32), o.gbuffer().view(np.uint32)), tag
        _, nr, mv = r.GetDenoiserInputs(); _, onr, omv = o.denoiser_inputs()
        assert np.array_equal(mv.reshape(-1, 2), omv), tag
        c, s = r.GetCounters(), o.stats(24)
        assert list(c[:4]) == list(s[:4]), (tag, c[:4], s[:4])

    for k in range(5):                                   # moving camera, frames enqueued back to back
        p = cam(k)
        r.SetCamera(*p); o.set_camera(*p)
        assert r.TraceFrameAsync(); assert o.trace_frame() == 0
        if k in (0, 2, 4):
            check(("move", k))
    assert mv_nonzero(r)
    r.SetDepth(3); o.set_depth(3)
    for k in range(2):
        assert r.TraceFrameAsync(); assert o.trace_frame() == 0
    check("depth 3")
    r.SetBlendMode(False); o.set_blend(False)
    assert r.TraceFrameAsync(); assert o.trace_frame() == 0
    r.SetBlendMode(True); o.set_blend(True)
    for k in range(2):
        p = cam(5 + k); r.SetCamera(*p); o.set_camera(*p)
        assert r.TraceFrameAsync(); assert o.trace_frame() == 0
    check("blend toggled")
    r.SetRenderResolution(90, 70); o.set_resolution(90, 70)
    for k in range(3):
        assert r.TraceFrameAsync(); assert o.trace_frame() == 0
    check("resolution changed")
    r.close(); o.close()


def mv_nonzero(r):
    _, _, mv = r.GetDenoiserInputs()
    return bool(mv.any())


def _rigid(angle_y, t):
    c, s = np.cos(angle_y), np.sin(angle_y)
    m = np.eye(4, dtype=np.float32)
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s, -s, c
    m[:3, 3] = t
    return m


def _textured_scene(seed=7):
    """Cornell box + a UV-mapped triangle soup whose material uses every texture slot with multi-texel images of different,
    non-power-of-two sizes (bilinear filtering, wrap addressing, sRGB decode of base colour / emissive: PTTexture.cpp:57-73,
    GPUExtractSurfaceData.cu:59-181), + a transmissive clear-coated material."""
    rng = np.random.default_rng(seed)
    d = cornell()
    def tex(w, h, srgb, lo=0, hi=256):
        return d.add_texture(rng.integers(lo, hi, (h, w, 4), dtype=np.uint8), srgb)
    nm = rng.integers(96, 160, (20, 24, 4), dtype=np.uint8); nm[..., 2] = 255
    mr = rng.integers(1, 256, (9, 33, 4), dtype=np.uint8)
    mat = d.add_material(diffuse_color=(0.9, 0.8, 0.7, 1.0), roughness_factor=0.7, metallic_factor=0.6,
                         diffuse_texture=tex(37, 29, True), normal_map=d.add_texture(nm, False), metallic_roughness_texture=d.add_texture(mr, False),
                         emissive_texture=tex(8, 8, True), emission=(0.0, 0.0, 0.0), transmission_texture=tex(5, 7, False),
                         clearcoat_texture=tex(6, 3, False), clearcoat_roughness_texture=tex(4, 4, False), tint_texture=tex(16, 2, False),
                         sheen_factor=0.4, sheen_tint_factor=0.5, specular_factor=0.5, specular_tint_factor=0.3, subsurface_factor=0.2)
    glass = d.add_material(diffuse_color=(0.8, 0.9, 1.0, 1.0), roughness_factor=0.2, metallic_factor=0.0, transmission_factor=0.8,
                           index_of_refraction=1.45, clearcoat_factor=0.7, clearcoat_roughness_factor=0.3, transmittance=(0.2, 0.1, 0.05),
                           tint_factor=(0.9, 0.8, 0.7), luminance=0.6, diffuse_texture=tex(13, 11, True), transmission_texture=tex(7, 5, False, 128, 256))
    glow = d.add_material(diffuse_color=(1, 1, 1, 1), emission=(2.0, 1.5, 1.0), emissive_texture=tex(12, 10, True, 64, 256))
    for k, (m, n, scale, off) in enumerate(((mat, 300, 0.11, (0.0, 1.0, 0.0)), (glass, 120, 0.08, (0.4, 0.6, 0.3)), (glow, 24, 0.04, (-0.4, 1.5, -0.2)))):
        soup = random_soup(n, seed + 10 + k, extent=6.0, size=1.0)
        p = soup.primitives[0]
        v = np.array(p["vertices"], np.float32).reshape(-1, 12).copy()
        v[:, 0:3] *= np.float32(scale)
        v[:, 3:5] = rng.uniform(-1.5, 2.5, (len(v), 2)).astype(np.float32)          # UVs outside [0,1]: wrap addressing
        d.add_instance(d.add_mesh([d.add_primitive(v, p["indices"], m)]), _rigid(0.2 * k, off))
    return d


def test_textured_materials_match_oracle():
    d = _textured_scene()
    r = product_from(d, 144, 112, 5, blend=True); o = oracle_from(d, 144, 112, 5, blend=True)
    _compare_frames(r, o, 3)
    g = r.GetGBuffer()
    assert len(np.unique(g[..., 4, :3].reshape(-1, 3), axis=0)) > 50         # many distinct albedo values: the textures are really sampled
    lights, cdf = r.GetLights(); ol, ocdf = o.lights()
    assert np.array_equal(lights.view(np.uint32), ol.view(np.uint32)) and np.array_equal(cdf.view(np.uint32), ocdf.view(np.uint32))
    r.close(); o.close()


def test_device_texture_fetch_follows_the_published_cuda_filter_rule(bare):
    """lm_tex2D (the fetch of every extraction kernel, through lumen_mi_test_tex2d) against the oracle bit for bit, and against the CUDA C Programming Guide's
    linear-filtering rule restated independently in tests/tex_rule.py (1.8 fixed-point weights, wrap by frac, sRGB decode per texel) to fp32 rounding — under both
    settings of the tuning key tex_filter (decision D6)."""
    import tex_rule
    from oracle_lib import Oracle
    from lumenrenderer_amd import LumenRendererMI
    uv = tex_rule.test_coordinates()
    for mode in (0, 1):
        r = LumenRendererMI(); r.Init(depth=2, render_resolution=(16, 16)); r.SetTuning("tex_filter", mode)
        o = Oracle(1); o.set_tex_filter(mode)
        for px, srgb in tex_rule.test_textures():
            got = r.TestTex2D(r.CreateTexture(px, normalize=srgb), uv)
            want = o.tex2d(o.add_texture(px, srgb), uv)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (mode, px.shape, int(np.sum(got.view(np.uint32) != want.view(np.uint32))))
            rule = tex_rule.guide_tex2d(px, srgb, uv, quantise=mode == 0)
            assert np.abs(got.astype(np.float64) - rule).max() <= (4e-7 if mode == 0 else 2e-5), (mode, px.shape)
        r.close(); o.close()


def test_single_texel_slots_folded_at_upload_equal_the_real_fetch():
    """A material slot with a 1x1 texture is folded into the material record when the material is created (LmDevMaterial::constMask); the
    same texel repeated over a 3x2 image is not, and goes through descriptor + bilinear fetch + sRGB table.  Both must give the same frame
    bit for bit (lerp(a, a, w) == a), in both arithmetic modes, and both equal the oracle."""
    import copy
    rng = np.random.default_rng(21)
    d = cornell()
    one = lambda srgb, lo=1, hi=256: d.add_texture(rng.integers(lo, hi, (1, 1, 4), dtype=np.uint8), srgb)
    nm = np.array([[[140, 120, 255, 255]]], np.uint8)
    mats = [d.add_material(diffuse_color=(0.9, 0.8, 0.7, 1.0), roughness_factor=0.6, metallic_factor=0.7, diffuse_texture=one(True, 160), normal_map=d.add_texture(nm, False),
                           metallic_roughness_texture=one(False, 40), emissive_texture=one(True), tint_texture=one(False), sheen_factor=0.3, specular_factor=0.4),
            d.add_material(diffuse_color=(0.8, 0.9, 1.0, 1.0), roughness_factor=0.3, metallic_factor=0.0, transmission_factor=0.7, index_of_refraction=1.4,
                           clearcoat_factor=0.6, clearcoat_roughness_factor=0.2, transmission_texture=one(False, 128), clearcoat_texture=one(False, 128),
                           clearcoat_roughness_texture=one(False), diffuse_texture=one(True, 200)),
            d.add_material(diffuse_color=(1, 1, 1, 1), emission=(2.0, 1.5, 1.0), emissive_texture=one(True, 64))]
    for k, (m, n, scale, off) in enumerate(((mats[0], 200, 0.11, (0.0, 1.0, 0.0)), (mats[1], 100, 0.08, (0.4, 0.6, 0.3)), (mats[2], 16, 0.04, (-0.4, 1.5, -0.2)))):
        p = random_soup(n, 40 + k, extent=6.0, size=1.0).primitives[0]
        v = np.array(p["vertices"], np.float32).reshape(-1, 12).copy()
        v[:, 0:3] *= np.float32(scale)
        v[:, 3:5] = rng.uniform(-1.5, 2.5, (len(v), 2)).astype(np.float32)
        d.add_instance(d.add_mesh([d.add_primitive(v, p["indices"], m)]), _rigid(0.2 * k, off))
    wide = copy.deepcopy(d)
    for t in wide.textures:
        if t["pixels"].shape[:2] == (1, 1):
            t["pixels"] = np.ascontiguousarray(np.broadcast_to(t["pixels"], (2, 3, 4)))
    for fast in (0, 1):
        ra = product_from(d, 120, 96, 5, blend=True); rb = product_from(wide, 120, 96, 5, blend=True)
        ra.SetTuning("fast_resample", fast); rb.SetTuning("fast_resample", fast)
        for _ in range(3):
            ra.TraceFrame(); rb.TraceFrame()
            assert np.array_equal(ra.GetRadiance().view(np.uint32), rb.GetRadiance().view(np.uint32))
            assert np.array_equal(ra.GetGBuffer().view(np.uint32), rb.GetGBuffer().view(np.uint32))
        ra.close(); rb.close()
    r = product_from(d, 120, 96, 5, blend=True); o = oracle_from(d, 120, 96, 5, blend=True)
    _compare_frames(r, o, 2)
    r.close(); o.close()


def test_scene_with_more_entries_and_materials_than_the_lds_tables_hold():
    """The extraction kernels stage the entry / material tables in LDS up to 128 entries / 32 materials (lm_shade.h lm_stage_tables); a scene
    beyond either limit reads them from global memory.  150 instances x 40 materials (some emissive, some textured) against the oracle, in
    every schedule that extracts surfaces (per-wave kernels, path tail)."""
    rng = np.random.default_rng(77)
    d = cornell()
    mats = []
    for k in range(40):
        tex = d.add_texture(rng.integers(32, 256, (5, 7, 4), dtype=np.uint8), True) if k % 5 == 0 else d.tex_white
        mats.append(d.add_material(diffuse_color=tuple(rng.uniform(0.2, 1.0, 3)) + (1.0,), roughness_factor=float(rng.uniform(0.2, 1.0)),
                                   metallic_factor=float(rng.uniform(0.0, 1.0)), emission=(1.5, 1.2, 0.9) if k % 13 == 0 else (0.0, 0.0, 0.0), diffuse_texture=tex))
    meshes = []
    for k in range(40):
        p = random_soup(6, 300 + k, extent=0.15, size=0.12).primitives[0]
        meshes.append(d.add_mesh([d.add_primitive(np.array(p["vertices"], np.float32), p["indices"], mats[k])]))
    for i in range(150):
        d.add_instance(meshes[i % 40], _rigid(0.37 * i, tuple(rng.uniform(-0.8, 0.8, 3) + np.array([0.0, 1.0, 0.0]))))
    for tuning in ({}, {"tail_below": 1 << 30}, {"tail_below": 0}):
        r = product_from(d, 112, 88, 6, blend=True, tuning=tuning); o = oracle_from(d, 112, 88, 6, blend=True)
        _compare_frames(r, o, 3)
        assert r.GetCounters(8)[0] > 0
        r.close(); o.close()


def test_kernel_timing_classes_and_modes():
    """lumen_mi_enable_kernel_timing: mode 1 times every class, mode 2 only the closest-hit launches and the frame; the path tail is class 5."""
    from lumenrenderer_amd.scenes import sponza_standin
    d = sponza_standin()
    r = product_from(d, 640, 360, 6, blend=True)
    for _ in range(3):
        r.TraceFrame()                                   # the tail needs the ray counts of earlier frames
    for mode, expect_all in ((1, True), (2, False)):
        r.EnableKernelTiming(mode)
        for _ in range(4):
            r.TraceFrameAsync()
        r.EnableKernelTiming(0)
        r.GetCounters(8)
        t = [r.GetKernelTime(i) for i in range(6)]
        assert t[4][1] == 4 and t[0][1] >= 4 and t[0][0] > 0.0           # four frames, at least one closest-hit launch each
        assert (t[2][1] > 0) == expect_all and (t[3][1] > 0) == expect_all  # extraction / shading and the ReSTIR passes: only in mode 1
        if mode == 1:
            assert t[5][1] == 4 and t[5][0] > 0.0                        # small window: the deep waves run as the path tail, one launch per frame
    r.close()


@pytest.mark.parametrize("fixture", ["ref_cube_textured.npz", "ref_milk_truck.npz", "ref_emissive_sphere.npz", "ref_box.npz", "ref_glass.npz"])
def test_reference_sample_assets_match_oracle(fixture):
    """Five of the reference's own sample models, geometry AND texels (ingested in the build container by tests/golden/make_textured_fixture.py): the textured cube (two 512 x 512
    maps), the Cesium milk truck (2048 x 2048 map, node hierarchy, several materials), EmissiveSphere (an emissive MATERIAL on a real mesh: 1 472 triangle lights out of FindEmissives),
    box.glb (eight materials), Glass/scene.gltf (77 124 triangles, alpha-blended materials — camera and indirect rays pass through them, shadow rays do not —, an emissive
    material worth 15 359 triangle lights: the candidate pick's global-gather path, a 14-level CDF search in the NEE).  Lit by a quad light on top of whatever they emit."""
    from lumenrenderer_amd.scenes import scene_from_npz, interleave
    d = scene_from_npz(os.path.join(GOLDEN, fixture))
    lo = np.full(3, np.inf); hi = np.full(3, -np.inf)
    for inst in d.instances:
        M = np.asarray(inst["transform"], np.float64).reshape(4, 4)
        for pi in d.meshes[inst["mesh"]]:
            v = np.asarray(d.primitives[pi]["vertices"], np.float64).reshape(-1, 12)[:, :3]
            w = v @ M[:3, :3].T + M[:3, 3]
            lo = np.minimum(lo, w.min(0)); hi = np.maximum(hi, w.max(0))
    c, e = (lo + hi) / 2, float(np.max(hi - lo))
    # a quad light above and in front of the model, facing down
    y = hi[1] + 0.6 * e
    quad = np.float32([[c[0] - e, y, c[2] - e], [c[0] + e, y, c[2] - e], [c[0] + e, y, c[2] + e], [c[0] - e, y, c[2] + e]])
    v = interleave(quad, None, np.tile(np.float32([0, -1, 0]), (4, 1)), np.tile(np.float32([1, 0, 0, 1]), (4, 1)))
    m = d.add_material(diffuse_color=(1, 1, 1, 1), emission=(1.0, 1.0, 1.0))     # emissive material: the frame is skipped when the
    # primitives' material-based light count is zero (WaveFrontRenderer.cpp:456-464), whatever the instance override says
    d.add_instance(d.add_mesh([d.add_primitive(v, np.uint32([0, 2, 1, 0, 3, 2, 0, 1, 2, 0, 2, 3]), m)]), None, emission_mode=2, override_radiance=(6.0, 5.5, 5.0), scale=1.0)
    eye = c + np.float64([0.35 * e, 0.25 * e, 1.3 * e])
    fwd = c - eye; fwd /= np.linalg.norm(fwd)
    right = np.cross([0.0, 1.0, 0.0], fwd); right /= np.linalg.norm(right); up = np.cross(fwd, right)
    d.set_camera(eye, right, up, fwd, 70.0)
    r = product_from(d, 160, 120, 4, blend=True); o = oracle_from(d, 160, 120, 4, blend=True)
    _compare_frames(r, o, 2)
    g = r.GetGBuffer()
    assert (g[..., 0, 3] > 0).mean() > 0.15                                  # the model covers a good part of the image
    r.close(); o.close()


def test_denoiser_inputs_match_oracle():
    """depth / normal-roughness / motion exports (GPUExtractNRD_DLSSdata.cu, GPUExtractDepthData.cu)."""
    from lumenrenderer_amd.scenes import sponza_standin
    for d, (w, h) in ((cornell(), (96, 64)), (sponza_standin(), (160, 90))):
        r = product_from(d, w, h, 3, blend=True); o = oracle_from(d, w, h, 3, blend=True)
        for _ in range(2):
            assert r.TraceFrame() and o.trace_frame() == 0
        depth, nr, mv = r.GetDenoiserInputs(0.1, 1000.0); od, onr, omv = o.denoiser_inputs(0.1, 1000.0)
        assert np.array_equal(depth.ravel().view(np.uint32), od.view(np.uint32))
        assert np.array_equal(nr.reshape(-1, 4), onr) and np.array_equal(mv.reshape(-1, 2), omv)
        assert depth.max() > 0 and nr.any()
        r.close(); o.close()


def test_fp16_quantised_radiance_report():
    """beside the fp32 contract, the result as the reference would STORE it — its pixel buffers are half4
    (GPUMergeOutputChannels.cu:5-88).  lumen_mi_get_radiance_half4 rounds the merged fp32 radiance once to binary16 (round to nearest
    even, the vendored __float2half pinned by ref_kat.npz rows "half"); it equals the oracle's radiance rounded the same way bit for bit,
    and the quantisation alone costs <= 2^-11 relative per pixel — half of the 1e-3 tolerance, which is why fp32 is the contract."""
    for d, (w, h, depth, frames) in ((cornell(), (160, 120, 5, 3)), (_textured_scene(), (144, 112, 4, 2))):
        r = product_from(d, w, h, depth, blend=True); o = oracle_from(d, w, h, depth, blend=True)
        for _ in range(frames):
            assert r.TraceFrame() and o.trace_frame() == 0
        got = r.GetRadianceHalf4()
        want32 = o.radiance()
        with np.errstate(over="ignore"):
            want = want32.astype(np.float16)
        assert np.array_equal(got.view(np.uint16), want.view(np.uint16))
        L = orc_lib()                                             # ... and numpy's rounding is the pinned conversion
        sample = want32.reshape(-1)[:: max(1, want32.size // 5000)]
        assert [L.orc_f32_to_f16(float(x)) for x in sample] == want.reshape(-1)[:: max(1, want32.size // 5000)].view(np.uint16).tolist()
        fin = np.isfinite(got.astype(np.float32))
        rel = np.abs(got.astype(np.float32)[fin] - want32[fin]) / np.maximum(np.abs(want32[fin]), 6.2e-5)      # below the smallest normal half the step is absolute
        assert rel.max() <= 2.0 ** -11 + 1e-7
        err = rel_l2(got.astype(np.float32)[..., :3], want32[..., :3])
        print(f"fp16-quantised radiance: rel-L2 vs fp32 {err:.3e}")
        assert 1e-5 < err < 5e-4
        r.close(); o.close()


def test_fp16_blend_chain_as_the_reference_stores_it_report():
    """Decision D1 on the blend, as a number (VERDICT r4 missing #4): the reference blends in binary16 (GPUMergeOutputChannels.cu:53-72, Half4.h:105-196), this build in fp32
    with one rounding on export.  Per-frame fp32 channels of the product (blending off) replayed through both chains on the host (tools/fp16_blend_chain.py; numpy float16 is
    correctly rounded per operation = the best case of __hmul2 / __hadd2 / __h2div): the two stay within a few binary16 ulp of each other — 4 frames ~ 5e-4, 8 frames
    ~ 7e-4 relative L2 — i.e. the fp16 chain alone uses up most of the 1e-3 tolerance, which is why fp32 is the contract."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from fp16_blend_chain import chains, rel_l2_finite
    from lumenrenderer_amd.scenes import sponza_standin
    r = product_from(sponza_standin(), 640, 360, 6, blend=False)
    fd, fi = [], []
    for _ in range(8):
        assert r.TraceFrame()
        fd.append(r.GetChannel(0).copy()); fi.append(r.GetChannel(1).copy())
    r.close()
    c = chains(fd, fi)
    errs = {n: rel_l2_finite(c[n][0], c[n][1])[0] for n in (1, 4, 8)}
    print("fp32 blend rounded once vs binary16 blend chain, rel-L2 by blended frames:", {n: f"{e:.3e}" for n, e in errs.items()})
    assert errs[1] < 6e-4 and errs[4] < 2e-3 and errs[8] < 2e-3, errs          # frame 1: two roundings of the channels against one of the sum
    assert errs[4] > 1e-5 and errs[8] > 1e-5, errs                             # ... and the chains do differ: the report is not vacuous


def test_cornell_blended_frames_depth5():
    d = cornell()
    r = product_from(d, 160, 120, 5, blend=True); o = oracle_from(d, 160, 120, 5, blend=True)
    _compare_frames(r, o, 4)                             # "4 spp" = 4 blended TraceFrame()s; exercises temporal reuse
    r.close(); o.close()


def test_cornell_window_matches_oracle_window():
    d = cornell()
    win = (32, 16, 160, 112)
    r = product_from(d, 192, 128, 3, window=win); o = oracle_from(d, 192, 128, 3, window=win)
    for _ in range(2):
        assert r.TraceFrame() and o.trace_frame() == 0
        want = o.radiance()[win[1]:win[3], win[0]:win[2]]
        assert np.array_equal(r.GetRadiance().view(np.uint32), np.ascontiguousarray(want).view(np.uint32))
    r.close(); o.close()


def test_unaligned_window_async_frames():
    """A render window whose origin and size are not multiples of the 8 / 16 pixel tiles (what an odd tile grid produces),
    several frames enqueued back to back: tile schedule (path tail, pick-ahead) on a window with halo semantics."""
    d = cornell()
    win = (37, 19, 151, 103)
    r = product_from(d, 192, 128, 4, blend=True, window=win); o = oracle_from(d, 192, 128, 4, blend=True, window=win)
    for _ in range(4):
        assert r.TraceFrameAsync() and o.trace_frame() == 0
    r.Synchronize()
    want = o.radiance()[win[1]:win[3], win[0]:win[2]]
    assert np.array_equal(r.GetRadiance().view(np.uint32), np.ascontiguousarray(want).view(np.uint32))
    r.close(); o.close()


def test_sponza_standin_small_frame_matches_oracle():
    from lumenrenderer_amd.scenes import sponza_standin
    d = sponza_standin()
    r = product_from(d, 192, 108, 6, blend=True); o = oracle_from(d, 192, 108, 6, blend=True)
    _compare_frames(r, o, 2, check_gbuffer=True)
    assert r.GetBvhInfo()["triangles"] == d.triangle_count()
    r.close(); o.close()


def test_sponza_with_1024_emissive_triangles_matches_oracle():
    from lumenrenderer_amd.scenes import sponza_standin
    d = sponza_standin(extra_lights=512)
    r = product_from(d, 128, 72, 4); o = oracle_from(d, 128, 72, 4)
    _compare_frames(r, o, 1)
    lights, cdf = r.GetLights(); ol, ocdf = o.lights()
    assert np.array_equal(lights.view(np.uint32), ol.view(np.uint32)) and np.array_equal(cdf.view(np.uint32), ocdf.view(np.uint32))
    r.close(); o.close()


TUNINGS = [{"tail_below": 0}, {"tail_below": 1 << 30}, {"tail_below": 6000}, {"tail_below": 6000, "tail_lanes": 64},
           {"single_stream": 1, "tail_below": 0}, {"single_stream": 1}, {"refill": 0, "tail_below": 0},
           {"pick_ahead": 0}, {"pick_ahead": 1, "tail_below": 0}, {"shadow_on_wave": 1}, {"shadow_on_wave": 1, "tail_below": 0},
           {"sort_rays": 1}, {"sort_rays": 16, "tail_below": 0}, {"sort_rays": 2, "single_stream": 1, "tail_below": 6000},
           {"packet_primary": 1}, {"packet_primary": 1, "single_stream": 1, "tail_below": 0},
           {"tail_pair": 1, "tail_below": 1 << 30}, {"tail_pair": 1, "tail_below": 6000, "tail_lanes": 16}, {"tail_pair": 1, "tail_lanes": 64, "single_stream": 1, "tail_below": 1 << 30},
           {"tail_pair": 1, "tail_lanes": 5, "tail_below": 1 << 30, "wave_streams": 2},
           {"packet_visibility": 1}, {"packet_visibility": 1, "packet_primary": 1, "pick_ahead": 0}, {"packet_visibility": 0, "packet_primary": 0},
           {"wave_streams": 2}, {"wave_streams": 2, "tail_below": 0}, {"wave_streams": 2, "pick_ahead": 0}, {"wave_streams": 2, "tail_below": 6000, "pick_ahead": 0},
           {"fuse_primary": 1}, {"fuse_primary": 1, "packet_primary": 1, "wave_streams": 2}, {"fuse_primary": 1, "packet_primary": 1, "single_stream": 1},
           {"lazy_reuse": 1}, {"lazy_reuse": 0}, {"lazy_reuse": 1, "wave_streams": 2}, {"lazy_reuse": 1, "single_stream": 1}, {"lazy_reuse": 1, "pick_ahead": 0, "tail_below": 0},
           {"lazy_reuse": 1, "pick_ahead": 1, "wave_streams": 2, "tail_below": 6000},
           {"tail_repack": 1, "tail_below": 1 << 30}, {"tail_repack": 1, "tail_below": 6000}, {"tail_repack": 1, "tail_below": 1 << 30, "single_stream": 1, "fast_shade": 0},
           {"gpu_build": 1}, {"gpu_build": 1, "packet_primary": 1, "tail_below": 0},
           {"trace_blocks_main": 8, "trace_blocks_vis": 8}, {"trace_blocks_main": 4, "trace_blocks_vis": 4, "lazy_reuse": 1}, {"trace_blocks_main": 1, "trace_blocks_vis": 3, "trace_blocks_aux": 2},
           {"trace_blocks_main": 2, "trace_blocks_aux": 5, "wave_streams": 2, "tail_below": 0},
           {"fuse_combine": 0}, {"fuse_combine": 0, "lazy_reuse": 0, "pick_ahead": 0}, {"fuse_combine": 1, "lazy_reuse": 0, "pick_ahead": 0}, {"fuse_combine": 1, "lazy_reuse": 0, "single_stream": 1}]
DEEP = [{"tail_repack": 1, "tail_below": 1 << 30}, {"tail_repack": 1}, {"lazy_reuse": 0}, {"lazy_reuse": 1, "wave_streams": 2}, {"lazy_reuse": 1, "pick_ahead": 0, "single_stream": 1}, {}, {"packet_visibility": 1}, {"tail_pair": 1, "tail_below": 1 << 30}, {"tail_pair": 1}, {"pick_ahead": 0, "tail_below": 0}, {"pick_ahead": 1, "tail_below": 1 << 30}, {"single_stream": 1}, {"shadow_on_wave": 1}, {"sort_rays": 16, "tail_below": 0}, {"wave_streams": 2}]


@pytest.mark.parametrize("tuning", DEEP, ids=lambda t: ",".join(f"{k}={v}" for k, v in t.items()) or "default")
def test_early_wave_loop_exit_under_every_schedule(tuning):
    """Depth 16 on a small image: every frame executes a different number of waves (the queue runs empty), so the swap chain
    parity and the age of the 'previous' reservoirs change from frame to frame."""
    d = cornell()
    r = product_from(d, 61, 67, 16, blend=True, tuning=tuning); o = oracle_from(d, 61, 67, 16, blend=True)
    _compare_frames(r, o, 7, check_gbuffer=False)
    r.close(); o.close()



@pytest.mark.parametrize("tuning", TUNINGS, ids=lambda t: ",".join(f"{k}={v}" for k, v in t.items()))
def test_schedules_do_not_change_results(tuning):
    """Per-wave kernels vs the one-launch path tail (chosen on the device from the queue length), stream overlap on/off,
    lane refill: every schedule must reproduce the oracle bit for bit, counters included."""
    from lumenrenderer_amd.scenes import sponza_standin
    for d, (w, h, depth), frames in ((cornell(), (160, 120, 5), 3), (sponza_standin(), (192, 108, 6), 2)):
        r = product_from(d, w, h, depth, blend=True, tuning=tuning); o = oracle_from(d, w, h, depth, blend=True)
        _compare_frames(r, o, frames, check_gbuffer=False)
        c, s = r.GetCounters(), o.stats(24)
        assert list(c[4:4 + depth]) == list(s[4:4 + depth]), (c[4:4 + depth], s[4:4 + depth])     # rays per wave
        r.close(); o.close()


def test_pipelined_async_frames_match_oracle():
    """Frames enqueued back to back (TraceFrameAsync, one Synchronize at the end) run software-pipelined on four streams:
    the next frame's front overlaps the ReSTIR tail of the current one.  The blended result must still be the oracle's."""
    from lumenrenderer_amd.scenes import sponza_standin
    for d, (w, h, depth), frames in ((cornell(), (160, 120, 5), 5), (sponza_standin(), (192, 108, 6), 4), (cornell(), (96, 64, 3), 3)):
        r = product_from(d, w, h, depth, blend=True); o = oracle_from(d, w, h, depth, blend=True)
        for _ in range(frames):
            assert r.TraceFrameAsync() is True
            assert o.trace_frame() == 0
        r.Synchronize()
        assert np.array_equal(r.GetRadiance().view(np.uint32), o.radiance().view(np.uint32))
        for ch in (0, 1):
            assert np.array_equal(r.GetChannel(ch).view(np.uint32), o.channel(ch).view(np.uint32)), ch
        c, s = r.GetCounters(), o.stats(24)
        assert list(c[:4]) == list(s[:4])
        assert np.array_equal(r.GetGBuffer().view(np.uint32), o.gbuffer().view(np.uint32))
        assert np.array_equal(r.GetOutputTexturePixels(), o.output_pixels())
        r.close(); o.close()


def test_deep_traversal_stack_spills_to_global_memory():
    """A stack of large overlapping sheets seen edge-on keeps many 4-wide-node children pending at once: the per-lane stack
    grows past its LDS part (LM_STACK_LDS = 16) and uses the global spill area.  Hits must still equal brute force."""
    from lumenrenderer_amd.scenes import SceneDescription, interleave
    rng = np.random.default_rng(11)
    n_sheets = 3000
    pos = np.zeros((n_sheets * 3, 3), np.float32)
    z = np.linspace(0.0, 30.0, n_sheets, dtype=np.float32)
    for k in range(n_sheets):                            # long thin triangles, all crossing the same corridor along +z
        a = rng.uniform(-1, 1, 2).astype(np.float32) * 0.2
        pos[3 * k + 0] = (-40 + a[0], -40 + a[1], z[k])
        pos[3 * k + 1] = (40 + a[0], -40 + a[1], z[k] + 0.004)
        pos[3 * k + 2] = (a[0], 40 + a[1], z[k] + 0.002)
    d = SceneDescription()
    m = d.add_material(emission=(1.0, 1.0, 1.0))
    v = interleave(pos, None, np.tile(np.float32([0, 0, -1]), (len(pos), 1)), np.tile(np.float32([1, 0, 0, 1]), (len(pos), 1)))
    d.add_instance(d.add_mesh([d.add_primitive(v, np.arange(len(pos), dtype=np.uint32), m)]))
    r = product_from(d, 16, 16, 2); o = oracle_from(d, 16, 16, 2)
    nr = 4096
    org = np.concatenate([rng.uniform(-30, 30, (nr, 2)), rng.uniform(-5, 35, (nr, 1))], axis=1).astype(np.float32)
    dr = rng.normal(size=(nr, 3)).astype(np.float32); dr[:, 2] *= 4.0
    dr /= np.linalg.norm(dr, axis=1, keepdims=True)
    ip, uvt = r.QueryClosest(org, dr); oip, ouvt = o.trace_closest(org, dr, use_bvh=False)
    assert np.array_equal(uvt.view(np.uint32), ouvt.view(np.uint32)) and np.array_equal(ip, oip)
    assert (uvt[:, 2] > 0).sum() > nr // 4
    tmax = np.full(nr, 100.0, np.float32)
    assert np.array_equal(r.QueryAny(org, dr, tmax=tmax), o.trace_any(org, dr, tmax))
    info = r.GetBvhInfo()
    assert info["triangles"] == n_sheets
    r.close(); o.close()


@pytest.mark.parametrize("refit", [1, 0], ids=["gpu-refit", "host-rebuild"])
def test_moving_instances_refit_matches_oracle(refit):
    """instance transforms change between frames (PTMeshInstance.cpp:123-178).  The product refits its BVH on
    the GPU (re-transformed triangles, recomputed Woop packets, boxes propagated bottom-up); the oracle rebuilds everything.
    Ray queries and blended frames (temporal reuse across the move) must stay bit-identical."""
    from lumenrenderer_amd.scenes import SceneDescription
    rng = np.random.default_rng(5)
    soup = random_soup(4000, 9, extent=6.0, size=0.6)
    d = cornell()
    # second mesh instance: the soup's primitive placed inside the Cornell box, moved every frame
    mat = d.add_material(diffuse_color=(0.7, 0.5, 0.3, 1.0), roughness_factor=0.6, metallic_factor=0.0)
    p = soup.primitives[0]
    v = np.array(p["vertices"], np.float32).reshape(-1, 12).copy(); v[:, 0:3] *= np.float32(0.08)
    prim = d.add_primitive(v, p["indices"], mat)
    inst = d.add_instance(d.add_mesh([prim]), _rigid(0.0, (0.0, 1.0, 0.0)))
    r = product_from(d, 128, 96, 4, blend=True, tuning={"refit": refit}); o = oracle_from(d, 128, 96, 4, blend=True)
    nr = 3000
    org = rng.uniform(-0.9, 0.9, (nr, 3)).astype(np.float32); org[:, 1] += 1.0
    dr = rng.normal(size=(nr, 3)).astype(np.float32); dr /= np.linalg.norm(dr, axis=1, keepdims=True)
    for step in range(4):
        m = _rigid(0.37 * step, (0.25 * np.sin(step), 1.0 + 0.1 * step, 0.2 * step - 0.3))
        r.m_Scene.m_MeshInstances[inst].SetTransform(m); o.set_instance_transform(inst, m)
        ip, uvt = r.QueryClosest(org, dr); oip, ouvt = o.trace_closest(org, dr, use_bvh=False)
        assert np.array_equal(uvt.view(np.uint32), ouvt.view(np.uint32)) and np.array_equal(ip, oip), step
        assert np.array_equal(r.QueryAny(org, dr, tmax=np.full(nr, 3.0, np.float32)), o.trace_any(org, dr, np.full(nr, 3.0, np.float32))), step
        assert r.TraceFrame() is True and o.trace_frame() == 0
        assert np.array_equal(r.GetRadiance().view(np.uint32), o.radiance().view(np.uint32)), step
        assert np.array_equal(r.GetGBuffer().view(np.uint32), o.gbuffer().view(np.uint32)), step
    r.close(); o.close()


@pytest.mark.parametrize("assemble", [1, 0], ids=["assembled", "host-rebuild"])
def test_instances_added_and_removed_between_frames(assemble):
    """Topology edits (the reference rebuilds its instance acceleration structure, PTScene.cpp:74-156): instances are
    added, moved and the scene is cleared and refilled between blended frames.  After the first (SAH) build the product assembles
    the scene tree from cached per-mesh trees + a top tree and refits on the GPU (lm_assemble_bvh); the hit rule does not depend on
    the tree, so every frame still equals the oracle bit for bit, and the ray queries equal the oracle's brute force."""
    soup = random_soup(600, 21, extent=5.0, size=0.6)
    d = cornell()
    p = soup.primitives[0]
    v = np.array(p["vertices"], np.float32).reshape(-1, 12).copy(); v[:, 0:3] *= np.float32(0.06)
    mat = d.add_material(diffuse_color=(0.3, 0.6, 0.7, 1.0), roughness_factor=0.6, metallic_factor=0.0)
    extra_mesh = d.add_mesh([d.add_primitive(v, p["indices"], mat)])
    W, H, D = 96, 72, 3
    r = product_from(d, W, H, D, blend=True, tuning={"assemble": assemble}); o = oracle_from(d, W, H, D, blend=True)
    _compare_frames(r, o, 2, check_gbuffer=False)                                # first build: full SAH
    meshes = r.m_Meshes
    added = []
    for k in range(4):                                                           # add four copies of the extra mesh, one per frame
        xf = _rigid(0.4 * k, (0.35 * (k - 1.5), 0.6 + 0.25 * k, 0.2 * k))
        added.append(r.m_Scene.AddMesh(meshes[extra_mesh])); added[-1].SetTransform(xf)
        o.add_instance(extra_mesh, xf)
        _compare_frames(r, o, 1, check_gbuffer=False)
    # a brand-new mesh while frames are being rendered: only ITS tree is built (and the vertex / index pools grow)
    soup2 = random_soup(150, 77, extent=4.0, size=0.9).primitives[0]
    v2 = np.array(soup2["vertices"], np.float32).reshape(-1, 12).copy(); v2[:, 0:3] *= np.float32(0.05)
    prim2, _ = r.CreatePrimitive(r.m_Materials[mat], soup2["indices"], vertices=v2)
    mesh2 = r.CreateMesh([prim2])
    xf2 = _rigid(-0.7, (-0.3, 1.4, 0.1))
    r.m_Scene.AddMesh(mesh2).SetTransform(xf2)
    o.add_instance(o.add_mesh([o.add_primitive(v2, soup2["indices"], mat)]), xf2)
    _compare_frames(r, o, 2, check_gbuffer=False)
    xf = _rigid(1.3, (0.1, 1.2, -0.3))                                           # move one of them: a refit of the assembled tree
    added[1].SetTransform(xf); o.set_instance_transform(len(d.instances) + 1, xf)
    _compare_frames(r, o, 2, check_gbuffer=False)
    rng = np.random.default_rng(8)
    org = rng.uniform(-0.8, 0.8, (300, 3)).astype(np.float32) + np.float32([0, 1, 2.5])
    dr = rng.normal(size=(300, 3)).astype(np.float32); dr /= np.linalg.norm(dr, axis=1, keepdims=True)
    ip, uvt = r.QueryClosest(org, dr); oip, ouvt = o.trace_closest(org, dr, use_bvh=False)
    assert np.array_equal(uvt.view(np.uint32), ouvt.view(np.uint32)) and np.array_equal(ip, oip)
    assert np.array_equal(r.QueryAny(org, dr, np.full(300, 4.0, np.float32)), o.trace_any(org, dr, np.full(300, 4.0, np.float32), use_bvh=False))
    assert np.array_equal(r.GetWorldTriangles().view(np.uint32), o.world_triangles().view(np.uint32))
    c = r.GetCounters(52)
    assert (c[51] >= 5) if assemble else (c[51] == 0)
    r.close(); o.close()


def test_set_scene_before_every_frame_is_not_a_scene_edit():
    """The C++ adapter calls lumen_mi_set_scene before every TraceFrame (include/lumen_mi_renderer.hpp TraceFrame; the reference
    re-reads m_Scene each frame, WaveFrontRenderer.cpp:443-449).  An unchanged handle must not re-flatten / refit / re-assemble:
    the refit and assembly counters stay where they were and the frames equal the oracle's."""
    d = cornell()
    r = product_from(d, 96, 64, 3, blend=True); o = oracle_from(d, 96, 64, 3, blend=True)
    _compare_frames(r, o, 1, check_gbuffer=False)
    before = r.GetCounters(52)
    for _ in range(4):
        r.SetScene(r.m_Scene)
        _compare_frames(r, o, 1, check_gbuffer=False)
    after = r.GetCounters(52)
    assert (after[50], after[51]) == (before[50], before[51])
    r.close(); o.close()


def test_emissiveness_and_override_material_changes_between_frames():
    """MeshInstance::SetEmissiveness / SetOverrideMaterial between frames (MeshInstance.h:57-98, PTMeshInstance.cpp:123-178):
    the product refreshes its scene data table and light list without rebuilding the BVH; the oracle rebuilds everything."""
    soup = random_soup(600, 21, extent=6.0, size=0.8)
    d = cornell()
    red = d.add_material(diffuse_color=(0.8, 0.1, 0.1, 1.0), roughness_factor=0.4, metallic_factor=0.0)
    grey = d.add_material(diffuse_color=(0.5, 0.5, 0.5, 1.0), roughness_factor=0.9, metallic_factor=0.0)
    p = soup.primitives[0]
    v = np.array(p["vertices"], np.float32).reshape(-1, 12).copy(); v[:, 0:3] *= np.float32(0.07)
    inst = d.add_instance(d.add_mesh([d.add_primitive(v, p["indices"], grey)]), _rigid(0.3, (0.0, 1.0, 0.0)))
    r = product_from(d, 112, 84, 4, blend=True); o = oracle_from(d, 112, 84, 4, blend=True)
    mi = r.m_Scene.m_MeshInstances[inst]
    steps = [lambda: None,
             lambda: (mi.SetEmissiveness(2, (3.0, 2.0, 1.0), 0.5), o.set_instance_emissiveness(inst, 2, (3.0, 2.0, 1.0), 0.5)),   # OVERRIDE: the soup glows
             lambda: (mi.SetEmissiveness(2, (1.0, 4.0, 1.0), 2.0), o.set_instance_emissiveness(inst, 2, (1.0, 4.0, 1.0), 2.0)),
             lambda: (mi.SetOverrideMaterial(r.m_Materials[red]), o.set_instance_override_material(inst, red)),
             lambda: (mi.SetEmissiveness(1, (0.0, 0.0, 0.0), 1.0), o.set_instance_emissiveness(inst, 1, (0.0, 0.0, 0.0), 1.0)),   # DISABLED
             lambda: (mi.SetTransform(_rigid(0.9, (0.2, 1.1, -0.2))), o.set_instance_transform(inst, _rigid(0.9, (0.2, 1.1, -0.2)))),
             # a null override falls back to the mesh's own materials (PTMeshInstance.cpp:163-165): handle 0 clears it
             lambda: (mi.SetOverrideMaterial(0), o.set_instance_override_material(inst, -1))]
    lights = []
    for k, step in enumerate(steps):
        step()
        assert r.TraceFrameAsync() and o.trace_frame() == 0
        assert r.TraceFrameAsync() and o.trace_frame() == 0
        r.Synchronize()
        assert np.array_equal(r.GetRadiance().view(np.uint32), o.radiance().view(np.uint32)), k
        assert np.array_equal(r.GetGBuffer().view(np.uint32), o.gbuffer().view(np.uint32)), k
        c, s_ = r.GetCounters(), o.stats(24)
        assert list(c[:4]) == list(s_[:4]), (k, c[:4], s_[:4])
        lights.append(c[3])
    assert lights[1] > lights[0] and lights[4] == lights[0]            # the override adds lights, DISABLED removes them again
    assert r.GetBvhInfo()["triangles"] == d.triangle_count()
    r.close(); o.close()


def _orbit(c0, k, turn=0.06, rise=0.03, slide=0.05):
    """camera step k around a description's own camera c0: turns, looks up and slides sideways"""
    f0, r0, u0 = np.float32(c0["forward"]), np.float32(c0["right"]), np.float32(c0["up"])
    fwd = np.cos(turn * k) * f0 + np.sin(turn * k) * r0 + np.float32(rise * k) * u0; fwd /= np.linalg.norm(fwd)
    right = np.cross(u0, fwd); right /= np.linalg.norm(right)
    if np.dot(right, r0) < 0: right = -right
    up = np.cross(fwd, right)
    if np.dot(up, u0) < 0: up = -up
    return np.float32(c0["position"]) + np.float32(slide * k) * r0, right.astype(np.float32), up.astype(np.float32), fwd.astype(np.float32)


@pytest.mark.parametrize("case", ["cornell-exact", "textured-fast", "sponza-exact", "sponza-fast"])
def test_history_passes_run_only_when_their_result_can_be_read(case):
    """Lazy reuse (tuning key lazy_reuse, frame.cpp): both spatial passes and CombineReservoirBuffers (ReSTIR.cpp:181-233) only build the reservoirs a LATER
    temporal pass reads as "previous".  They are launched at the start of the next frame's ReSTIR chain and run only if the swap chain has turned (an odd
    number of executed waves).  If it has not — every frame of an even path depth: the reference's swap quirk (one SwapBuffers per wave,
    WaveFrontRenderer.cpp:827) leaves that history unread — they are dropped, and only the entries that outlive the next candidate pick (pixels flagged in the
    next frame, where the pick zeroes the weight and nothing else) get the one field a later frame can observe: the sample count the combine would have left.
    Every image must equal, bit for bit, that of the renderer which launches the passes with their frame (lazy_reuse 0), through camera motion (pixels turn
    into emitter / miss pixels and back), path depths changing between even and odd (entries read frames later through a probe plane of another age) and
    frames enqueued back to back; and leaving the count completion out (lazy_reuse 2, test only) must show, so that this test cannot pass vacuously.
    Counter [54] = deferred executions of the passes, [55] = entries completed."""
    from lumenrenderer_amd.scenes import sponza_standin
    scene, mode = case.split("-")
    fast = int(mode == "fast")
    d = {"cornell": cornell, "textured": _textured_scene, "sponza": sponza_standin}[scene]()      # textured: glass and clear coat, the fast mode's second (exact) launch runs too
    W, H = (240, 136) if scene == "sponza" else (120, 88)
    step = (lambda k: _orbit(d.camera, k)) if scene == "sponza" else (lambda k: _orbit(d.camera, k, 0.07, -0.02, 0.1))
    # (depth, camera step) per frame
    plan = [(4, 0), (4, 3), (3, 3), (4, 0), (4, 0), (4, 0), (4, 2), (4, 1), (5, 1), (6, 0), (6, 0)] + [(4, 0)] * 3 + [(4, k) for k in range(1, 5)] + \
           [(3, 4), (3, 5), (4, 5), (4, 6), (5, 6), (4, 6), (4, 6), (6, 7), (6, 7), (3, 7), (4, 7), (4, 7), (4, 7)]

    def run(lazy, sync_every):
        r = product_from(d, W, H, 4, blend=False, tuning={"lazy_reuse": lazy, "fast_resample": fast})
        images, ran, fixed, waves = [], [], [], []
        for f, (depth, k) in enumerate(plan):
            r.SetDepth(depth); r.SetCamera(*step(k))
            assert r.TraceFrameAsync()
            if f % sync_every == sync_every - 1 or f == len(plan) - 1:
                r.Synchronize(); images.append(r.GetRadiance().copy())
            if sync_every == 1:
                c = r.GetCounters(64); ran.append(c[54]); fixed.append(c[55]); waves.append(sum(1 for x in c[4:4 + depth] if x))
        r.close()
        return images, ran, fixed, waves

    eager, ran0, fixed0, waves = run(0, 1)
    assert ran0[-1] == 0 and fixed0[-1] == 0
    lazy, ran, fixed, _ = run(1, 1)
    for f in range(len(plan)):
        assert np.array_equal(lazy[f].view(np.uint32), eager[f].view(np.uint32)), (f, plan[f], int(np.sum(lazy[f] != eager[f])))
    turned = [f + 1 for f in range(len(plan) - 1) if waves[f] & 1]                               # (the wave loop ends when a queue runs empty)
    assert turned and [f for f in range(1, len(plan)) if ran[f] > ran[f - 1]] == turned, (ran, turned)      # the passes ran exactly after the frames that turned the swap chain
    assert fixed[1] > 0 and fixed[-1] > fixed[6], fixed                                          # the camera moved: entries outlived the pick
    piped, _, _, _ = run(1, 5)                                                                     # frames enqueued back to back
    marks = [f for f in range(len(plan)) if f % 5 == 4 or f == len(plan) - 1]
    assert len(piped) == len(marks) and all(np.array_equal(a.view(np.uint32), eager[i].view(np.uint32)) for a, i in zip(piped, marks))
    if scene == "sponza":                                                                       # (the Cornell box at this size hardly ever accepts two reuse candidates: nothing to complete)
        broken, _, _, _ = run(2, 1)
        assert any(not np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(broken, eager))


def test_pending_history_passes_survive_exports_toggles_and_skipped_frames():
    """Lazy reuse, the bookkeeping around it: between two frames the history passes of the first may be pending.  Everything that can happen in that gap must
    find them done or keep them owed: a history export (lumen_mi_export_history: the exported reservoirs are complete, every word, whenever the swap chain has
    turned), the tuning key switched off and on again, a frame skipped because the scene has no light (WaveFrontRenderer.cpp:456-464), a resolution change
    (ResizeBuffers drops the history, WaveFrontRenderer.cpp:1424-1540).  Two renderers get the same calls, one with lazy_reuse 1, one with 0."""
    import torch
    from lumenrenderer_amd.scenes import sponza_standin
    d = sponza_standin()
    W, H = 208, 120
    lazy = product_from(d, W, H, 3, blend=False, tuning={"lazy_reuse": 1})
    eager = product_from(d, W, H, 3, blend=False, tuning={"lazy_reuse": 0})
    both = (lazy, eager)
    c0 = d.camera

    def frames(n, tag, k0=0):
        for k in range(n):
            for r in both:
                r.SetCamera(*_orbit(c0, k0 + k))
                assert r.TraceFrameAsync()
        imgs = []
        for r in both:
            r.Synchronize(); imgs.append(r.GetRadiance().copy())
        assert np.array_equal(imgs[0].view(np.uint32), imgs[1].view(np.uint32)), (tag, int(np.sum(imgs[0] != imgs[1])))

    def history():
        out = []
        for r in both:
            w, h = r.GetRadiance().shape[1], r.GetRadiance().shape[0]
            buf = torch.zeros(w * h * 20, dtype=torch.float32, device="cuda")
            r.ExportHistory((0, 0, w, h), buf.data_ptr()); r.Synchronize(); torch.cuda.synchronize()
            out.append(buf.cpu().numpy().view(np.uint32).reshape(-1, 20))
        return out

    frames(3, "odd depth")
    a, b = history()                                        # depth 3: the chain has turned, the export runs the pending passes first
    assert np.array_equal(a, b), int(np.sum(np.any(a != b, axis=1)))
    assert a[:, 5].max() > 32                               # (sample counts beyond one frame's 32 candidates: merged history)
    ran = lazy.GetCounters(64)[54]
    frames(2, "after the export", 3)
    assert lazy.GetCounters(64)[54] == ran + 1              # the passes of the frame before the export ran once (in the export), those of the next frame in the frame after it
    for r in both: r.SetDepth(4)
    frames(3, "even depth", 5)
    a, b = history()                                        # the chain has not turned: what is exported is the other buffer, untouched by the pending passes
    assert np.array_equal(a, b)
    lazy.SetTuning("lazy_reuse", 0); frames(2, "switched off", 8)
    lazy.SetTuning("lazy_reuse", 1); frames(2, "switched on", 10)
    for r in both: r.m_Scene.m_MeshInstances[1].SetEmissiveness(1, (0.0, 0.0, 0.0), 1.0)          # the only light DISABLED: the frame is skipped
    for r in both: assert r.TraceFrame() is False
    inst = d.instances[1]
    for r in both: r.m_Scene.m_MeshInstances[1].SetEmissiveness(inst["emission_mode"], inst["override_radiance"], inst["scale"])
    frames(3, "after the skipped frame", 12)
    for r in both: r.SetDepth(5)
    frames(2, "odd again", 15)
    for r in both: r.SetRenderResolution(160, 96)
    frames(3, "resized", 17)
    for r in both: r.close()


def test_counter_totals_sum_every_traceframe_on_the_device():
    """lumen_mi_get_counter_totals: the frame's last kernel adds its counter block to a 64-bit block on the device, so that a throughput
    measurement counts the rays of ALL the frames it timed without reading counters back in between (bench.py).  The sums must equal the
    per-frame counters added up on the host, for frames enqueued back to back, and a reset must start over."""
    d = cornell()
    r = product_from(d, 96, 80, 5, blend=True)
    want = [0] * 12
    for k in range(6):
        assert r.TraceFrame()
        c = r.GetCounters(12)
        for i in (0, 1, 2, 4, 5, 6, 7, 8): want[i] += c[i]
    t = r.GetCounterTotals(50)
    assert t[3] == 6 and [t[i] for i in (0, 1, 2, 4, 5, 6, 7, 8)] == [want[i] for i in (0, 1, 2, 4, 5, 6, 7, 8)], (t[:12], want)
    assert t[2] == t[48] + t[49]
    r.GetCounterTotals(4, reset=True)
    for _ in range(3):
        assert r.TraceFrameAsync()                      # no synchronisation between frames
    t2 = r.GetCounterTotals(50)
    assert t2[3] == 3 and 0 < t2[0] < t[0] and t2[4] == 3 * 96 * 80
    r.close()


def test_render_thread_with_concurrent_scene_edits():
    """StartRendering() (WaveFrontRenderer.cpp:1109-1117): the render thread traces frames while the main thread moves an
    instance, the camera and the light, reads pixels back and creates resources.  No oracle here (the number of frames the
    thread gets to is not deterministic): the run must stay alive and every read-back must be a complete, finite frame."""
    import time
    soup = random_soup(2000, 33, extent=6.0, size=0.7)
    d = cornell()
    p = soup.primitives[0]
    v = np.array(p["vertices"], np.float32).reshape(-1, 12).copy(); v[:, 0:3] *= np.float32(0.07)
    mat = d.add_material(diffuse_color=(0.6, 0.6, 0.2, 1.0), roughness_factor=0.5, metallic_factor=0.0)
    inst = d.add_instance(d.add_mesh([d.add_primitive(v, p["indices"], mat)]), _rigid(0.0, (0.0, 1.0, 0.0)))
    r = product_from(d, 160, 120, 4, blend=False)
    mi = r.m_Scene.m_MeshInstances[inst]
    from lumenrenderer_amd.capi import LumenMIError
    r.StartRendering()
    t0 = time.time()
    while True:                                   # until the thread has produced its first frame a read-back reports "no frame traced yet"
        try:
            r.GetRadiance(); break
        except LumenMIError as e:
            assert e.code == 3 and time.time() - t0 < 20.0, str(e)
            time.sleep(0.005)
    t0 = time.time(); k = 0
    while time.time() - t0 < 1.0:
        k += 1
        mi.SetTransform(_rigid(0.1 * k, (0.3 * np.sin(0.2 * k), 1.0, 0.2 * np.cos(0.2 * k))))
        if k % 3 == 0:
            mi.SetEmissiveness(2 if (k // 3) % 2 else 0, (2.0, 2.0, 2.0), 1.0)
        if k % 5 == 0:
            r.SetCamera((0.02 * (k % 7), 1.0, 3.4), (-1, 0, 0), (0, 1, 0), (0, 0, -1))
        if k % 4 == 0:
            r.CreateTexture(np.full((2, 2, 4), 200, np.uint8), normalize=False)
        rad = r.GetRadiance()
        assert np.isfinite(rad).all() and rad.shape == (120, 160, 4)
        px = r.GetOutputTexturePixels()
        assert px.shape[:2] == (120, 160)
    r.StopRendering()
    assert k > 10
    c = r.GetCounters()
    assert c[4] == 160 * 120 and c[0] >= c[4]
    # the renderer is still consistent: a synchronous frame after the thread has stopped renders the final scene state
    assert r.TraceFrame() is True
    assert np.isfinite(r.GetRadiance()).all() and r.GetRadiance()[..., :3].max() > 0
    r.close()


def test_full_size_overlapped_schedule_equals_the_serial_one():
    """At the benchmark size the kernels are long enough for real overlap: four streams, two frames in flight, candidates picked
    ahead.  Eight blended frames enqueued back to back must equal, bit for bit, the same frames rendered on one stream with a
    synchronisation after every frame (no oracle at this size: it would take minutes)."""
    from lumenrenderer_amd.scenes import sponza_standin
    W, H, D = 2560, 1440, 6
    out = []
    for tuning, sync_each in (({}, False), ({"single_stream": 1, "tail_below": 0, "pick_ahead": 0}, True), ({"tail_below": 1 << 30}, False)):
        r = product_from(sponza_standin(), W, H, D, blend=True, tuning=tuning)
        for _ in range(8):
            assert r.TraceFrameAsync()
            if sync_each:
                r.Synchronize()
        r.Synchronize()
        out.append((r.GetRadiance().copy(), r.GetChannel(0).copy(), r.GetChannel(1).copy(), list(r.GetCounters()[:20])))
        r.close()
    for k in (1, 2):
        for a, b in zip(out[0][:3], out[k][:3]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), k
        assert out[0][3] == out[k][3], k


def test_c2_at_full_size_is_bit_exact_against_the_oracle():
    """BASELINE config C2 at its full size against the ORACLE (not only against another schedule): 2560x1440, depth 6, ReSTIR on,
    the four blended TraceFrames of one benchmark frame (4 spp) enqueued back to back with the default overlapped schedule.
    Radiance, both light channels, the sRGB8 output and the ray counters equal the CPU restatement bit for bit on all 3.7 M pixels
    (the oracle needs about two seconds per TraceFrame on the host cores)."""
    from lumenrenderer_amd.scenes import sponza_standin
    W, H, D = 2560, 1440, 6
    d = sponza_standin()
    r = product_from(d, W, H, D, blend=True)
    o = oracle_from(d, W, H, D, blend=True)
    for _ in range(4):
        assert r.TraceFrameAsync()
        assert o.trace_frame() == 0
    r.Synchronize()
    got, want = r.GetRadiance(), o.radiance()
    assert rel_l2(got, want) <= RADIANCE_TOL
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), int(np.sum(got.view(np.uint32) != want.view(np.uint32)))
    for ch in (0, 1):
        assert np.array_equal(r.GetChannel(ch).view(np.uint32), o.channel(ch).view(np.uint32)), ch
    assert np.array_equal(r.GetOutputTexturePixels(), o.output_pixels())
    c, s = r.GetCounters(), o.stats(24)
    assert list(c[:4 + D]) == list(s[:4 + D]), (c[:12], s[:12])
    assert c[0] > W * H and c[2] > 1_000_000
    r.close(); o.close()


def test_c2_textured_at_full_size_is_bit_exact_against_the_oracle():
    """Bench workload c2t: the C2 scene with seeded 1024 x 1024 base-colour (sRGB-decoded), normal and metal-roughness maps on its 22
    opaque / metal materials, so that every hit performs the bilinear fetches of GPUExtractSurfaceData.cu:59-60,169-181 and the shading
    normal / roughness / albedo vary per pixel.  2560x1440, depth 6, two blended TraceFrames, default schedule, exact mode: radiance,
    G-buffer and counters equal the oracle bit for bit."""
    from lumenrenderer_amd.scenes import sponza_standin
    W, H, D = 2560, 1440, 6
    d = sponza_standin(textured=True)
    r = product_from(d, W, H, D, blend=True)
    o = oracle_from(d, W, H, D, blend=True)
    for _ in range(2):
        assert r.TraceFrameAsync()
        assert o.trace_frame() == 0
    r.Synchronize()
    got, want = r.GetRadiance(), o.radiance()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), int(np.sum(got.view(np.uint32) != want.view(np.uint32)))
    g = r.GetGBuffer()
    assert np.array_equal(g.view(np.uint32), o.gbuffer().view(np.uint32))
    c, s = r.GetCounters(), o.stats(24)
    assert list(c[:4 + D]) == list(s[:4 + D]), (c[:12], s[:12])
    # the maps are really sampled: tens of thousands of distinct albedo values and shading normals that leave the geometric ones
    hit = g[..., 0, 3] > 0
    assert len(np.unique(g[..., 4, :3][hit].view(np.uint32).reshape(-1, 3), axis=0)) > 50000
    assert len(np.unique(g[..., 7, 0][hit].view(np.uint32) >> 24)) > 50                    # roughness bytes from the G channel
    r.close(); o.close()


def test_c3_at_full_size_is_bit_exact_against_the_oracle():
    """BASELINE config C3 at its full size against the oracle: the atrium with 1 026 emissive triangles (513 quads; the reference's
    slice quirk keeps one light-list entry per quad, GPUDataBufferKernels.cu:37), 2560x1440, depth 6, two blended TraceFrames,
    default schedule."""
    from lumenrenderer_amd.scenes import sponza_standin
    W, H, D = 2560, 1440, 6
    d = sponza_standin(extra_lights=512)
    r = product_from(d, W, H, D, blend=True)
    o = oracle_from(d, W, H, D, blend=True)
    for _ in range(2):
        assert r.TraceFrameAsync()
        assert o.trace_frame() == 0
    r.Synchronize()
    got, want = r.GetRadiance(), o.radiance()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), int(np.sum(got.view(np.uint32) != want.view(np.uint32)))
    c, s = r.GetCounters(), o.stats(24)
    assert list(c[:4 + D]) == list(s[:4 + D]), (c[:12], s[:12])
    assert len(r.GetLights()[0]) == 513
    r.close(); o.close()
    # The candidate pick of a light list above 512 records (round 6): 1024-thread blocks, four tiles around one LDS light table ("pick_wide"; the default uses it in the
    # fast mode only).  The exact instantiation of that kernel against the oracle's frames, and the fast one against the fast global-gather kernel: the same bits.
    w = product_from(d, W, H, D, blend=True, tuning={"pick_wide": 2})
    for _ in range(2):
        assert w.TraceFrameAsync()
    w.Synchronize()
    assert np.array_equal(w.GetRadiance().view(np.uint32), want.view(np.uint32))
    assert list(w.GetCounters()[:4 + D]) == list(s[:4 + D])
    w.close()
    fast = []
    for wide in (0, 1):
        f = product_from(d, W, H, D, blend=True, tuning={"fast_resample": 1, "pick_wide": wide})
        for _ in range(2):
            assert f.TraceFrameAsync()
        f.Synchronize()
        fast.append((f.GetRadiance().copy(), list(f.GetCounters()[:4 + D])))
        f.close()
    assert np.array_equal(fast[0][0].view(np.uint32), fast[1][0].view(np.uint32)) and fast[0][1] == fast[1][1]


def test_fused_second_spatial_pass_and_combine_is_the_same_image_in_both_modes():
    """On eager frames the second spatial reuse pass ends with the pixel's combine ("fuse_combine", default) instead of a launch of its own: the exact mode against the
    oracle (odd depth = the history is read every frame; a window that cuts tiles), and the fast mode — which has no bit-level oracle — fused against unfused, bit for bit."""
    from lumenrenderer_amd.scenes import sponza_standin
    W, H, D = 331, 203, 5
    d = sponza_standin()
    o = oracle_from(d, W, H, D, blend=True)
    for _ in range(4):
        assert o.trace_frame() == 0
    want = o.radiance()
    imgs = {}
    for mode, fuse in ((0, 1), (0, 0), (1, 1), (1, 0)):
        r = product_from(d, W, H, D, blend=True, tuning={"fast_resample": mode, "fuse_combine": fuse, "lazy_reuse": 0})
        for _ in range(4):
            assert r.TraceFrameAsync()
        r.Synchronize()
        imgs[(mode, fuse)] = (r.GetRadiance().copy(), list(r.GetCounters()[:4 + D]))
        r.close()
    for fuse in (1, 0):
        assert np.array_equal(imgs[(0, fuse)][0].view(np.uint32), want.view(np.uint32)), fuse
        assert imgs[(0, fuse)][1] == list(o.stats(24)[:4 + D])
    assert np.array_equal(imgs[(1, 1)][0].view(np.uint32), imgs[(1, 0)][0].view(np.uint32)) and imgs[(1, 1)][1] == imgs[(1, 0)][1]
    o.close()


def test_wide_candidate_pick_with_a_ragged_tile_count():
    """The four-tiles-per-block candidate pick (light lists of 513 .. 1 984 records) on a window whose tile count is not a multiple of four (13 x 7 = 91 tiles of 16 x 16; the
    last block has one live quarter) and whose right / bottom tiles are cut: the exact instantiation against the oracle, the fast one against the fast global-gather kernel."""
    from lumenrenderer_amd.scenes import sponza_standin
    W, H, D = 203, 101, 3
    d = sponza_standin(extra_lights=512)
    o = oracle_from(d, W, H, D, blend=True)
    r = product_from(d, W, H, D, blend=True, tuning={"pick_wide": 2})
    for _ in range(3):
        assert r.TraceFrameAsync()
        assert o.trace_frame() == 0
    r.Synchronize()
    assert np.array_equal(r.GetRadiance().view(np.uint32), o.radiance().view(np.uint32))
    assert list(r.GetCounters()[:4 + D]) == list(o.stats(24)[:4 + D])
    r.close(); o.close()
    fast = []
    for wide in (0, 1):
        f = product_from(d, W, H, D, blend=True, tuning={"fast_resample": 1, "pick_wide": wide})
        for _ in range(3):
            assert f.TraceFrameAsync()
        f.Synchronize()
        fast.append(f.GetRadiance().copy())
        f.close()
    assert np.array_equal(fast[0].view(np.uint32), fast[1].view(np.uint32))
    assert np.isfinite(fast[0]).all() and float(fast[0][..., :3].mean()) > 0.01
    # the largest table the wide block takes (1 984 records = 127 KB of dynamic LDS beside the four bags: one block per CU) and the first list that gathers again
    for extra in (1983, 1984):
        big = sponza_standin(extra_lights=extra)
        imgs = []
        for wide in (0, 1):
            f = product_from(big, W, H, D, blend=True, tuning={"fast_resample": 1, "pick_wide": wide})
            assert len(f.GetLights()[0]) == extra + 1
            for _ in range(2):
                assert f.TraceFrameAsync()
            f.Synchronize()
            imgs.append(f.GetRadiance().copy())
            f.close()
        assert np.array_equal(imgs[0].view(np.uint32), imgs[1].view(np.uint32)), extra
        assert np.isfinite(imgs[0]).all() and float(imgs[0][..., :3].mean()) > 0.01


def test_c4_at_full_size_is_bit_exact_against_the_oracle():
    """BASELINE config C4 at its full size against the oracle: 3840x2160, depth 8, two blended TraceFrames, default schedule."""
    from lumenrenderer_amd.scenes import sponza_standin
    W, H, D = 3840, 2160, 8
    d = sponza_standin()
    r = product_from(d, W, H, D, blend=True)
    o = oracle_from(d, W, H, D, blend=True)
    for _ in range(2):
        assert r.TraceFrameAsync()
        assert o.trace_frame() == 0
    r.Synchronize()
    got, want = r.GetRadiance(), o.radiance()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), int(np.sum(got.view(np.uint32) != want.view(np.uint32)))
    assert np.array_equal(r.GetOutputTexturePixels(), o.output_pixels())
    c, s = r.GetCounters(), o.stats(24)
    assert list(c[:4 + D]) == list(s[:4 + D]), (c[:12], s[:12])
    r.close(); o.close()


def sandbox_camera(desc, k):
    """Camera pose of frame k of the `sandbox` workload (bench.py --workload sandbox uses the same path): the scene's own camera walking forward and to the
    side while it yaws and nods a little — every TraceFrame sees a different view, so motion vectors are non-zero everywhere and the temporal pass
    reprojects (Sandbox: any camera move switches blending off for that frame, OutputLayer.cpp:492-495)."""
    from lumenrenderer_amd.scenes import sandbox_camera_pose
    return sandbox_camera_pose(desc, k)


def test_sandbox_default_workload_is_bit_exact_and_the_fast_mode_does_not_drift():
    """The reference's OWN default workload (Sandbox/src/Application.cpp:89-93: 1280x720, depth 5, ReSTIR on; OutputLayer.cpp:492-495: blending off while the
    camera moves): an ODD depth, so the reservoir swap chain turns every frame (WaveFrontRenderer.cpp:827) and the temporal pass (ReSTIRKernels.cu:1015-1121)
    reads a LIVE history through non-zero motion vectors, every frame; the history passes run with their frame (no lazy reuse at odd depths).
    16 TraceFrames with a moving camera on the stand-in scene, against the oracle at full size:
      exact mode  radiance, DIRECT / INDIRECT, motion vectors and every counter bit-identical at frames 1, 2, 8 and 16;
      fast mode   (what bench.py's headline runs) <= 1e-3 relative L2 AT FRAME 16 — reservoir decisions that flip within an ulp do not compound through
                  16 generations of history — and the error does not grow between frame 8 and frame 16 by more than noise."""
    from lumenrenderer_amd.scenes import sponza_standin
    W, H, D, FRAMES = 1280, 720, 5, 16
    d = sponza_standin()
    r = product_from(d, W, H, D, blend=False)
    rf = product_from(d, W, H, D, blend=False, tuning={"fast_resample": 1})
    o = oracle_from(d, W, H, D, blend=False)
    errs = {}
    for k in range(FRAMES):
        pose = sandbox_camera(d, k)
        r.SetCamera(*pose); rf.SetCamera(*pose); o.set_camera(*pose)
        assert r.TraceFrameAsync() and rf.TraceFrameAsync()
        assert o.trace_frame() == 0
        if k + 1 in (1, 2, 8, 16):
            r.Synchronize(); rf.Synchronize()
            got, want = r.GetRadiance(), o.radiance()
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (k, int(np.sum(got.view(np.uint32) != want.view(np.uint32))))
            for ch in (0, 1):
                assert np.array_equal(r.GetChannel(ch).view(np.uint32), o.channel(ch).view(np.uint32)), (k, ch)
            c, s = r.GetCounters(), o.stats(24)
            assert list(c[:4 + D]) == list(s[:4 + D]), (k, c[:12], s[:12])
            _, _, mv = r.GetDenoiserInputs(); _, _, omv = o.denoiser_inputs()
            assert np.array_equal(mv.reshape(-1, 2), omv), k
            if k:
                assert (mv.reshape(-1, 2) != 0).any(axis=1).mean() > 0.5                    # the camera really moved: most pixels reproject
            fast = rf.GetRadiance()
            assert np.isfinite(fast).all()
            errs[k + 1] = rel_l2(fast[..., :3], want[..., :3])
            cf = rf.GetCounters()
            assert list(cf[4:4 + D]) == list(s[4:4 + D])
    print("sandbox workload, fast mode rel-L2 vs oracle by frame:", {k: f"{v:.3e}" for k, v in errs.items()})
    assert errs[16] <= RADIANCE_TOL, errs
    assert errs[16] <= 3.0 * max(errs[8], 1e-6) + 1e-5, errs                                # no compounding through the history
    # the history is live: the same pose rendered by a renderer without any history gives a different DIRECT channel
    fresh = product_from(d, W, H, D, blend=False)
    fresh.SetCamera(*sandbox_camera(d, FRAMES - 1))
    for _ in range(1):
        assert fresh.TraceFrame()
    a, b = r.GetChannel(0), fresh.GetChannel(0)
    assert (np.any(a.view(np.uint32) != b.view(np.uint32), axis=-1)).mean() > 0.2
    r.close(); rf.close(); o.close(); fresh.close()


def test_1440p_odd_depth_static_camera_live_history_both_modes():
    """BASELINE's resolution at an odd depth (2560x1440, depth 5, static camera, blending on): the temporal history is live (unlike C2's depth 6, where the
    reference's swap quirk leaves it reset) and lazy reuse is off.  Four blended TraceFrames: exact mode bit-identical to the oracle, fast mode <= 1e-3."""
    from lumenrenderer_amd.scenes import sponza_standin
    W, H, D = 2560, 1440, 5
    d = sponza_standin()
    r = product_from(d, W, H, D, blend=True)
    rf = product_from(d, W, H, D, blend=True, tuning={"fast_resample": 1})
    o = oracle_from(d, W, H, D, blend=True)
    for _ in range(4):
        assert r.TraceFrameAsync() and rf.TraceFrameAsync()
        assert o.trace_frame() == 0
    r.Synchronize(); rf.Synchronize()
    got, want = r.GetRadiance(), o.radiance()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), int(np.sum(got.view(np.uint32) != want.view(np.uint32)))
    c, s = r.GetCounters(56), o.stats(24)
    assert list(c[:4 + D]) == list(s[:4 + D])
    assert c[54] == 0                                                                       # no deferred history pass at an odd depth: they ran with their frames
    err = rel_l2(rf.GetRadiance()[..., :3], want[..., :3])
    print(f"1440p depth 5, live history, fast mode rel-L2 {err:.3e}")
    assert err <= RADIANCE_TOL
    r.close(); rf.close(); o.close()


FAST_CASES = {                                                    # BASELINE configs at their full sizes: scene, size, depth, blended TraceFrames
    "c1": ("cornell", {}, 256, 256, 2, 1),
    "c2": ("sponza", {}, 2560, 1440, 6, 4),
    "c3": ("sponza", {"extra_lights": 512}, 2560, 1440, 6, 2),
    "c4": ("sponza", {}, 3840, 2160, 8, 2),
    "c5": ("foliage", {}, 1920, 1080, 6, 1),
    # dielectric + clear-coat + textured materials and an odd depth (temporal history is live): the surfaces the contracted evaluation
    # does not cover go through the second, exact launch of each ReSTIR pass (LM_RARE), the others through the first
    "mixed": ("textured", {}, 288, 224, 5, 4),
    # C2 with 1024^2 procedural base-colour / normal / metal-roughness maps on 22 of its 25 materials (bench workload c2t)
    "c2t": ("sponza", {"textured": True}, 2560, 1440, 6, 2),
}


@pytest.mark.parametrize("case", sorted(FAST_CASES))
def test_fast_resampling_mode_stays_within_the_north_star_tolerance(case):
    """Tuning key "fast_resample": the ReSTIR target function and resampling weights are evaluated with hardware reciprocal /
    reciprocal-square-root / square-root instructions and in contracted form (lm_bsdf.h LmFast / LmQuick, lm_restir.h) instead of the
    correctly rounded sequences of the exact mode.  Results are no longer bit-identical — a reservoir decision `rnd <= w / sum` can
    flip for a handful of pixels — but BASELINE.json asks for 1e-3 relative L2 on radiance, and that is the bar here, against the
    ORACLE, for every BASELINE configuration at its full size; ray counters agree to 0.1 % (only visibility rays can differ)."""
    from lumenrenderer_amd import scenes
    kind, kw, W, H, D, frames = FAST_CASES[case]
    d = cornell() if kind == "cornell" else scenes.sponza_standin(**kw) if kind == "sponza" else _textured_scene() if kind == "textured" else scenes.foliage_stress()
    r = product_from(d, W, H, D, blend=True, tuning={"fast_resample": 1})
    o = oracle_from(d, W, H, D, blend=True)
    for _ in range(frames):
        assert r.TraceFrameAsync()
        assert o.trace_frame() == 0
    r.Synchronize()
    got, want = r.GetRadiance(), o.radiance()
    err = rel_l2(got[..., :3], want[..., :3])
    differing = float(np.mean(np.any(got.view(np.uint32) != want.view(np.uint32), axis=-1)))
    print(f"fast_resample {case}: rel-L2 {err:.3e}, pixels not bit-identical {differing:.4f}")
    assert np.isfinite(got).all()
    assert err <= RADIANCE_TOL, (case, err)
    assert differing > 0.0 or case == "c1"                           # the mode is really on
    c, s = r.GetCounters(), o.stats(24)
    for k in range(3):                                                # closest-hit rays, NEE shadow rays, ReSTIR visibility rays
        assert abs(int(c[k]) - int(s[k])) <= 1e-3 * max(1, int(s[k])), (case, k, c[:4], s[:4])
    assert list(c[4:4 + D]) == list(s[4:4 + D])                       # the path waves do not depend on the resampling arithmetic
    c2 = r.GetCounters(56)
    assert c2[52] == 0 or c2[53] == 1                                 # the second (exact) launch is only skipped when no material can need it
    assert (c2[53] == 1) == (case == "mixed") and (c2[52] > 0) == (case == "mixed")
    if case == "mixed":
        g = r.GetGBuffer()
        p = g[..., 7, :3].copy().view(np.uint32)
        rare = ((p[..., 2] & 0x00ff00ff) != 0) | ((p[..., 0] >> 24) == 0)
        hit = (g[..., 1, 3].copy().view(np.uint32) == 0) & (g[..., 0, 3] > 0)
        assert 0.0005 < (rare & hit).mean() < 0.9 and (~rare & hit).mean() > 0.05      # both launches had work
    r.close(); o.close()


@pytest.mark.parametrize("case", ["c1", "c2", "c3", "c5", "mixed", "c2t"])
def test_fast_shade_changes_only_the_last_bits(case):
    """Tuning key "fast_shade" (on top of fast_resample): the NEE contribution at depth >= 1 is evaluated with hardware reciprocal / square
    root.  It decides nothing but two thresholds of a shadow ray's emission; sampling and Russian roulette stay exact, so the rays of
    every wave are the oracle's, and the radiance stays within the north-star tolerance (measured figure printed)."""
    from lumenrenderer_amd import scenes
    kind, kw, W, H, D, frames = FAST_CASES[case]
    d = cornell() if kind == "cornell" else scenes.sponza_standin(**kw) if kind == "sponza" else _textured_scene() if kind == "textured" else scenes.foliage_stress()
    r = product_from(d, W, H, D, blend=True, tuning={"fast_resample": 1, "fast_shade": 1})
    o = oracle_from(d, W, H, D, blend=True)
    for _ in range(frames):
        assert r.TraceFrameAsync()
        assert o.trace_frame() == 0
    r.Synchronize()
    got, want = r.GetRadiance(), o.radiance()
    err = rel_l2(got[..., :3], want[..., :3])
    print(f"fast_shade {case}: rel-L2 {err:.3e}")
    assert np.isfinite(got).all() and err <= RADIANCE_TOL, (case, err)
    c, s = r.GetCounters(), o.stats(24)
    assert list(c[4:4 + D]) == list(s[4:4 + D])                       # which rays exist does not depend on the NEE arithmetic
    for k in range(3):
        assert abs(int(c[k]) - int(s[k])) <= 1e-3 * max(1, int(s[k])), (case, k, c[:4], s[:4])
    r.close(); o.close()


def test_fast_mode_with_an_srgb_flagged_metal_roughness_map_takes_the_exact_launch():
    """whether fast mode enqueues the second (exact, LM_RARE) launch of each ReSTIR pass is predicted on the host from the
    smallest green texel of a material's metal-roughness map.  A map created with normalize = 1 is sRGB-decoded by the device fetch
    (byte 10 -> 0.003 -> roughness byte 0, mirror-like: outside the contracted evaluation), so the prediction must use the same decode:
    with the raw byte (10 / 255 -> byte 7) it said "not rare", both launches skipped those pixels and their reservoirs went stale."""
    rng = np.random.default_rng(5)
    d = cornell()
    mr = rng.integers(8, 14, (8, 8, 4), dtype=np.uint8)                       # green 8..13: raw x 0.8 -> byte 6..10; sRGB-decoded -> byte 0
    mat = d.add_material(diffuse_color=(0.8, 0.7, 0.6, 1.0), roughness_factor=0.8, metallic_factor=0.5, specular_factor=0.5,
                         metallic_roughness_texture=d.add_texture(mr, True))
    soup = random_soup(200, 31, extent=6.0, size=1.0)
    pr = soup.primitives[0]
    v = np.array(pr["vertices"], np.float32).reshape(-1, 12).copy(); v[:, 0:3] *= np.float32(0.12)
    v[:, 3:5] = rng.uniform(0, 1, (len(v), 2)).astype(np.float32)
    d.add_instance(d.add_mesh([d.add_primitive(v, pr["indices"], mat)]), _rigid(0.3, (0.0, 1.0, 0.0)))
    W, H, D = 160, 128, 5
    r = product_from(d, W, H, D, blend=True, tuning={"fast_resample": 1})
    o = oracle_from(d, W, H, D, blend=True)
    for _ in range(3):
        assert r.TraceFrameAsync() and o.trace_frame() == 0
    r.Synchronize()
    c = r.GetCounters(56)
    assert c[53] == 1 and c[52] > 0, c[50:56]                                # predicted rare, and such surfaces were really extracted
    g = r.GetGBuffer()
    p0 = g[..., 7, 0].copy().view(np.uint32)
    hit = (g[..., 1, 3].copy().view(np.uint32) == 0) & (g[..., 0, 3] > 0)
    assert ((p0 >> 24) == 0)[hit].mean() > 0.01                              # roughness byte 0 on the textured soup
    err = rel_l2(r.GetRadiance()[..., :3], o.radiance()[..., :3])
    print(f"fast mode, sRGB-flagged metal-roughness map: rel-L2 {err:.3e}, rare surfaces {c[52]}")
    assert err <= RADIANCE_TOL, err
    r.close(); o.close()


@pytest.mark.parametrize("shape", [(2560, 1440, 6), (333, 217, 5)])
def test_spatial_pass_with_probes_in_lds_gives_the_same_image(shape):
    """Tuning key spatial_lds (fast mode): the first spatial pass stages the probes of a 32 x 32 tile + 30-pixel border in LDS (132 KB) and
    tests its five neighbours there instead of gathering them; the probes are the same bits, so the image must be the other kernel's to
    the bit — at the benchmark size and on a window whose edges cut the tiles and the border."""
    from lumenrenderer_amd.scenes import sponza_standin
    W, H, D = shape
    imgs = []
    for lds in (0, 1, 2):                                           # 2: the 16 x 16 tile + border (76 x 76 probes, 92 KB), round 4
        r = product_from(sponza_standin(), W, H, D, blend=True, tuning={"fast_resample": 1, "spatial_lds": lds})
        for _ in range(3):
            assert r.TraceFrameAsync()
        r.Synchronize()
        imgs.append((r.GetRadiance().copy(), list(r.GetCounters(8))))
        r.close()
    for other in imgs[1:]:
        assert np.array_equal(imgs[0][0].view(np.uint32), other[0].view(np.uint32)) and imgs[0][1] == other[1]


def test_c4_4k_depth8_overlapped_schedule_equals_the_serial_one():
    """BASELINE config C4 at its full size (3840x2160, depth 8, blended frames): the overlapped schedule equals the serial one bit
    for bit, and the properties the radiance must have hold."""
    from lumenrenderer_amd.scenes import sponza_standin
    W, H, D = 3840, 2160, 8
    out = []
    for tuning, sync_each in (({}, False), ({"single_stream": 1, "tail_below": 0, "pick_ahead": 0}, True)):
        r = product_from(sponza_standin(), W, H, D, blend=True, tuning=tuning)
        for _ in range(4):
            assert r.TraceFrameAsync()
            if sync_each:
                r.Synchronize()
        r.Synchronize()
        out.append((r.GetRadiance().copy(), r.GetChannel(0).copy(), r.GetChannel(1).copy(), list(r.GetCounters()[:20])))
        r.close()
    for a, b in zip(out[0][:3], out[1][:3]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert out[0][3] == out[1][3]
    rad, c = out[0][0], out[0][3]
    assert rad.shape == (H, W, 4) and np.isfinite(rad).all() and (rad[..., :3] >= 0).all() and rad[..., :3].max() > 0.05
    assert c[4] == W * H and all(c[4 + d] >= c[5 + d] for d in range(D - 1)) and c[4 + D - 1] > 0        # rays per wave: all pixels, then thinning


def test_device_tree_build_gives_the_same_hits_and_images():
    """Tuning key gpu_build (round 4, VERDICT r3 missing #6): the scene tree built on the device (csrc/bvh_gpu.hip: Morton sort, radix tree, collapse, refit) instead of
    by the host's SAH builder.  The hit rule does not depend on the tree, so closest-hit records of random rays equal those of the SAH-built tree exactly, frames equal the
    oracle bit for bit, and moving an instance afterwards refits the device-built tree."""
    from lumenrenderer_amd.scenes import sponza_standin
    d = sponza_standin()
    a = product_from(d, 160, 96, 4, blend=True, tuning={"gpu_build": 1}); b = product_from(d, 160, 96, 4, blend=True)
    rng = np.random.default_rng(3)
    o = rng.uniform(-8, 8, (20000, 3)).astype(np.float32); o[:, 1] = rng.uniform(0.2, 8, 20000)
    v = rng.normal(size=(20000, 3)).astype(np.float32); v /= np.linalg.norm(v, axis=1, keepdims=True)
    ipa, uva = a.QueryClosest(o, v); ipb, uvb = b.QueryClosest(o, v)
    assert np.array_equal(ipa, ipb) and np.array_equal(uva.view(np.uint32), uvb.view(np.uint32)) and (uva[:, 2] > 0).mean() > 0.5
    assert a.GetCounters(60)[56] == 1 and b.GetCounters(60)[56] == 0
    orc = oracle_from(d, 160, 96, 4, blend=True)
    _compare_frames(a, orc, 3, check_gbuffer=False)
    a.close(); b.close(); orc.close()
    # dynamic scene on top of a device-built tree: transform edits refit it
    dd = cornell()
    r = product_from(dd, 96, 64, 3, tuning={"gpu_build": 1}); oo = oracle_from(dd, 96, 64, 3)
    assert r.TraceFrame() is True and oo.trace_frame() == 0
    m = _rigid(0.3, (0.1, 0.05, -0.1))
    r.m_Scene.m_MeshInstances[1].SetTransform(m); oo.set_instance_transform(1, m)
    assert r.TraceFrame() is True and oo.trace_frame() == 0
    assert np.array_equal(r.GetRadiance().view(np.uint32), oo.radiance().view(np.uint32))
    assert r.GetCounters(60)[56] == 1 and r.GetCounters(60)[50] >= 1
    r.close(); oo.close()


def test_c5_ten_million_triangles_queries_and_schedules():
    """BASELINE config C5 (10 M-triangle foliage stand-in, 1080p, depth 6): the parallel host build, the 4-wide tree and its stack
    bound at that size.  Closest-hit / any-hit queries against the oracle's brute-force loop over all triangles (no BVH on the
    oracle side), and the overlapped schedule against the serial one."""
    from lumenrenderer_amd.scenes import foliage_stress
    d = foliage_stress()
    assert d.triangle_count() >= 10_000_000
    W, H, D = 1920, 1080, 6
    out = []
    for tuning, sync_each in (({}, False), ({"single_stream": 1, "tail_below": 0, "pick_ahead": 0}, True)):
        r = product_from(d, W, H, D, blend=True, tuning=tuning)
        for _ in range(3):
            assert r.TraceFrameAsync()
            if sync_each:
                r.Synchronize()
        r.Synchronize()
        out.append((r.GetRadiance().copy(), list(r.GetCounters()[:12])))
        if not sync_each:
            info = r.GetBvhInfo()
            assert info["triangles"] == d.triangle_count() and info["nodes"] > info["triangles"] // 8
            rng = np.random.default_rng(5)
            c = d.camera
            org = np.tile(np.float32(c["position"]), (256, 1)) + rng.uniform(-0.5, 0.5, (256, 3)).astype(np.float32)
            dr = rng.normal(size=(256, 3)).astype(np.float32); dr /= np.linalg.norm(dr, axis=1, keepdims=True)
            ip, uvt = r.QueryClosest(org, dr)
            occ = r.QueryAny(org, dr, np.full(256, 50.0, np.float32))
            o = oracle_from(d, W, H, D, blend=True)
            for _ in range(3):
                assert o.trace_frame() == 0                       # the C5 frame itself against the oracle (its own median-split BVH)
            assert np.array_equal(out[0][0].view(np.uint32), o.radiance().view(np.uint32))
            assert list(out[0][1][:4 + D]) == list(o.stats(24)[:4 + D])
            oip, ouvt = o.trace_closest(org, dr, use_bvh=False)
            oocc = o.trace_any(org, dr, np.full(256, 50.0, np.float32), use_bvh=False)
            assert np.array_equal(uvt.view(np.uint32), ouvt.view(np.uint32)) and np.array_equal(ip, oip) and np.array_equal(occ, oocc)
            assert (uvt[:, 2] > 0).sum() > 32
            o.close()
        r.close()
    assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32)) and out[0][1] == out[1][1]


@pytest.mark.parametrize("n_ranks", [2, 4, 8])
def test_stitched_tiles_equal_the_single_gpu_frame(n_ranks):
    """The multi-GPU decomposition on one GPU: every rank's window (tile + 60-pixel halo, lumenrenderer_amd/tiles.py) is
    rendered by its own renderer and the tiles are stitched.  With an even path depth the swap-chain quirk leaves no
    cross-frame ReSTIR history, so the stitched image must equal the full-frame render bit for bit on EVERY blended frame;
    with an odd depth the first frame must."""
    from lumenrenderer_amd import tiles
    from lumenrenderer_amd.scenes import sponza_standin
    W, H = 416, 232
    d = sponza_standin()
    for depth, frames in ((4, 3), (3, 1)):
        full = product_from(d, W, H, depth, blend=True)
        for _ in range(frames):
            assert full.TraceFrameAsync()
        full.Synchronize()
        want = full.GetRadiance().copy(); full.close()
        got = np.zeros_like(want)
        for rank in range(n_ranks):
            tile = tiles.tile_rect(rank, n_ranks, W, H); win = tiles.window_rect(tile, W, H)
            r = product_from(d, W, H, depth, blend=True, window=win)
            if rank % 2 == 0:
                r.SetTile(*tile)                         # halo pixels then skip indirect light, the second reuse pass and combine
            for _ in range(frames):
                assert r.TraceFrameAsync()
            r.Synchronize()
            rad = r.GetRadiance()
            got[tile[1]:tile[3], tile[0]:tile[2]] = rad[tile[1] - win[1]: tile[3] - win[1], tile[0] - win[0]: tile[2] - win[0]]
            r.close()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (depth, int(np.sum(got != want)))


@pytest.mark.parametrize("n_ranks,size,scene,depth,lazy", [(2, (416, 232), "sponza", 5, -1), (4, (416, 300), "sponza", 5, -1), (8, (1280, 720), "sponza", 5, -1),
                                                          (2, (200, 160), "cornell", 16, -1), (4, (200, 160), "cornell", 16, -1),
                                                          (4, (416, 300), "sponza", 5, 1), (4, (200, 160), "cornell", 16, 0)])
def test_seam_history_exchange_makes_odd_depth_tiles_exact(n_ranks, size, scene, depth, lazy):
    """Odd path depth: temporal reuse reads real history, and the history of a window's halo ring belongs to the neighbours.  All
    ranks of the decomposition live in this process (one renderer per window); after every frame each rank exports the part of
    its tile that lies in a neighbour's halo and imports its own ring (tiles.halo_plan, lumen_mi_export/import_history) -- the
    same packing the RCCL exchange uses, device to device.  The stitched image then equals the full-frame render bit for bit on
    every blended frame; without the exchange it does not (asserted, so that this test cannot pass vacuously).
    Cornell at depth 16: the rays run out after a few waves, at a different depth in every tile and frame; the ranks agree on the
    number of executed waves (the maximum: lumen_mi_export/import_wave_count) before the swap chain picks the history buffer."""
    import torch
    from lumenrenderer_amd import tiles
    from lumenrenderer_amd.scenes import sponza_standin
    W, H = size
    frames = 4 if scene == "sponza" else 6
    d = sponza_standin() if scene == "sponza" else cornell()
    full = product_from(d, W, H, depth, blend=True)
    want = []
    for _ in range(frames):
        assert full.TraceFrameAsync()
        full.Synchronize(); want.append(full.GetRadiance().copy())
    full.close()

    executed = []

    def stitched(exchange):
        ranks = []
        for rank in range(n_ranks):
            tile = tiles.tile_rect(rank, n_ranks, W, H); win = tiles.window_rect(tile, W, H)
            r = product_from(d, W, H, depth, blend=True, window=win, tuning={"lazy_reuse": lazy})     # (the history passes launched with their frame / when owed:
            r.SetTile(*tile)                                                                          #  the export below must find them done either way)
            ranks.append((r, tile, win, tiles.HistoryExchange(r, rank, n_ranks, W, H, "cuda:0")))
        images = []
        for _ in range(frames):
            for r, _, _, _ in ranks:
                assert r.TraceFrameAsync()
            if exchange:
                for r, _, _, hx in ranks:
                    r.ExportWaveCount(hx.waves.data_ptr())
                for r, _, _, _ in ranks:
                    r.Synchronize()
                most = max(int(hx.waves.item()) for _, _, _, hx in ranks)                  # dist.all_reduce(MAX) between processes
                executed.append(sorted(int(hx.waves.item()) for _, _, _, hx in ranks))
                for r, _, _, hx in ranks:
                    hx.waves.fill_(most); torch.cuda.synchronize()
                    r.ImportWaveCount(hx.waves.data_ptr())
                for _, _, _, hx in ranks:
                    hx.pack()
                for r, _, _, _ in ranks:
                    r.Synchronize()
                for rank, (_, _, _, hx) in enumerate(ranks):       # what batch_isend_irecv does between processes
                    for peer, _, recv in hx.plan:
                        if recv:
                            hx.recv[peer].copy_(ranks[peer][3].send[rank])
                torch.cuda.synchronize()
                for _, _, _, hx in ranks:
                    hx.unpack()
            img = np.zeros_like(want[0])
            for r, tile, win, _ in ranks:
                r.Synchronize()
                rad = r.GetRadiance()
                img[tile[1]:tile[3], tile[0]:tile[2]] = rad[tile[1] - win[1]: tile[3] - win[1], tile[0] - win[0]: tile[2] - win[0]]
            images.append(img)
        for r, _, _, _ in ranks:
            r.close()
        return images

    got = stitched(True)
    for f in range(frames):
        assert np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)), (f, int(np.sum(got[f] != want[f])))
    plain = stitched(False)
    assert np.array_equal(plain[0].view(np.uint32), want[0].view(np.uint32))                  # the first frame has no history
    assert any(not np.array_equal(plain[f].view(np.uint32), want[f].view(np.uint32)) for f in range(1, frames))
    if scene == "cornell":
        assert any(e[0] != e[-1] for e in executed), executed                                   # the ranks did disagree in some frame
    assert tiles.history_needed(5) and not tiles.history_needed(6)


def test_full_size_moving_scene_async_equals_serial():
    """Refit, scene-table refresh and light rebuild between asynchronously enqueued full-size frames.  The scene exists twice on
    the device: an edit is written into the set no frame in flight reads (scene.cpp syncScene), on the wave stream.  The edit
    pattern is irregular (moves, a frame without edits, an emissive-only edit, a material-only edit), so every combination of a
    stale / current set is crossed.  Compared with the serial schedule on a host-rebuilt BVH."""
    from lumenrenderer_amd.scenes import sponza_standin
    W, H, D = 2560, 1440, 6
    out = []
    for tuning, sync_each in (({}, False), ({"single_stream": 1, "tail_below": 0, "pick_ahead": 0, "refit": 0}, True)):
        d = sponza_standin()
        base = np.array(d.instances[0]["transform"], np.float32).reshape(4, 4)
        r = product_from(d, W, H, D, blend=True, tuning=tuning)
        for k in range(9):
            if k in (0, 1, 3, 6, 7):
                m = base.copy(); m[1, 3] += 0.002 * k; m[0, 3] -= 0.001 * k
                r.m_Scene.m_MeshInstances[0].SetTransform(m)
            if k == 3:
                r.m_Scene.m_MeshInstances[1].SetEmissiveness(2, (9.0, 8.0, 7.0), 40.0)
            if k == 4:
                r.m_Scene.m_MeshInstances[1].SetEmissiveness(2, (3.0, 8.0, 7.0), 25.0)        # lights only, geometry of the other set is stale
            if k == 5:
                r.m_Scene.m_MeshInstances[0].SetOverrideMaterial(r.m_Materials[1])            # scene table only
            assert r.TraceFrameAsync()
            if sync_each:
                r.Synchronize()
        r.Synchronize()
        out.append((r.GetRadiance().copy(), r.GetChannel(0).copy(), r.GetChannel(1).copy(), list(r.GetCounters()[:12])))
        r.close()
    for a, b in zip(out[0][:3], out[1][:3]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert out[0][3] == out[1][3]


@pytest.mark.parametrize("fuzz_seed", list(range(1, 13)))
def test_schedule_fuzzing_does_not_change_the_image(fuzz_seed):
    """Tuning key "fuzz": idle wavefronts of random length (0-200 us) in front of a random quarter of the launches of every frame
    shift what overlaps with what across the four streams and the two frames in flight.  A missing event dependency would show as a
    changed image; camera and scene move so that every cross-frame hazard (G-buffer rotation, reservoir swap chain, scene sets,
    motion vectors, tail queue) is live.  Compared with the serial schedule."""
    from lumenrenderer_amd.scenes import sponza_standin
    W, H, D = 1280, 720, 5                                           # odd depth: temporal reuse reads real history
    def run(tuning, sync_each):
        d = sponza_standin()
        base = np.array(d.instances[0]["transform"], np.float32).reshape(4, 4)
        r = product_from(d, W, H, D, blend=True, tuning=tuning)
        for k in range(10):
            if k % 3 != 2:
                m = base.copy(); m[1, 3] += 0.002 * k
                r.m_Scene.m_MeshInstances[0].SetTransform(m)
            if k == 4:
                r.m_Scene.m_MeshInstances[1].SetEmissiveness(2, (9.0, 8.0, 7.0), 30.0)
            if k in (5, 8):                       # topology edit: the scene is cleared and refilled (k = 8: with a second light quad)
                sc = r.m_Scene; sc.Clear()
                for n, inst in enumerate(d.instances + ([d.instances[1]] if k == 8 else [])):
                    mi = sc.AddMesh(r.m_Meshes[inst["mesh"]])
                    t = np.array(inst["transform"], np.float32).reshape(4, 4).copy(); t[0, 3] += 1.5 * (n >= len(d.instances))
                    mi.SetTransform(t); mi.SetEmissiveness(inst["emission_mode"], inst["override_radiance"], inst["scale"])
            c = d.camera
            r.SetCamera((c["position"][0] + 0.01 * k, c["position"][1], c["position"][2]), c["right"], c["up"], c["forward"], c["fov"])
            assert r.TraceFrameAsync()
            if sync_each:
                r.Synchronize()
        r.Synchronize()
        out = (r.GetRadiance().copy(), r.GetChannel(0).copy(), r.GetChannel(1).copy(), list(r.GetCounters()[:12]))
        r.close()
        return out
    global _FUZZ_REFERENCE
    if "_FUZZ_REFERENCE" not in globals():
        _FUZZ_REFERENCE = run({"single_stream": 1, "tail_below": 0, "pick_ahead": 0}, True)
    got = run({"fuzz": 0x9E3779B1 * fuzz_seed & 0x7fffffff, "tail_below": (0, 1 << 30, 60000)[fuzz_seed % 3]}, False)
    for a, b in zip(got[:3], _FUZZ_REFERENCE[:3]):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), int(np.sum(a != b))
    assert got[3] == _FUZZ_REFERENCE[3]


def test_full_size_stitched_tiles_async():
    """The 8-rank decomposition of the benchmark frame (1440p, depth 6, 4 blended frames enqueued back to back per rank, owned-
    tile restriction on): the rank windows are large enough for real overlap of streams and frames, and the path tail / pick-
    ahead schedules of small windows are active.  Stitched tiles must equal the single-GPU frame bit for bit."""
    from lumenrenderer_amd import tiles
    from lumenrenderer_amd.scenes import sponza_standin
    W, H, D, N = 2560, 1440, 6, 8
    d = sponza_standin()
    full = product_from(d, W, H, D, blend=True)
    for _ in range(4):
        assert full.TraceFrameAsync()
    full.Synchronize()
    want = full.GetRadiance().copy(); full.close()
    got = np.zeros_like(want)
    for rank in range(N):
        tile = tiles.tile_rect(rank, N, W, H); win = tiles.window_rect(tile, W, H)
        r = product_from(d, W, H, D, blend=True, window=win)
        r.SetTile(*tile)
        for _ in range(4):
            assert r.TraceFrameAsync()
        r.Synchronize()
        rad = r.GetRadiance()
        got[tile[1]:tile[3], tile[0]:tile[2]] = rad[tile[1] - win[1]: tile[3] - win[1], tile[0] - win[0]: tile[2] - win[0]]
        r.close()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), int(np.sum(np.any(got != want, axis=-1)))


# ---- size-independent properties at the full BASELINE size (no oracle run: it would take minutes) -----------------------
def test_full_size_properties_1440p():
    from lumenrenderer_amd.scenes import sponza_standin
    W, H, D = 2560, 1440, 6
    r = product_from(sponza_standin(), W, H, D, blend=True)
    assert r.TraceFrame()
    a = r.GetRadiance().copy(); c = r.GetCounters()
    n = W * H
    assert c[4] == n and all(c[4 + k] >= c[5 + k] for k in range(D - 1))          # one primary ray per pixel; waves only shrink
    assert c[0] == sum(c[4:4 + D]) and c[1] <= sum(c[5:4 + D]) and c[2] <= 2 * n   # ray accounting of SURVEY.md §8 d1
    assert np.isfinite(a).all() and (a[..., :3] >= 0).all() and a[..., :3].max() > 0
    r.close()
    # determinism: a second renderer reproduces the frame bit for bit (no dependence on wave / atomic order)
    r2 = product_from(sponza_standin(), W, H, D, blend=True)
    assert r2.TraceFrame()
    assert np.array_equal(r2.GetRadiance().view(np.uint32), a.view(np.uint32))
    r2.close()
    # linearity: doubling the light's radiance scale doubles the radiance exactly (power-of-two scaling is exact in fp32)
    r3 = product_from(sponza_standin(light_scale=100.0), W, H, D, blend=True)
    assert r3.TraceFrame()
    b = r3.GetRadiance()
    assert np.array_equal(b.view(np.uint32), (a * np.float32(2.0)).view(np.uint32))
    r3.close()


def test_exact_mode_is_the_same_image_whichever_compilation_a_kernel_comes_from():
    """The library holds every kernel twice — compiled with and without the SLP vectoriser (kernels.hip LM_NOSLP_VARIANT) — and the renderer picks per kernel class
    (LUMEN_MI_NOSLP_KERNELS / _EXACT; renderer.cpp applyNoSlpKernels).  Both compilations are -ffp-contract=off: in the exact mode every mix must render the same bits.
    Three processes (the lists are read when a renderer is created): everything with SLP, everything without, the default mix; radiance, channels and counters identical,
    and identical to the oracle."""
    import subprocess, sys
    body = ("import sys, hashlib; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from helpers import cornell, product_from\n"
            "r = product_from(cornell(), 160, 120, 5, blend=True)\n"
            "for _ in range(3): assert r.TraceFrame()\n"
            "h = hashlib.sha256(); h.update(r.GetRadiance().tobytes()); h.update(r.GetChannel(0).tobytes()); h.update(r.GetChannel(1).tobytes()); h.update(str(r.GetCounters()[:8]).encode())\n"
            "print('IMG', h.hexdigest())\n") % (os.path.dirname(GOLDEN), os.path.dirname(os.path.dirname(GOLDEN)))
    digests = []
    for mix in ({"LUMEN_MI_NOSLP_KERNELS": "none", "LUMEN_MI_NOSLP_KERNELS_EXACT": "none"}, {"LUMEN_MI_NOSLP_KERNELS": "all"}, {}):
        env = dict(os.environ, **mix)
        run = subprocess.run([sys.executable, "-c", body], capture_output=True, text=True, timeout=600, env=env)
        assert run.returncode == 0 and "IMG " in run.stdout, (mix, run.stdout[-500:], run.stderr[-2000:])
        digests.append(run.stdout.split("IMG ")[1].split()[0])
    assert digests[0] == digests[1] == digests[2], digests
    import hashlib
    o = oracle_from(cornell(), 160, 120, 5, blend=True)
    for _ in range(3):
        assert o.trace_frame() == 0
    r = product_from(cornell(), 160, 120, 5, blend=True)
    for _ in range(3):
        assert r.TraceFrame()
    assert np.array_equal(r.GetRadiance().view(np.uint32), o.radiance().view(np.uint32))
    r.close(); o.close()
