#!/usr/bin/env python3
"""usage (GPU box): python tools/fp16_blend_chain.py > profiles/r05_fp16_blend_chain.txt — what a reference maintainer should expect from decision D1 on the BLEND:
the reference keeps its pixel buffers as half4 surfaces and blends in binary16 (GPUMergeOutputChannels.cu:20-72 with the operators of Half4.h:105-196:
merged = half(DIRECT) + half(INDIRECT) by __hadd2; new = ((old * half(n)) + merged) / half(n + 1) by __hmul2 / __hadd2 / __h2div), this build accumulates and
blends in fp32 and rounds once on export (lumen_mi_get_radiance_half4).  Per-frame fp32 channels come from the product with blending off (C2: 1440p, depth 6, fast
mode is irrelevant here: exact); both chains are then replayed on the host: numpy float16 arithmetic is correctly rounded per operation, __h2div is not guaranteed
to be — the fp16 chain below is therefore the BEST case of the reference's arithmetic."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def chains(frames_direct, frames_indirect):
    """Per-frame DIRECT / INDIRECT fp32 channels -> {n: (fp32 blend rounded once, fp16 blend chain)} for every n."""
    acc32 = None; acc16 = None; out = {}
    with np.errstate(over="ignore", invalid="ignore"):
        for k, (d, i) in enumerate(zip(frames_direct, frames_indirect)):
            m32 = d[..., :3] + i[..., :3]                                            # lm_k_merge_output: channel sum in fp32
            acc32 = m32 if k == 0 else ((acc32 * np.float32(k)) + m32) / np.float32(k + 1)
            m16 = d[..., :3].astype(np.float16) + i[..., :3].astype(np.float16)     # the channels as the reference stores them, summed by __hadd2
            acc16 = m16 if k == 0 else ((acc16 * np.float16(k)) + m16) / np.float16(k + 1)
            out[k + 1] = (acc32.astype(np.float16), acc16.copy())
    return out


def rel_l2_finite(a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    ok = np.isfinite(a).all(-1) & np.isfinite(b).all(-1)
    return float(np.sqrt(np.sum((a[ok] - b[ok]) ** 2) / np.sum(b[ok] ** 2))), float(1.0 - ok.mean())


def main():
    from helpers import product_from
    from lumenrenderer_amd import scenes
    W, H, D, N = 2560, 1440, 6, 8
    r = product_from(scenes.sponza_standin(), W, H, D, blend=False)
    fd, fi = [], []
    for _ in range(N):
        assert r.TraceFrame()
        fd.append(r.GetChannel(0).copy()); fi.append(r.GetChannel(1).copy())
    r.close()
    c = chains(fd, fi)
    print(f"C2 (sponza stand-in, {W}x{H}, depth {D}), {N} TraceFrames: fp32 blend rounded once to binary16 (this build's export) against the reference's binary16 blend chain")
    for n in (1, 2, 4, 8):
        e, lost = rel_l2_finite(c[n][0], c[n][1])
        ref32 = None
        print(f"  {n} blended frames: rel-L2 {e:.3e}   (pixels overflowing binary16 in either chain, excluded: {lost:.2e})")
    # and each against the fp32 running mean itself
    acc = None
    for k in range(N):
        m = fd[k][..., :3] + fi[k][..., :3]; acc = m if k == 0 else ((acc * np.float32(k)) + m) / np.float32(k + 1)
        if k + 1 in (4, 8):
            a, _ = rel_l2_finite(c[k + 1][0], acc); b, _ = rel_l2_finite(c[k + 1][1], acc)
            print(f"  {k + 1} frames against the fp32 mean: rounded once {a:.3e}, binary16 chain {b:.3e}")


if __name__ == "__main__":
    main()
