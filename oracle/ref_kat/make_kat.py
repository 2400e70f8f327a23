#!/usr/bin/env python3
"""Build tests/golden/ref_kat.npz from the reference's own headers (container-only).

Usage (in the build container, where /root/reference exists):  python oracle/ref_kat/make_kat.py
Compiles gen_kat.cpp against /root/reference/Lumen_Engine/LumenPT (headers only, via shim.h), runs it,
and stores the rows as float64/uint32 arrays.  Only numbers are committed, never reference text.
"""
import os, subprocess, sys, numpy as np
here = os.path.dirname(os.path.abspath(__file__))
R = "/root/reference/Lumen_Engine/LumenPT"
rows = {}
for src in ("gen_kat.cpp", "gen_kat2.cpp"):        # two translation units: ReSTIRData.h needs __CUDACC__ undefined, disney.cuh needs it defined
    exe = "/tmp/lumen_" + src[:-4]
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-D_GNU_SOURCE", "-DNDEBUG", "-w",
                           f"-I{R}/vendor/Include", f"-I{R}/vendor/Include/Cuda", f"-I{R}/src",
                           os.path.join(here, src), "-o", exe])
    for line in subprocess.check_output([exe], text=True).splitlines():
        tag, *vals = line.split()
        rows.setdefault(tag, []).append([float(v) for v in vals])
out = {k: np.asarray(v, dtype=np.float64) for k, v in rows.items()}
dst = os.path.join(here, "..", "..", "tests", "golden", "ref_kat.npz")
np.savez_compressed(dst, **out)
print({k: v.shape for k, v in out.items()}, "->", os.path.normpath(dst), os.path.getsize(dst), "bytes")
