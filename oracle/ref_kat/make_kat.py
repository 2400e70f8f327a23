#!/usr/bin/env python3
"""Build tests/golden/ref_kat.npz from the reference's own headers (container-only).

Usage (in the build container, where /root/reference exists):  python oracle/ref_kat/make_kat.py
Compiles gen_kat.cpp / gen_kat2.cpp against /root/reference/Lumen_Engine/LumenPT (headers only, via shim.h) and gen_kat3.cpp together
with the reference's Camera.cpp (its own source file, on the vendored glm), runs them,
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
# third unit: the reference's camera, compiled from its own source file (plain C++ on the vendored glm) + the vendored sutil matrix
L = "/root/reference/Lumen_Engine/Lumen"
exe = "/tmp/lumen_gen_kat3"
subprocess.check_call(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-DNDEBUG", "-w", "-DGLM_ENABLE_EXPERIMENTAL",
                       f"-I{L}/vendor/glm", f"-I{L}/src", f"-I{R}/vendor/Include", f"-I{R}/vendor/Include/Cuda",
                       os.path.join(here, "gen_kat3.cpp"), f"{L}/src/Lumen/Renderer/Camera.cpp", "-o", exe])
for line in subprocess.check_output([exe], text=True).splitlines():
    tag, *vals = line.split()
    rows.setdefault(tag, []).append([float(v) for v in vals])
out = {k: np.asarray(v, dtype=np.float64) for k, v in rows.items()}
dst = os.path.join(here, "..", "..", "tests", "golden", "ref_kat.npz")
np.savez_compressed(dst, **out)
print({k: v.shape for k, v in out.items()}, "->", os.path.normpath(dst), os.path.getsize(dst), "bytes")
