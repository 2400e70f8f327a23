#!/usr/bin/env python3
"""Build tests/golden/ref_kat.npz from the reference's own headers (container-only).

Usage (in the build container, where /root/reference exists):  python oracle/ref_kat/make_kat.py
Compiles gen_kat.cpp / gen_kat2.cpp against /root/reference/Lumen_Engine/LumenPT (headers only, via shim.h) and gen_kat3.cpp together
with the reference's Camera.cpp (its own source file, on the vendored glm); gen_kat4.cpp is compiled around the reference's own text
of Resample / CombineBiased / CombineUnbiased (ReSTIRKernels.cu:1123-1325) and HaltonSequence (GPUGeneratePrimRay.cu:8-26): those line
ranges — plain __device__ functions without __global__, surface, texture or atomic operations — are sliced into /tmp (never into the
repository) and #included behind the include order of the reference's .cu.  Runs them and stores the rows as float64 arrays
(float32 values and integers are exact in float64).  Only numbers are committed, never reference text.
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
# fourth unit: plain __device__ functions of the ReSTIR / primary-ray kernels, sliced by line range from the reference's .cu files
def slice_lines(path, first, last, dst):
    with open(path) as f:
        lines = f.readlines()[first - 1:last]
    with open(dst, "w") as f:
        f.writelines(lines)
slice_lines(f"{R}/src/CUDAKernels/ReSTIRKernels.cu", 1123, 1325, "/tmp/lumen_slice_restir.inc")
slice_lines(f"{R}/src/CUDAKernels/WaveFrontKernels/GPUGeneratePrimRay.cu", 8, 26, "/tmp/lumen_slice_halton.inc")
exe = "/tmp/lumen_gen_kat4"
subprocess.check_call(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-D_GNU_SOURCE", "-DNDEBUG", "-w", "-DGLM_ENABLE_EXPERIMENTAL",
                       f"-I{R}/vendor/Include", f"-I{R}/vendor/Include/Cuda", f"-I{R}/src", f"-I{L}/vendor/glm",
                       os.path.join(here, "gen_kat4.cpp"), "-o", exe])
for line in subprocess.check_output([exe], text=True).splitlines():
    tag, *vals = line.split()
    rows.setdefault(tag, []).append([float(v) for v in vals])
out = {k: np.asarray(v, dtype=np.float64) for k, v in rows.items()}
dst = os.path.join(here, "..", "..", "tests", "golden", "ref_kat.npz")
np.savez_compressed(dst, **out)
print({k: v.shape for k, v in out.items()}, "->", os.path.normpath(dst), os.path.getsize(dst), "bytes")
