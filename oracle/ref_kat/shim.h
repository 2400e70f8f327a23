// Host-compile shim for the reference's header-only device math (container-only; see gen_kat.cpp).
// It only supplies the CUDA built-ins those headers assume; it contains no reference code.
#pragma once
#include <cmath>
#include <algorithm>
#include <cstdio>
#include <cstdint>
using std::min; using std::max; using std::abs;
#include <cuda_runtime.h>
#include <sutil/vec_math.h>
static inline float saturate(float x) { return fminf(fmaxf(x, 0.f), 1.f); }
static inline void sincosf_shim(float x, float* s, float* c) { *s = sinf(x); *c = cosf(x); }
#define lerp lerp_ref
#define __CUDACC__ 1
