// Shared device helpers for the gfx950 kernels (wave64, bf16 storage, fp32 math).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "drn.h"

typedef unsigned short bf16_t;   // raw bf16 bits in memory

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;    // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define DRN_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t u) { return __uint_as_float(((uint32_t)u) << 16); }

// round-to-nearest-even f32 -> bf16 (v_cvt_pk_bf16_f32 on gfx950; keeps NaN a NaN)
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 h = (__bf16)f;
    return *reinterpret_cast<bf16_t*>(&h);
}
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }   // round through bf16

typedef __bf16 drn_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float drn_f32x2_t __attribute__((ext_vector_type(2)));
// two f32 -> packed bf16x2 in ONE v_cvt_pk_bf16_f32 (two scalar casts cost 2 cvt + shift + or)
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    drn_f32x2_t f = {lo, hi};
    drn_bf16x2_t b = __builtin_convertvector(f, drn_bf16x2_t);
    return *reinterpret_cast<uint32_t*>(&b);
}
__device__ __forceinline__ float bflo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfhi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    f[0] = bflo(v.x); f[1] = bfhi(v.x); f[2] = bflo(v.y); f[3] = bfhi(v.y);
    f[4] = bflo(v.z); f[5] = bfhi(v.z); f[6] = bflo(v.w); f[7] = bfhi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 v;
    v.x = pack_bf2(f[0], f[1]); v.y = pack_bf2(f[2], f[3]);
    v.z = pack_bf2(f[4], f[5]); v.w = pack_bf2(f[6], f[7]);
    return v;
}

// erf-GELU of the MLP (nn.GELU(), CleanGeneralDIT.py:446): x Phi(x) = max(x, 0) - |x| Phi(-|x|), and
// log2 Phi(-a) = q(a), a degree-5 polynomial fitted on a in [0, 4] (Chebyshev nodes; |err| < 2e-3 in log2 units up to a = 5.5, and
// q -> -inf beyond, so the tail underflows to 0 like the reference's).  8 VALU instructions, one of them transcendental
// (round 2: Abramowitz-Stegun 7.1.26 with v_rcp + v_exp, ~17; the epilogue was 9.4 % of the MLP-up GEMM with the matrix pipe idle).
// Against torch's bf16 GELU over N(0,1) inputs: 99.7 % of the bf16 results identical, the rest 1 ulp (tests: <= 2 ulp, >= 97 %);
// for x < -4.5 the reference's own 1 + erf(x / sqrt 2) cancels in fp32, both forms differ from it there by < 1e-4 absolute.
// The result is rounded to bf16 (2^-9 relative) right after.
#ifndef DRN_GELU_AS
#define DRN_GELU_AS 0      // 1: the round-2 form (A/B timing builds only: tools/build_variants.py)
#endif
__device__ __forceinline__ float gelu_erf_fast(float x) {
#if DRN_GELU_AS
    const float z = x * 0.70710678118654752440f, az = fabsf(z);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float u = p * t * __builtin_amdgcn_exp2f(-1.44269504088896340736f * az * az);
    return 0.5f * x * (z < 0.f ? u : 2.0f - u);
#endif
    const float a = fabsf(x);
    float q = fmaf(-3.593512520e-04f, a, 6.193101872e-03f);
    q = fmaf(q, a, -4.942968488e-02f);
    q = fmaf(q, a, -4.625552893e-01f);
    q = fmaf(q, a, -1.149885178e+00f);
    q = fmaf(q, a, -1.000069737e+00f);
    // exp2, max(x, 0) and the last fma as ONE asm statement: plain fmaxf adds a canonicalising v_max in front (x comes from bit
    // ops), hipcc pairs the last fma of two elements into a v_pk_fma_f32 (no |x| modifier: a v_and per element) - and a
    // transcendental result needs one wait state before a VALU instruction may read it (gfx950 trans-use hazard), which hipcc
    // only provides for instructions it emits itself: here the v_max sits in that slot.
    float r, relu;
    asm("v_exp_f32 %0, %2\n\tv_max_f32 %1, %3, 0\n\tv_fma_f32 %0, -|%3|, %0, %1" : "=&v"(r), "=&v"(relu) : "v"(q), "v"(x));   // r = relu - |x| Phi(-|x|)
    return r;
}

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
// streaming (read-once) 16-byte load
__device__ __forceinline__ uint4 ld16_nt(const void* p) {
    u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

#define DRN_CHECK_ARG(cond)            \
    do {                               \
        if (!(cond)) return DRN_EINVAL; \
    } while (0)

static inline int drn_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DRN_OK : (int)e;
}
