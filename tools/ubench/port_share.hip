// Micro-benchmark: how much VALU issue does a partner wave get beside an MFMA stream on the same SIMD?
// 8 waves per workgroup (waves w and w + 4 share a SIMD, as in attention.hip): waves 0-3 run back-to-back MFMAs at s_setprio 1
// (32 x v_mfma_f32_32x32x16_bf16 or 64 x v_mfma_f32_16x16x32_bf16 per "segment": the same FLOPs, the same 1024 matrix-pipe cycles),
// waves 4-7 run a softmax-shaped VALU block (32 fma + 32 exp + 36 add + 16 cvt_pk + 23 max = what S(t) of the attention issues).
// Reported: cycles per MFMA segment and per VALU block, each alone and both together, for the two MFMA shapes.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/port_share.hip -o build/port_share && build/port_share
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int SHAPE>   // 0: 32x32x16, 1: 16x16x32
__device__ __forceinline__ void mfma_segment(f32x16_t (&acc32)[4], f32x4_t (&acc16)[16], const bf16x8_t& a, const bf16x8_t& b) {
    if (SHAPE == 0) {
#pragma unroll
        for (int i = 0; i < 32; ++i) acc32[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc32[i & 3], 0, 0, 0);
    } else {
#pragma unroll
        for (int i = 0; i < 64; ++i) acc16[i & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc16[i & 15], 0, 0, 0);
    }
}

__device__ __forceinline__ float valu_block(float (&s)[32], float mc, float sc) {
    // the softmax of one 64-key tile for one query row pair: max tree, fma + exp2, row sum, bf16 pack
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 32; r += 2) mx = fmaxf(mx, fmaxf(s[r], s[(r + 1) & 31]));
    float sum = 0.f;
    uint32_t pk = 0;
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[r], sc, -mc));
        sum += p;
        s[r] = p * 0.5f + mx * 1e-9f;                      // keep the chain alive for the next block
    }
#pragma unroll
    for (int r = 0; r < 32; r += 2) {
        uint32_t w;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(s[r]), "v"(s[r + 1]));
        pk ^= w;
    }
    return sum + __uint_as_float(pk & 0x007fffffu);
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int SHAPE, int MODE>   // MODE bit 0: MFMA waves run, bit 1: VALU waves run, bits 2..: LDS-DMA pieces (1 KiB) per MFMA segment
__global__ __launch_bounds__(512, 2) void port_kernel(unsigned long long* out, float* sink, int iters, const char* src) {
    __shared__ __attribute__((aligned(1024))) char lds[64 * 1024];
    constexpr int PIECES = MODE >> 2;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    unsigned long long t0 = 0, t1 = 0;
    float res = 0.f;
    if (wave < 4) {
        if (MODE & 1) {
            bf16x8_t a, b;
#pragma unroll
            for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + ((lane * 37 + i * 11) & 0x7f)); b[i] = (short)(0xbf00 + ((lane * 53 + i * 29) & 0xff)); }
            f32x16_t acc32[4] = {};
            f32x4_t acc16[16] = {};
            __builtin_amdgcn_s_setprio(1);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
            const uint32_t voff = (uint32_t)(lane * 16 + wave * 4096);
            for (int it = 0; it < iters; ++it) {
                if (PIECES == 0) {
                    mfma_segment<SHAPE>(acc32, acc16, a, b);
                } else {
                    // the segment in PIECES parts, one 1 KiB global -> LDS DMA (L2-resident source) after each part
#pragma unroll
                    for (int p = 0; p < PIECES; ++p) {
                        if (SHAPE == 0) {
#pragma unroll
                            for (int i = 0; i < 32 / PIECES; ++i) acc32[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc32[i & 3], 0, 0, 0);
                        } else {
#pragma unroll
                            for (int i = 0; i < 64 / PIECES; ++i) acc16[i & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc16[i & 15], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        __builtin_amdgcn_global_load_lds((gptr_t)(src + voff + ((it * PIECES + p) & 15) * 16384), (lptr_t)(lds + wave * 8192 + p * 1024), 16, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                }
                asm volatile("" : "+v"(a), "+v"(b));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) res += acc32[i][0] + acc32[i][7];
#pragma unroll
            for (int i = 0; i < 16; ++i) res += acc16[i][0];
        }
    } else {
        if (MODE & 2) {
            float s[32];
#pragma unroll
            for (int r = 0; r < 32; ++r) s[r] = (float)((lane * 7 + r * 13) & 31) * 0.25f - 3.0f;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
            for (int it = 0; it < iters; ++it) res += valu_block(s, 1.5f, 0.127f);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        }
    }
    if (lane == 0) out[(size_t)blockIdx.x * 8 + wave] = t1 - t0;
    if (res == 123.456f) sink[threadIdx.x] = res + lds[threadIdx.x];
}

template <int SHAPE, int MODE>
static void run(const char* name, int iters) {
    const int blocks = 256;
    unsigned long long* d;
    float* sink;
    char* src;
    hipMalloc(&d, blocks * 8 * sizeof(unsigned long long));
    hipMalloc(&sink, 512 * sizeof(float));
    hipMalloc(&src, 1 << 20);
    hipMemset(src, 1, 1 << 20);
    for (int rep = 0; rep < 3; ++rep) port_kernel<SHAPE, MODE><<<blocks, 512>>>(d, sink, iters, src);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) port_kernel<SHAPE, MODE><<<blocks, 512>>>(d, sink, iters, src);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> m, v;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < 8; ++w) (w < 4 ? m : v).push_back((double)h[b * 8 + w] / iters);
    std::sort(m.begin(), m.end());
    std::sort(v.begin(), v.end());
    printf("%-34s MFMA segment %7.0f cycles   VALU block %7.0f cycles   kernel %.3f ms per launch\n", name, m[m.size() / 2], v[v.size() / 2], ms / 5);
    hipFree(d); hipFree(sink); hipFree(src);
}

int main() {
    const int iters = 2000;
    run<0, 1>("32x32x16: MFMA waves alone", iters);
    run<1, 1>("16x16x32: MFMA waves alone", iters);
    run<0, 2>("VALU waves alone", iters);
    run<0, 3>("32x32x16 beside the VALU block", iters);
    run<1, 3>("16x16x32 beside the VALU block", iters);
    // 4 (as the attention's M segment) and 8 (as a GEMM K step per wave) LDS-DMA pieces inside the MFMA segment
    run<0, 1 + 16>("32x32x16 + 4 DMA pieces, alone", iters);
    run<1, 1 + 16>("16x16x32 + 4 DMA pieces, alone", iters);
    run<0, 3 + 16>("32x32x16 + 4 DMA pieces | VALU block", iters);
    run<1, 3 + 16>("16x16x32 + 4 DMA pieces | VALU block", iters);
    run<1, 1 + 32>("16x16x32 + 8 DMA pieces, alone", iters);
    run<1, 3 + 32>("16x16x32 + 8 DMA pieces | VALU block", iters);
    return 0;
}
