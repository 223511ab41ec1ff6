// Weight-streaming GEMV family for the batch-1 vectors of the DiT (timestep MLP, AdaLN-LoRA for every
// sub-block, the single-key cross-attention).  HBM-bound: each weight byte is read once, 16 B per lane,
// one wave per output row, fp32 accumulate, bf16 result with the reference's rounding points.
#include "drn_common.h"

__device__ __forceinline__ float silu_bf(float x) {   // bf16(silu(x)) as torch computes it for a bf16 input
    return rbf(x / (1.0f + expf(-x)));
}

template <int ACT>
__global__ __launch_bounds__(256) void gemv_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ W,
                                                   bf16_t* __restrict__ y, int64_t N, int K, int64_t x_gs, int64_t x_bs,
                                                   int64_t w_gs, int64_t y_gs, int64_t y_bs,
                                                   const bf16_t* __restrict__ add, int64_t add_gs, int64_t add_bs,
                                                   const bf16_t* __restrict__ mul, int64_t mul_gs, int64_t mul_bs) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int g = blockIdx.y, b = blockIdx.z;
    const bf16_t* xr = x + g * x_gs + b * x_bs;
    const bf16_t* wr = W + g * w_gs + n * (int64_t)K;
    float acc = 0.f;
    for (int c = lane * 8; c < K; c += 512) {
        float wv[8], xv[8];
        unpack8(ld16_nt(wr + c), wv);
        unpack8(*reinterpret_cast<const uint4*>(xr + c), xv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xa = ACT == DRN_ACT_SILU ? silu_bf(xv[j]) : xv[j];
            acc = fmaf(wv[j], xa, acc);
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        float r = rbf(acc);
        if (add) r = rbf(r + bf2f(add[g * add_gs + b * add_bs + n]));
        if (mul) r = rbf(bf2f(mul[g * mul_gs + b * mul_bs + n]) * r);
        y[g * y_gs + b * y_bs + n] = f2bf(r);
    }
}

extern "C" int drn_gemv_bf16(const void* x, const void* W, void* y, int64_t N, int64_t K, int groups, int batch,
                             int64_t x_gstride, int64_t x_bstride, int64_t w_gstride, int64_t y_gstride,
                             int64_t y_bstride, const void* add, int64_t add_gstride, int64_t add_bstride,
                             const void* mul, int64_t mul_gstride, int64_t mul_bstride, int act, void* stream) {
    DRN_CHECK_ARG(x && W && y && N > 0 && K > 0 && K % 8 == 0 && groups > 0 && batch > 0);
    DRN_CHECK_ARG(groups <= 65535 && batch <= 65535);
    DRN_CHECK_ARG(x_gstride % 8 == 0 && x_bstride % 8 == 0 && w_gstride % 8 == 0);
    DRN_CHECK_ARG(act == DRN_ACT_NONE || act == DRN_ACT_SILU);
    dim3 grid((unsigned)((N + 3) / 4), (unsigned)groups, (unsigned)batch), block(256);
    hipStream_t st = (hipStream_t)stream;
#define ARGS                                                                                                       \
    (const bf16_t*)x, (const bf16_t*)W, (bf16_t*)y, N, (int)K, x_gstride, x_bstride, w_gstride, y_gstride, y_bstride, \
        (const bf16_t*)add, add_gstride, add_bstride, (const bf16_t*)mul, mul_gstride, mul_bstride
    if (act == DRN_ACT_SILU) gemv_kernel<DRN_ACT_SILU><<<grid, block, 0, st>>>(ARGS);
    else gemv_kernel<DRN_ACT_NONE><<<grid, block, 0, st>>>(ARGS);
#undef ARGS
    return drn_launch_status();
}
