// HBM-bound token-local kernels of the DiT hot path and the sampler (gfx950, wave64).
// One read + one write of each activation row; all bf16 traffic is 16 B per lane.
// fp contraction is off in this file: several kernels are bit-exact restatements of unfused torch fp32 ops.
#pragma clang fp contract(off)
#include "drn_common.h"

// ------------------------------------------------------------------------------------------------
// LayerNorm (no affine) + AdaLN modulate, optional broadcast pre-add.
// CleanGeneralDIT.py:7-11, :481, :506, :517 (rounding points: SURVEY.md Appendix C).
//
// A row is NCH chunk columns of 64 lanes x 8 elements (chunk i of lane l = elements (l + 64 i) * 8 .. + 7).  Two kernels share ONE
// summation tree for the mean and the variance, so they return the same bits and the launcher may pick either by row count:
//     per lane and chunk: the 8 elements in order;  per lane and GROUP of NCH / 4 chunks: the chunks in order;
//     per group: the 64-lane butterfly;  total = (G0 + G1) + (G2 + G3).
//   * ln_modulate_kernel<NCH>:   one wave per row (a row stays in one wave's registers) - many rows;
//   * ln_modulate_kernel4<NCH>:  four waves per row, wave w owns group w, partial sums meet in LDS - few rows (cfg 1: 256 rows
//     on 64 workgroups of the one-wave kernel took 11.7 us per call, a chain of three memory round trips on a quarter of the CUs).
// NCH < 4 (D <= 1024: the tiny test networks) keeps a single butterfly and has no four-wave form.
template <int NCH>
__device__ __forceinline__ float ln_tree_sum(const float (&part)[NCH]) {
    if (NCH < 4) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) s += part[i];
        return wave_sum(s);
    }
    constexpr int G = NCH >= 4 ? NCH / 4 : 1;
    float g[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < G; ++i) t += part[(NCH >= 4 ? w * G : 0) + i];
        g[w] = t;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {                   // four independent butterflies, interleaved
#pragma unroll
        for (int w = 0; w < 4; ++w) g[w] += __shfl_xor(g[w], o, 64);
    }
    return (g[0] + g[1]) + (g[2] + g[3]);
}

template <int NCH>
__global__ __launch_bounds__(256) void ln_modulate_kernel(bf16_t* __restrict__ x, const bf16_t* __restrict__ add,
                                                          const bf16_t* __restrict__ shift,
                                                          const bf16_t* __restrict__ scale, bf16_t* __restrict__ h,
                                                          int64_t rows, int D, int64_t rpb, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t b = row / rpb;
    bf16_t* xr = x + row * D;
    const bf16_t* ar = add ? add + b * D : nullptr;
    float v[NCH][8];
    float part[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * 8;
        float s = 0.f;
        if (c < D) {
            uint4 raw = *reinterpret_cast<const uint4*>(xr + c);
            unpack8(raw, v[i]);
            if (ar) {
                float a[8];
                unpack8(*reinterpret_cast<const uint4*>(ar + c), a);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[i][j] = rbf(v[i][j] + a[j]);
                *reinterpret_cast<uint4*>(xr + c) = pack8(v[i]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[i][j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
        }
        part[i] = s;
    }
    const float mean = ln_tree_sum<NCH>(part) / (float)D;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * 8;
        float q = 0.f;
        if (c < D) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = v[i][j] - mean;
                q += d * d;
            }
        }
        part[i] = q;
    }
    const float var = ln_tree_sum<NCH>(part) / (float)D;
    const float rstd = 1.0f / sqrtf(var + eps);
    const bf16_t* sh = shift + b * D;
    const bf16_t* sc = scale + b * D;
    bf16_t* hr = h + row * D;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < D) {
            float fs[8], fc[8], o[8];
            unpack8(*reinterpret_cast<const uint4*>(sh + c), fs);
            unpack8(*reinterpret_cast<const uint4*>(sc + c), fc);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float n = rbf((v[i][j] - mean) * rstd);
                const float s1 = rbf(1.0f + fc[j]);
                o[j] = rbf(rbf(n * s1) + fs[j]);
            }
            *reinterpret_cast<uint4*>(hr + c) = pack8(o);
        }
    }
}

// four waves per row: wave w owns chunks w * G .. w * G + G - 1 (G = NCH / 4); D must fill all NCH chunk columns' lanes or leave
// whole lanes empty exactly as above (c < D test per chunk)
template <int NCH>
__global__ __launch_bounds__(256) void ln_modulate_kernel4(bf16_t* __restrict__ x, const bf16_t* __restrict__ add,
                                                           const bf16_t* __restrict__ shift,
                                                           const bf16_t* __restrict__ scale, bf16_t* __restrict__ h,
                                                           int64_t rows, int D, int64_t rpb, float eps) {
    static_assert(NCH >= 4 && NCH % 4 == 0, "four-wave form needs whole chunk groups");
    constexpr int G = NCH / 4;
    __shared__ float red[2][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t row = blockIdx.x;
    const int64_t b = row / rpb;
    bf16_t* xr = x + row * D;
    const bf16_t* ar = add ? add + b * D : nullptr;
    const bf16_t* sh = shift + b * D;
    const bf16_t* sc = scale + b * D;
    float v[G][8], fs[G][8], fc[G][8];
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int c = (lane + 64 * (w * G + g)) * 8;
        float s = 0.f;
        if (c < D) {
            unpack8(*reinterpret_cast<const uint4*>(xr + c), v[g]);
            unpack8(*reinterpret_cast<const uint4*>(sh + c), fs[g]);     // (independent of the statistics: in flight with the row)
            unpack8(*reinterpret_cast<const uint4*>(sc + c), fc[g]);
            if (ar) {
                float a[8];
                unpack8(*reinterpret_cast<const uint4*>(ar + c), a);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[g][j] = rbf(v[g][j] + a[j]);
                *reinterpret_cast<uint4*>(xr + c) = pack8(v[g]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[g][j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[g][j] = 0.f;
        }
        t += s;
    }
    t = wave_sum(t);
    if (lane == 0) red[0][w] = t;
    __syncthreads();
    const float mean = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)D;
    float tq = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int c = (lane + 64 * (w * G + g)) * 8;
        float q = 0.f;
        if (c < D) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = v[g][j] - mean;
                q += d * d;
            }
        }
        tq += q;
    }
    tq = wave_sum(tq);
    if (lane == 0) red[1][w] = tq;
    __syncthreads();
    const float var = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)D;
    const float rstd = 1.0f / sqrtf(var + eps);
    bf16_t* hr = h + row * D;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int c = (lane + 64 * (w * G + g)) * 8;
        if (c < D) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float n = rbf((v[g][j] - mean) * rstd);
                const float s1 = rbf(1.0f + fc[g][j]);
                o[j] = rbf(rbf(n * s1) + fs[g][j]);
            }
            *reinterpret_cast<uint4*>(hr + c) = pack8(o);
        }
    }
}

// Split-K sum + gated residual + the NEXT sub-block's LayerNorm / modulate in one pass (few-token shapes: cfg 1 ran
// gemm_splitk_epilogue_kernel<2> and ln_modulate back to back on 2 MB of activations, ~9 us + ~6 us + a launch gap, 56 times per forward):
//   a     = part[0][row] + part[1][row] + ...            fp32, slices in order        (= gemm_splitk_epilogue_kernel)
//   x     = bf16(x + bf16(gate * bf16(a)))               written back                 (= its DRN_EPI_GATE_RES epilogue)
//   x     = bf16(x + add_vec) if add_vec                 written back instead         (= ln_modulate's broadcast pre-add)
//   h     = modulate(LayerNorm(x))                       four waves per row           (= ln_modulate_kernel4: same summation tree)
// Every step rounds where the separate kernels round: the same bits as epilogue kernel + ln_modulate.
template <int NCH>
__global__ __launch_bounds__(256) void splitk_gate_res_ln_kernel(const float* __restrict__ part, int splits, int64_t part_stride,
                                                                 bf16_t* __restrict__ x, const bf16_t* __restrict__ gate,
                                                                 const bf16_t* __restrict__ add, const bf16_t* __restrict__ shift,
                                                                 const bf16_t* __restrict__ scale, bf16_t* __restrict__ h,
                                                                 int D, int64_t rpb, float eps) {
    static_assert(NCH >= 4 && NCH % 4 == 0, "four waves per row");
    constexpr int G = NCH / 4;
    __shared__ float red[2][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t row = blockIdx.x;
    const int64_t b = row / rpb;
    bf16_t* xr = x + row * D;
    const float* pr = part + row * D;
    const bf16_t* gr = gate + b * D;
    const bf16_t* ar = add ? add + b * D : nullptr;
    const bf16_t* sh = shift + b * D;
    const bf16_t* sc = scale + b * D;
    float v[G][8], fs[G][8], fc[G][8];
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int c = (lane + 64 * (w * G + g)) * 8;
        float s = 0.f;
        if (c < D) {
            // slices summed in order (as gemm_splitk_epilogue_kernel), loaded four at a time: eight 16-byte loads in flight
            f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(pr + c), a1 = *reinterpret_cast<const f32x4_t*>(pr + c + 4);
            int sp = 1;
            for (; sp + 4 <= splits; sp += 4) {
                f32x4_t l0[4], l1[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    l0[u] = *reinterpret_cast<const f32x4_t*>(pr + (sp + u) * part_stride + c);
                    l1[u] = *reinterpret_cast<const f32x4_t*>(pr + (sp + u) * part_stride + c + 4);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    a0 += l0[u];
                    a1 += l1[u];
                }
            }
            for (; sp < splits; ++sp) {
                a0 += *reinterpret_cast<const f32x4_t*>(pr + sp * part_stride + c);
                a1 += *reinterpret_cast<const f32x4_t*>(pr + sp * part_stride + c + 4);
            }
            float xv[8], gv[8];
            unpack8(*reinterpret_cast<const uint4*>(xr + c), xv);
            unpack8(*reinterpret_cast<const uint4*>(gr + c), gv);
            unpack8(*reinterpret_cast<const uint4*>(sh + c), fs[g]);
            unpack8(*reinterpret_cast<const uint4*>(sc + c), fc[g]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float acc = j < 4 ? a0[j] : a1[j - 4];
                v[g][j] = rbf(xv[j] + rbf(gv[j] * rbf(acc)));
            }
            if (ar) {
                float a[8];
                unpack8(*reinterpret_cast<const uint4*>(ar + c), a);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[g][j] = rbf(v[g][j] + a[j]);
            }
            *reinterpret_cast<uint4*>(xr + c) = pack8(v[g]);
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[g][j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[g][j] = 0.f;
        }
        t += s;
    }
    t = wave_sum(t);
    if (lane == 0) red[0][w] = t;
    __syncthreads();
    const float mean = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)D;
    float tq = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int c = (lane + 64 * (w * G + g)) * 8;
        float q = 0.f;
        if (c < D) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = v[g][j] - mean;
                q += d * d;
            }
        }
        tq += q;
    }
    tq = wave_sum(tq);
    if (lane == 0) red[1][w] = tq;
    __syncthreads();
    const float var = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)D;
    const float rstd = 1.0f / sqrtf(var + eps);
    bf16_t* hr = h + row * D;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int c = (lane + 64 * (w * G + g)) * 8;
        if (c < D) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float n = rbf((v[g][j] - mean) * rstd);
                const float s1 = rbf(1.0f + fc[g][j]);
                o[j] = rbf(rbf(n * s1) + fs[g][j]);
            }
            *reinterpret_cast<uint4*>(hr + c) = pack8(o);
        }
    }
}

extern "C" int drn_splitk_gate_res_ln_modulate(const void* partials, int splits, void* x, const void* gate, const void* add_vec,
                                               const void* shift, const void* scale, void* h, int64_t rows, int64_t D,
                                               int64_t rows_per_batch, float eps, void* stream) {
    DRN_CHECK_ARG(partials && x && gate && shift && scale && h && splits >= 1 && rows >= 0 && rows < (1ll << 31));
    DRN_CHECK_ARG(D > 1024 && D % 8 == 0 && D <= 8192 && rows_per_batch > 0 && ((uintptr_t)partials & 15) == 0);
    if (rows == 0) return DRN_OK;
    const int nch = (int)((D + 511) / 512);
    dim3 grid((unsigned)rows), block(256);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(N)                                                                                                     \
    splitk_gate_res_ln_kernel<N><<<grid, block, 0, st>>>((const float*)partials, splits, rows * D, (bf16_t*)x,       \
        (const bf16_t*)gate, (const bf16_t*)add_vec, (const bf16_t*)shift, (const bf16_t*)scale, (bf16_t*)h, (int)D,  \
        rows_per_batch, eps)
    if (nch <= 4) LAUNCH(4);
    else if (nch <= 8) LAUNCH(8);
    else LAUNCH(16);
#undef LAUNCH
    return drn_launch_status();
}

static int g_ln_force = -1;      // tests: 0 = one wave per row always, 1 = four waves per row wherever it exists, -1 = by row count
extern "C" void drn_ln_force_kernel(int which) { g_ln_force = which; }

extern "C" int drn_ln_modulate(void* x, const void* add_vec, const void* shift, const void* scale, void* h,
                               int64_t rows, int64_t D, int64_t rows_per_batch, float eps, void* stream) {
    DRN_CHECK_ARG(x && shift && scale && h && rows >= 0 && D > 0 && D % 8 == 0 && D <= 8192 && rows_per_batch > 0);
    if (rows == 0) return DRN_OK;
    hipStream_t st = (hipStream_t)stream;
    const int nch = (int)((D + 511) / 512);
    // few rows: four waves per row (same bits as one wave per row: shared summation tree) - 4 x the waves in flight
    // (many rows, measured at 18432 x 4096 with tools/lnbench.py: without the pre-add 77-81 us one wave per row, 64-67 us four
    //  waves per row; with it 82-95 vs 85-95 us: a draw, left on the one-wave kernel)
    const bool four = nch > 2 && rows < (1ll << 31) && (g_ln_force == 1 || (g_ln_force < 0 && (rows <= 4096 || !add_vec)));
    dim3 grid(four ? (unsigned)rows : (unsigned)((rows + 3) / 4)), block(256);
#define LAUNCH(K, N)                                                                                                  \
    K<N><<<grid, block, 0, st>>>((bf16_t*)x, (const bf16_t*)add_vec, (const bf16_t*)shift, (const bf16_t*)scale,      \
                                 (bf16_t*)h, rows, (int)D, rows_per_batch, eps)
    if (nch <= 1) LAUNCH(ln_modulate_kernel, 1);
    else if (nch <= 2) LAUNCH(ln_modulate_kernel, 2);
    else if (nch <= 4) { if (four) LAUNCH(ln_modulate_kernel4, 4); else LAUNCH(ln_modulate_kernel, 4); }
    else if (nch <= 8) { if (four) LAUNCH(ln_modulate_kernel4, 8); else LAUNCH(ln_modulate_kernel, 8); }
    else { if (four) LAUNCH(ln_modulate_kernel4, 16); else LAUNCH(ln_modulate_kernel, 16); }
#undef LAUNCH
    return drn_launch_status();
}

// ------------------------------------------------------------------------------------------------
// x[rows, D] <- bf16(x + vec[batch])   (broadcast cross-attention residual, CleanGeneralDIT.py:517 + F8)
__global__ __launch_bounds__(256) void bcast_add_kernel(bf16_t* __restrict__ x, const bf16_t* __restrict__ vec,
                                                        int64_t nchunks, int dchunks, int64_t rpb) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / dchunks;
        const int c = (int)(i - row * dchunks);
        const int64_t b = row / rpb;
        float a[8], v[8];
        unpack8(*reinterpret_cast<const uint4*>(x + i * 8), v);
        unpack8(*reinterpret_cast<const uint4*>(vec + (b * dchunks + c) * 8), a);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = rbf(v[j] + a[j]);
        *reinterpret_cast<uint4*>(x + i * 8) = pack8(v);
    }
}

extern "C" int drn_bcast_add(void* x, const void* vec, int64_t rows, int64_t D, int64_t rows_per_batch, void* stream) {
    DRN_CHECK_ARG(x && vec && rows >= 0 && D > 0 && D % 8 == 0 && rows_per_batch > 0);
    if (rows == 0) return DRN_OK;
    const int64_t nchunks = rows * (D / 8);
    int64_t blocks = (nchunks + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    bcast_add_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>((bf16_t*)x, (const bf16_t*)vec,
                                                                                    nchunks, (int)(D / 8), rows_per_batch);
    return drn_launch_status();
}

// ------------------------------------------------------------------------------------------------
// y[b][a][:] = x[a][b][:], 16-byte chunks (regroup around the sequence-parallel all-to-all; C >= 512 in practice, so every
// run of C elements is a coalesced >= 1 KB read and write)
__global__ __launch_bounds__(256) void permute_021_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int64_t A,
                                                          int64_t B, int64_t cch) {
    const int64_t n = A * B * cch;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t c = i % cch;
        const int64_t ba = i / cch;           // = b * A + a  (output order)
        const int64_t a = ba % A, b = ba / A;
        y[i] = x[(a * B + b) * cch + c];
    }
}

extern "C" int drn_permute_021(const void* x, void* y, int64_t A, int64_t B, int64_t C, void* stream) {
    DRN_CHECK_ARG(x && y && x != y && A >= 0 && B >= 0 && C > 0 && C % 8 == 0);
    DRN_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0);
    const int64_t n = A * B * (C / 8);
    if (n == 0) return DRN_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    permute_021_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>((const uint4*)x, (uint4*)y, A, B, C / 8);
    return drn_launch_status();
}

// ------------------------------------------------------------------------------------------------
// RMSNorm over the last dim (CleanGeneralDIT.py:14-33).  One wave per row, any D % 8 == 0.
__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                      bf16_t* __restrict__ y, int64_t rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* xr = x + row * D;
    float s = 0.f;
    for (int c = lane * 8; c < D; c += 512) {
        float v[8];
        unpack8(*reinterpret_cast<const uint4*>(xr + c), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j] * v[j];
    }
    const float r = 1.0f / sqrtf(wave_sum(s) / (float)D + eps);
    bf16_t* yr = y + row * D;
    for (int c = lane * 8; c < D; c += 512) {
        float v[8], g[8];
        unpack8(*reinterpret_cast<const uint4*>(xr + c), v);
        unpack8(*reinterpret_cast<const uint4*>(w + c), g);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (v[j] * r) * g[j];
        *reinterpret_cast<uint4*>(yr + c) = pack8(v);
    }
}

extern "C" int drn_rmsnorm(const void* x, const void* w, void* y, int64_t rows, int64_t D, float eps, void* stream) {
    DRN_CHECK_ARG(x && w && y && rows >= 0 && D > 0 && D % 8 == 0);
    if (rows == 0) return DRN_OK;
    rmsnorm_kernel<<<dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(
        (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rows, (int)D, eps);
    return drn_launch_status();
}

// ------------------------------------------------------------------------------------------------
// per-head RMSNorm(q,k) + RoPE, in place.  8 lanes per 128-wide head row (lane j owns elements 8j..8j+7 and
// 64+8j..64+8j+7, i.e. both members of every rotate_half pair), 8 head rows per wave.
// CleanGeneralDIT.py:288-295 (norm), :45-84 (RoPE; cos/sin tables are host-built bf16).
__global__ __launch_bounds__(256) void qk_norm_rope_kernel(bf16_t* __restrict__ q, bf16_t* __restrict__ k,
                                                           const bf16_t* __restrict__ wq, const bf16_t* __restrict__ wk,
                                                           const bf16_t* __restrict__ cs, const bf16_t* __restrict__ sn,
                                                           int64_t tokens, int heads, int64_t ldq, int64_t ldk,
                                                           int64_t tpb, int64_t pos_offset, float eps) {
    const int lane = threadIdx.x & 63;
    const int hr = lane >> 3, j = lane & 7;
    const int hgroups = (heads + 7) / 8;
    const int nwhich = (q ? 1 : 0) + (k ? 1 : 0);          // either tensor may be NULL (q and k handled by separate calls)
    const int64_t items = tokens * hgroups * nwhich;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t it = wave0; it < items; it += nwaves) {
        const int which = nwhich == 2 ? (int)(it & 1) : (q ? 0 : 1);
        const int64_t r = nwhich == 2 ? (it >> 1) : it;
        const int64_t tok = r / hgroups;
        const int head = (int)(r - tok * hgroups) * 8 + hr;
        const bool act = head < heads;
        bf16_t* base = (which ? k + tok * ldk : q + tok * ldq) + (int64_t)head * 128;
        const bf16_t* w = which ? wk : wq;
        float lo[8], hi[8];
        if (act) {
            unpack8(*reinterpret_cast<const uint4*>(base + 8 * j), lo);
            unpack8(*reinterpret_cast<const uint4*>(base + 64 + 8 * j), hi);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) lo[i] = hi[i] = 0.f;
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += lo[i] * lo[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) s += hi[i] * hi[i];
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        const float rinv = 1.0f / sqrtf(s / 128.0f + eps);
        float wl[8], wh[8];
        unpack8(*reinterpret_cast<const uint4*>(w + 8 * j), wl);
        unpack8(*reinterpret_cast<const uint4*>(w + 64 + 8 * j), wh);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            lo[i] = rbf((lo[i] * rinv) * wl[i]);
            hi[i] = rbf((hi[i] * rinv) * wh[i]);
        }
        if (cs) {
            const int64_t p = pos_offset + (tok % tpb);
            float cl[8], ch[8], sl[8], sh[8];
            unpack8(*reinterpret_cast<const uint4*>(cs + p * 128 + 8 * j), cl);
            unpack8(*reinterpret_cast<const uint4*>(cs + p * 128 + 64 + 8 * j), ch);
            unpack8(*reinterpret_cast<const uint4*>(sn + p * 128 + 8 * j), sl);
            unpack8(*reinterpret_cast<const uint4*>(sn + p * 128 + 64 + 8 * j), sh);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float a = lo[i], b = hi[i];
                lo[i] = rbf(rbf(a * cl[i]) + rbf((-b) * sl[i]));
                hi[i] = rbf(rbf(b * ch[i]) + rbf(a * sh[i]));
            }
        }
        if (act) {
            *reinterpret_cast<uint4*>(base + 8 * j) = pack8(lo);
            *reinterpret_cast<uint4*>(base + 64 + 8 * j) = pack8(hi);
        }
    }
}

extern "C" int drn_qk_norm_rope(void* q, void* k, const void* wq, const void* wk, const void* cos, const void* sin,
                                int64_t tokens, int heads, int64_t ldq, int64_t ldk, int64_t tokens_per_batch,
                                int64_t pos_offset, float eps, void* stream) {
    DRN_CHECK_ARG((q || k) && (!q || wq) && (!k || wk) && tokens >= 0 && heads > 0 && ldq % 8 == 0 && ldk % 8 == 0 &&
                  tokens_per_batch > 0);
    DRN_CHECK_ARG((cos == nullptr) == (sin == nullptr));
    if (tokens == 0) return DRN_OK;
    const int64_t items = tokens * ((heads + 7) / 8) * ((q ? 1 : 0) + (k ? 1 : 0));
    int64_t blocks = (items + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    qk_norm_rope_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(
        (bf16_t*)q, (bf16_t*)k, (const bf16_t*)wq, (const bf16_t*)wk, (const bf16_t*)cos, (const bf16_t*)sin, tokens,
        heads, ldq, ldk, tokens_per_batch, pos_offset, eps);
    return drn_launch_status();
}

// ------------------------------------------------------------------------------------------------
// patchify + concat [x | cond | ones] -> token rows (c r m n), zero K-padding.  CleanGeneralDIT.py:669-675, :409-414.
__global__ __launch_bounds__(256) void patchify_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ cond,
                                                       bf16_t* __restrict__ out, int B, int Cx, int Cc, int with_mask,
                                                       int Tl, int Hl, int Wl, int pt, int ps, int64_t ldo) {
    const int Tp = Tl / pt, Hp = Hl / ps, Wp = Wl / ps;
    const int C = Cx + Cc + (with_mask ? 1 : 0);
    const int kdim = C * pt * ps * ps;
    const int64_t total = (int64_t)B * Tp * Hp * Wp * ldo;
    const int64_t plane = (int64_t)Tl * Hl * Wl;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t tokrow = i / ldo;
        const int col = (int)(i - tokrow * ldo);
        bf16_t val = 0;
        if (col < kdim) {
            int n = col % ps;
            int m = (col / ps) % ps;
            int r = (col / (ps * ps)) % pt;
            int c = col / (ps * ps * pt);
            int64_t t = tokrow;
            const int w = (int)(t % Wp); t /= Wp;
            const int hh = (int)(t % Hp); t /= Hp;
            const int tt = (int)(t % Tp);
            const int b = (int)(t / Tp);
            const int64_t sp = ((int64_t)(tt * pt + r) * Hl + (hh * ps + m)) * Wl + (w * ps + n);
            if (c < Cx) val = x[((int64_t)b * Cx + c) * plane + sp];
            else if (c < Cx + Cc) val = cond[((int64_t)b * Cc + (c - Cx)) * plane + sp];
            else val = 0x3F80;   // 1.0 (padding mask, CleanGeneralDIT.py:672)
        }
        out[i] = val;
    }
}

extern "C" int drn_patchify_concat(const void* x, const void* cond, void* out, int B, int Cx, int Cc, int with_mask,
                                   int Tl, int Hl, int Wl, int pt, int ps, int64_t ldo, void* stream) {
    DRN_CHECK_ARG(x && out && B > 0 && Cx > 0 && Cc >= 0 && (Cc == 0 || cond) && pt > 0 && ps > 0);
    DRN_CHECK_ARG(Tl % pt == 0 && Hl % ps == 0 && Wl % ps == 0);
    DRN_CHECK_ARG(ldo >= (int64_t)(Cx + Cc + (with_mask ? 1 : 0)) * pt * ps * ps);
    const int64_t total = (int64_t)B * (Tl / pt) * (Hl / ps) * (Wl / ps) * ldo;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    patchify_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(
        (const bf16_t*)x, (const bf16_t*)cond, (bf16_t*)out, B, Cx, Cc, with_mask, Tl, Hl, Wl, pt, ps, ldo);
    return drn_launch_status();
}

// unpatchify '(B T)(H W)(ph pw pt C) -> B C (T pt)(H ph)(W pw)', CleanGeneralDIT.py:709-716
__global__ __launch_bounds__(256) void unpatchify_kernel(const bf16_t* __restrict__ y, int64_t ldy,
                                                         bf16_t* __restrict__ out, int B, int C, int Tp, int Hp, int Wp,
                                                         int pt, int ps) {
    const int To = Tp * pt, Ho = Hp * ps, Wo = Wp * ps;
    const int64_t total = (int64_t)B * C * To * Ho * Wo;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t t = i;
        const int wo = (int)(t % Wo); t /= Wo;
        const int ho = (int)(t % Ho); t /= Ho;
        const int to = (int)(t % To); t /= To;
        const int c = (int)(t % C);
        const int b = (int)(t / C);
        const int W = wo / ps, pw = wo % ps, H = ho / ps, ph = ho % ps, T = to / pt, ptt = to % pt;
        const int64_t tokrow = (((int64_t)b * Tp + T) * Hp + H) * Wp + W;
        const int col = ((ph * ps + pw) * pt + ptt) * C + c;
        out[i] = y[tokrow * ldy + col];
    }
}

extern "C" int drn_unpatchify(const void* y, int64_t ldy, void* out, int B, int C, int Tp, int Hp, int Wp, int pt,
                              int ps, void* stream) {
    DRN_CHECK_ARG(y && out && B > 0 && C > 0 && Tp > 0 && Hp > 0 && Wp > 0 && pt > 0 && ps > 0);
    DRN_CHECK_ARG(ldy >= (int64_t)C * pt * ps * ps);
    const int64_t total = (int64_t)B * C * Tp * pt * Hp * ps * Wp * ps;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    unpatchify_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>((const bf16_t*)y, ldy,
                                                                                     (bf16_t*)out, B, C, Tp, Hp, Wp, pt, ps);
    return drn_launch_status();
}

// ------------------------------------------------------------------------------------------------
// EDM Euler sampler, model_diffusion_renderer.py:30-82.  Scalars are computed on the host with the reference's
// own fp32 torch ops; the kernels repeat its unfused fp32 elementwise sequence, so results are bit-exact.
__global__ __launch_bounds__(256) void edm_scale_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ o, int64_t n,
                                                        float c_in) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        o[i] = f2bf(bf2f(x[i]) * c_in);
}
__global__ __launch_bounds__(256) void edm_step_kernel(const bf16_t* __restrict__ mo, const bf16_t* __restrict__ x,
                                                       bf16_t* __restrict__ o, int64_t n, float c_skip, float c_out,
                                                       float sigma, float dt) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float sm = bf2f(x[i]);
        const float m = bf2f(mo[i]);
        const float denoised = c_skip * sm + c_out * m;
        const float deriv = (sm - denoised) / sigma;
        o[i] = f2bf(sm + deriv * dt);
    }
}
// CFG: out = bf16(c + bf16(g * bf16(c - u)))   model_diffusion_renderer.py:232
__global__ __launch_bounds__(256) void cfg_kernel(const bf16_t* __restrict__ c, const bf16_t* __restrict__ u,
                                                  bf16_t* __restrict__ o, int64_t n, float g) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float cc = bf2f(c[i]);
        const float d = rbf(cc - bf2f(u[i]));
        o[i] = f2bf(cc + rbf(g * d));
    }
}
static inline dim3 ew_grid(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return dim3((unsigned)b);
}
extern "C" int drn_edm_scale_input(const void* x, void* out, int64_t n, float c_in, void* stream) {
    DRN_CHECK_ARG(x && out && n >= 0);
    if (n == 0) return DRN_OK;
    edm_scale_kernel<<<ew_grid(n), dim3(256), 0, (hipStream_t)stream>>>((const bf16_t*)x, (bf16_t*)out, n, c_in);
    return drn_launch_status();
}
extern "C" int drn_edm_step(const void* model_out, const void* sample, void* out, int64_t n, float c_skip, float c_out,
                            float sigma, float dt, void* stream) {
    DRN_CHECK_ARG(model_out && sample && out && n >= 0);
    if (n == 0) return DRN_OK;
    edm_step_kernel<<<ew_grid(n), dim3(256), 0, (hipStream_t)stream>>>((const bf16_t*)model_out, (const bf16_t*)sample,
                                                                       (bf16_t*)out, n, c_skip, c_out, sigma, dt);
    return drn_launch_status();
}
extern "C" int drn_cfg_combine(const void* cond, const void* uncond, void* out, int64_t n, float guidance, void* stream) {
    DRN_CHECK_ARG(cond && uncond && out && n >= 0);
    if (n == 0) return DRN_OK;
    cfg_kernel<<<ew_grid(n), dim3(256), 0, (hipStream_t)stream>>>((const bf16_t*)cond, (const bf16_t*)uncond,
                                                                  (bf16_t*)out, n, guidance);
    return drn_launch_status();
}

// ------------------------------------------------------------------------------------------------
// pipeline post-process -> uint8 (B,T,H,W,3); diffusion_renderer_pipeline.py:299-318 with every bf16 rounding kept.
__global__ __launch_bounds__(256) void postprocess_kernel(const bf16_t* __restrict__ v, uint8_t* __restrict__ o, int B,
                                                          int64_t thw, int normalize) {
    const int64_t total = (int64_t)B * thw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / thw, p = i - b * thw;
        float c[3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) c[ch] = bf2f(v[(b * 3 + ch) * thw + p]);
        if (normalize) {
            const float norm = rbf(sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]));
            const float den = rbf(fmaxf(norm, 1e-12f));
            // torch CPU rounds the Python scalar of a bf16 add/sub to bf16 first (0.2 -> 0.2001953125); div keeps fp32
            float blend = rbf(rbf(norm - 0.2001953125f) / 0.2f);
            blend = fminf(fmaxf(blend, 0.f), 1.f);
            const float om = rbf(1.0f - blend);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float vn = rbf(c[ch] / den);
                c[ch] = rbf(rbf(vn * blend) + rbf(c[ch] * om));
            }
        }
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            float a = rbf(1.0f + c[ch]);
            a = fminf(fmaxf(a, 0.f), 2.f);
            const float u = rbf((a / 2.0f) * 255.0f);
            o[i * 3 + ch] = (uint8_t)u;
        }
    }
}

extern "C" int drn_postprocess_u8(const void* video, void* out_u8, int B, int T, int H, int W, int normalize_normal,
                                  void* stream) {
    DRN_CHECK_ARG(video && out_u8 && B > 0 && T > 0 && H > 0 && W > 0);
    const int64_t thw = (int64_t)T * H * W;
    postprocess_kernel<<<ew_grid((int64_t)B * thw), dim3(256), 0, (hipStream_t)stream>>>(
        (const bf16_t*)video, (uint8_t*)out_u8, B, thw, normalize_normal);
    return drn_launch_status();
}

extern "C" int drn_abi_version(void) { return DRN_ABI_VERSION; }

extern "C" const char* drn_error_string(int code) {
    if (code == DRN_OK) return "ok";
    if (code == DRN_EINVAL) return "drn: unsupported shape, stride or alignment (nothing launched)";
    return hipGetErrorString((hipError_t)code);
}
