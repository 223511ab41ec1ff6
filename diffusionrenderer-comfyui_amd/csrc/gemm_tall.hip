// Few-token linears (S = 256 .. 1024 rows: BASELINE configs 1 and 2): C = epi(A[M,K] . W[N,K]^T) with M a multiple of 256.
//
// At M = 256 every weight byte is used for 256 FLOP: the product is a weight stream that also has to be multiplied.  What the
// round-2 measurements say about the split-K paths (gemm.hip 128^2 tiles, gemm256s.hip 256^2 tiles; QKV 100 MB of weights):
//   * ~45 us per launch + ~9-13 us for the reduce kernel whatever the prefetch depth (2 + 2 stages, 2 + 3 stages with W three K
//     steps ahead) and whatever the weight layout (row-major, K-step-major): neither DRAM latency nor page locality is the bound;
//   * the fp32 slices are: 4 slices of a [256, 12288] product = 50 MB written at the end of the GEMM launch and 50 MB read back by
//     the reduce launch - as many bytes as the weights themselves, moved while no MFMA runs.
// This kernel removes the slices where the shape allows: a workgroup owns ALL 256 rows of a clip x 64 output columns over the
// WHOLE K, so N / 64 workgroups (QKV 192, MLP-up 256) fill the chip without a K split and the epilogue (bias-free linear,
// GELU, gated residual) is fused again.  Cost: every workgroup takes in the whole A panel (256 x K x 2 B = 2 MB at K = 4096)
// through its CU's LDS-DMA path (L2 hits; ~70-80 GB/s per CU -> ~30 us), which is why the tile is as TALL as the clip and only
// 64 wide: (256 + 64) x 128 B = 40 KiB per K step, four stages = the whole 160 KiB of LDS, W and A three K steps ahead.
// Products with few output columns (out-proj, MLP-down: N = 4096 -> 64 tiles) still split K, by gridDim.y, into fp32 slices
// (4 slices x 4 MB = a third of the bytes the 256-wide tiles needed).
//
// 8 waves = 4 (rows) x 2 (columns): a wave owns 64 rows x 32 columns = 4 x 2 MFMA tiles (16x16x32 bf16, D = Wfrag x Afrag: a
// lane holds 4 consecutive columns of one row), 12 ds_read_b128 + 16 MFMAs per K step.  One barrier per K step: stage k is
// waited for with a counted vmcnt (the two younger stages stay in flight), the barrier publishes it and frees stage k - 1,
// which the DMA of K step k + 3 then refills.  128-byte LDS rows, chunk ^ ((row >> 1) & 7) swizzle on the DMA source address.
//
// Round 3: the tile is a template parameter.  Shape 1 = 128 rows x 128 columns (two row tiles per clip; 8 waves = 2 x 4 of the
// same 64 x 32 wave tile, same MFMA order per output element: same bits): a workgroup takes in (128 + 128) x 128 B = 32 KiB per K
// step instead of 40, five stages fit the LDS (operands four K steps ahead), and the two row tiles of a clip that share a W slab
// are N / 128 apart in dispatch order - a multiple of 8, i.e. the same XCD and L2 - so the weights still leave HBM once.
// Measured at S = 256 (tools/kbench.py gemm --S 256 --cold 6 --splitk 1, both tiles in one process): QKV 51 -> 45 us, MLP-up
// 56 -> 51 us.  With 32 pieces of 1 KiB per K step in ~1400 cycles the kernel sits at the CU's LDS-DMA piece rate (one per ~37
// cycles): the bytes a workgroup takes in per output are what is left to cut, and (128 + 128) is the minimum of
// TM + TN at TM x TN = 16 384 outputs per workgroup.
#include <stdlib.h>
#include "drn_common.h"

#define BK 64
#define EPI_PARTIAL 3                      // internal: fp32 slice [blockIdx.y][M][N] to the workspace
#ifndef TALL_SHAPE_DEFAULT
#define TALL_SHAPE_DEFAULT 1               // 0: 256 x 64 (4 stages), 1: 128 x 128 (5 stages); DRN_GEMM_TALL_SHAPE overrides
#endif

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int EPI, int TM, int TN, int NSTAGE>
__global__ __launch_bounds__(512, 2) void gemm_tall_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, void* Cv,
                                                           int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                                                           int64_t ldc, const bf16_t* __restrict__ gate, const bf16_t* R,
                                                           int64_t ldr, int64_t rpb) {
    constexpr int A_BYTES = TM * BK * 2, W_BYTES = TN * BK * 2, STAGE_BYTES = A_BYTES + W_BYTES;
    constexpr int WN = TN / 32;                // waves across the columns (8 / WN down the rows), wave tile 64 x 32
    constexpr int PA = TM / 64, PW = TN / 64;  // 1 KiB pieces (8 rows x 128 B) of A / W per wave and K step
    static_assert((TM / 64) * (TN / 32) == 8 && PA >= 1 && PW >= 1, "8 waves of 64 x 32");
    extern __shared__ __attribute__((aligned(1024))) char smem[];     // NSTAGE * STAGE_BYTES, the ONLY LDS object
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // tile: all column tiles of one row tile are neighbours in dispatch order - they share the A panel in L2
    const int tiles_n = (int)(N / TN);
    const int tm = (int)(blockIdx.x / tiles_n), tn = (int)(blockIdx.x % tiles_n);
    const int64_t m0 = (int64_t)tm * TM, n0 = (int64_t)tn * TN;
    if (EPI == EPI_PARTIAL) {
        K /= gridDim.y;
        A += (int64_t)blockIdx.y * K;
        W += (int64_t)blockIdx.y * K;
    }
    const int nk = (int)(K / BK);

    // ---- DMA sources: this wave's PA pieces of A (rows 8 PA w ..) and PW pieces of W (rows 8 PW w ..)
    const char* a_base = reinterpret_cast<const char*>(A + (m0 + wave * (8 * PA)) * lda);
    const char* w_base = reinterpret_cast<const char*>(W + (n0 + wave * (8 * PW)) * ldw);
    uint32_t voffa[PA], voffw[PW];
    {
        const int rl = lane >> 3;                                   // row inside a piece
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int r = wave * (8 * PA) + p * 8 + rl;             // row inside the tile (the swizzle needs the full row)
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            voffa[p] = (uint32_t)(((p * 8 + rl) * lda + c * 8) * 2);
        }
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            const int r = wave * (8 * PW) + p * 8 + rl;
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            voffw[p] = (uint32_t)(((p * 8 + rl) * ldw + c * 8) * 2);
        }
    }
    const int dma_a = wave * (PA * 1024), dma_w = A_BYTES + wave * (PW * 1024);
#define STAGE(KT, S)                                                                                                   \
    do {                                                                                                               \
        char* sa_ = smem + (S) * STAGE_BYTES;                                                                          \
        const int64_t kb_ = (int64_t)(KT) * (BK * 2);                                                                  \
        _Pragma("unroll") for (int p = 0; p < PA; ++p)                                                                 \
            __builtin_amdgcn_global_load_lds((gptr_t)(a_base + kb_ + voffa[p]), (lptr_t)(sa_ + dma_a + p * 1024), 16, 0, 0); \
        _Pragma("unroll") for (int p = 0; p < PW; ++p)                                                                 \
            __builtin_amdgcn_global_load_lds((gptr_t)(w_base + kb_ + voffw[p]), (lptr_t)(sa_ + dma_w + p * 1024), 16, 0, 0); \
    } while (0)

    // ---- fragment read offsets inside a stage
    const int fr = lane & 15, fq = lane >> 4;
    int offa[4][2], offw[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int c = ks * 4 + fq;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int r = wm * 64 + mt * 16 + fr;
            offa[mt][ks] = r * 128 + ((c ^ ((r >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int r = wn * 32 + nt * 16 + fr;
            offw[nt][ks] = A_BYTES + r * 128 + ((c ^ ((r >> 1) & 7)) << 4);
        }
    }

    f32x4_t acc[4][2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // ---- prologue: K steps 0 .. NSTAGE - 2 (past the end the last step is re-requested into a dead stage: uniform wait counts)
#pragma unroll
    for (int i = 0; i < NSTAGE - 1; ++i) STAGE(min(i, nk - 1), i);
    int scur = 0, sreq = NSTAGE - 1;                                  // stage of K step kt; stage that step kt + NSTAGE - 1 refills
    for (int kt = 0; kt < nk; ++kt) {
        // own pieces of stage kt have landed (the NSTAGE - 2 younger stages x (PA + PW) pieces may fly)
        asm volatile("s_waitcnt vmcnt(%0)" :: "i"((NSTAGE - 2) * (PA + PW)) : "memory");
        __builtin_amdgcn_s_barrier();                                 // stage kt visible; every wave is done reading stage kt - 1
        __builtin_amdgcn_sched_barrier(0);
        STAGE(min(kt + NSTAGE - 1, nk - 1), sreq);                    // refill the stage of step kt - 1
        const char* st = smem + scur * STAGE_BYTES;
        scur = scur == NSTAGE - 1 ? 0 : scur + 1;
        sreq = sreq == NSTAGE - 1 ? 0 : sreq + 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t af[4], wf[2];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) af[mt] = *reinterpret_cast<const bf16x8_t*>(st + offa[mt][ks]);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) wf[nt] = *reinterpret_cast<const bf16x8_t*>(st + offw[nt][ks]);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[mt], acc[mt][nt], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the re-requests: nothing may land after exit

    // ---- epilogue: a lane holds columns n .. n + 3 of row m
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int64_t m = m0 + wm * 64 + mt * 16 + fr;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int64_t n = n0 + wn * 32 + nt * 16 + fq * 4;
            if (EPI == EPI_PARTIAL) {
                float* part = reinterpret_cast<float*>(Cv) + (int64_t)blockIdx.y * M * N;
                *reinterpret_cast<f32x4_t*>(part + m * N + n) = acc[mt][nt];
                continue;
            }
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = rbf(acc[mt][nt][r]);
            if (EPI == DRN_EPI_GELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_erf_fast(v[r]);
            } else if (EPI == DRN_EPI_GATE_RES) {
                const int64_t b = (int64_t)((uint32_t)m / (uint32_t)rpb);
                const uint2 g2 = *reinterpret_cast<const uint2*>(gate + b * N + n);
                const uint2 r2 = *reinterpret_cast<const uint2*>(R + m * ldr + n);
                const float g[4] = {bflo(g2.x), bfhi(g2.x), bflo(g2.y), bfhi(g2.y)};
                const float x[4] = {bflo(r2.x), bfhi(r2.x), bflo(r2.y), bfhi(r2.y)};
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = x[r] + rbf(g[r] * v[r]);
            }
            uint2 o;
            o.x = pack_bf2(v[0], v[1]);
            o.y = pack_bf2(v[2], v[3]);
            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Cv) + m * ldc + n) = o;
        }
    }
}

template <int EPI, int TM, int TN, int NSTAGE>
static int launch_tall_shape(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                             int64_t ldc, const void* gate, const void* residual, int64_t ldr, int64_t rpb, int splits,
                             hipStream_t st) {
    constexpr int LDS = NSTAGE * (TM + TN) * BK * 2;
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tall_kernel<EPI, TM, TN, NSTAGE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    const int64_t tiles = (M / TM) * (N / TN);
    gemm_tall_kernel<EPI, TM, TN, NSTAGE><<<dim3((unsigned)tiles, (unsigned)splits), dim3(512), LDS, st>>>(
        (const bf16_t*)A, (const bf16_t*)W, C, M, N, K, lda, ldw, ldc, (const bf16_t*)gate, (const bf16_t*)residual, ldr, rpb);
    return drn_launch_status();
}

static int g_tall_shape = -1;          // -1: DRN_GEMM_TALL_SHAPE or the built-in default; drn_gemm_tall_force_shape (tests, A/B)
extern "C" int drn_gemm_tall_force_shape(int shape) {
    const int was = g_tall_shape;
    g_tall_shape = shape < 0 ? -1 : (shape ? 1 : 0);
    return was;
}
static int tall_shape(int64_t N) {
    static int env = -1;
    if (env < 0) {
        const char* e = getenv("DRN_GEMM_TALL_SHAPE");
        env = e ? (e[0] == '1' ? 1 : 0) : TALL_SHAPE_DEFAULT;
    }
    const int want = g_tall_shape >= 0 ? g_tall_shape : env;
    return (want == 1 && N % 128 == 0) ? 1 : 0;
}

template <int EPI>
static int launch_tall(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                       int64_t ldc, const void* gate, const void* residual, int64_t ldr, int64_t rpb, int splits, hipStream_t st) {
    if (tall_shape(N)) return launch_tall_shape<EPI, 128, 128, 5>(A, W, C, M, N, K, lda, ldw, ldc, gate, residual, ldr, rpb, splits, st);
    return launch_tall_shape<EPI, 256, 64, 4>(A, W, C, M, N, K, lda, ldw, ldc, gate, residual, ldr, rpb, splits, st);
}

// Called from gemm.hip.  splits == 1: C = epi(...) directly; splits > 1: fp32 slices into `partial` ([splits][M][N]), the
// caller runs gemm_splitk_epilogue_kernel afterwards.  Requirements (checked by the caller's rule, re-checked here): M % 256 == 0,
// N % 64 == 0, (K / 64) % splits == 0, operand byte offsets of a tile row panel < 4 GiB, 16-byte aligned pointers / strides.
int drn_gemm_tall_dispatch(const void* A, const void* W, void* C, float* partial, int64_t M, int64_t N, int64_t K, int64_t lda,
                           int64_t ldw, int64_t ldc, int epilogue, const void* gate, const void* residual, int64_t ldr,
                           int64_t rpb, int splits, void* stream) {
    if (M <= 0 || M % 256 != 0 || N % 64 != 0 || K % BK != 0 || splits < 1 || (K / BK) % splits != 0) return DRN_EINVAL;
    if (M >= (1ll << 31) || (M / 128) * (N / 64) >= (1ll << 31) || splits > 65535) return DRN_EINVAL;
    if (32 * lda * 2 + K * 2 >= (1ll << 32) || 16 * ldw * 2 + K * 2 >= (1ll << 32)) return DRN_EINVAL;   // 32-bit lane offsets
    if (rpb <= 0 || rpb > M) rpb = M;
    hipStream_t st = (hipStream_t)stream;
    if (splits > 1)
        return launch_tall<EPI_PARTIAL>(A, W, partial, M, N, K, lda, ldw, N, nullptr, nullptr, 0, M, splits, st);
    switch (epilogue) {
        case DRN_EPI_NONE: return launch_tall<DRN_EPI_NONE>(A, W, C, M, N, K, lda, ldw, ldc, gate, residual, ldr, rpb, 1, st);
        case DRN_EPI_GELU: return launch_tall<DRN_EPI_GELU>(A, W, C, M, N, K, lda, ldw, ldc, gate, residual, ldr, rpb, 1, st);
        case DRN_EPI_GATE_RES: return launch_tall<DRN_EPI_GATE_RES>(A, W, C, M, N, K, lda, ldw, ldc, gate, residual, ldr, rpb, 1, st);
        default: return DRN_EINVAL;
    }
}
