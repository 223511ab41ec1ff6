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
#include <stdlib.h>
#include "drn_common.h"

#define TM 256
#define TN 64
#define BK 64
#define A_BYTES (TM * BK * 2)              // 32 KiB
#define W_BYTES (TN * BK * 2)              // 8 KiB
#define STAGE_BYTES (A_BYTES + W_BYTES)    // 40 KiB
#define NSTAGE 4
#define EPI_PARTIAL 3                      // internal: fp32 slice [blockIdx.y][M][N] to the workspace

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_tall_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W, void* Cv,
                                                           int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                                                           int64_t ldc, const bf16_t* __restrict__ gate, const bf16_t* R,
                                                           int64_t ldr, int64_t rpb) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];     // NSTAGE * STAGE_BYTES, the ONLY LDS object
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // tile: all column tiles of one clip (row tile) are neighbours in dispatch order - they share the A panel in L2
    const int tiles_n = (int)(N / TN);
    const int tm = (int)(blockIdx.x / tiles_n), tn = (int)(blockIdx.x % tiles_n);
    const int64_t m0 = (int64_t)tm * TM, n0 = (int64_t)tn * TN;
    if (EPI == EPI_PARTIAL) {
        K /= gridDim.y;
        A += (int64_t)blockIdx.y * K;
        W += (int64_t)blockIdx.y * K;
    }
    const int nk = (int)(K / BK);

    // ---- DMA sources: this wave's 4 pieces of A (rows 32 w .. 32 w + 31) and 1 piece of W (rows 8 w .. 8 w + 7)
    const char* a_base = reinterpret_cast<const char*>(A + (m0 + wave * 32) * lda);
    const char* w_base = reinterpret_cast<const char*>(W + (n0 + wave * 8) * ldw);
    uint32_t voffa[4], voffw;
    {
        const int rl = lane >> 3;                                   // row inside a piece
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = p * 8 + rl;                               // row inside this wave's 32 A rows (32 w is a multiple of 16)
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            voffa[p] = (uint32_t)((r * lda + c * 8) * 2);
        }
        const int rw = wave * 8 + rl;                               // W row inside the tile: the swizzle needs the full row
        const int cw = (lane & 7) ^ ((rw >> 1) & 7);
        voffw = (uint32_t)((rl * ldw + cw * 8) * 2);
    }
    const int dma_a = wave * 4096, dma_w = A_BYTES + wave * 1024;
#define STAGE(KT, S)                                                                                                   \
    do {                                                                                                               \
        char* sa_ = smem + (S) * STAGE_BYTES;                                                                          \
        const int64_t kb_ = (int64_t)(KT) * (BK * 2);                                                                  \
        _Pragma("unroll") for (int p = 0; p < 4; ++p)                                                                  \
            __builtin_amdgcn_global_load_lds((gptr_t)(a_base + kb_ + voffa[p]), (lptr_t)(sa_ + dma_a + p * 1024), 16, 0, 0); \
        __builtin_amdgcn_global_load_lds((gptr_t)(w_base + kb_ + voffw), (lptr_t)(sa_ + dma_w), 16, 0, 0);            \
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

    // ---- prologue: K steps 0, 1, 2 (past the end the last step is re-requested into a dead stage: uniform wait counts)
    STAGE(0, 0);
    STAGE(min(1, nk - 1), 1);
    STAGE(min(2, nk - 1), 2);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");            // own pieces of stage kt (2 younger stages x 5 may fly)
        __builtin_amdgcn_s_barrier();                                 // stage kt visible; every wave is done reading stage kt - 1
        __builtin_amdgcn_sched_barrier(0);
        STAGE(min(kt + 3, nk - 1), (kt + 3) & 3);                     // refill stage kt - 1
        const char* st = smem + (kt & 3) * STAGE_BYTES;
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

template <int EPI>
static int launch_tall(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                       int64_t ldc, const void* gate, const void* residual, int64_t ldr, int64_t rpb, int splits, hipStream_t st) {
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tall_kernel<EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * STAGE_BYTES);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    const int64_t tiles = (M / TM) * (N / TN);
    gemm_tall_kernel<EPI><<<dim3((unsigned)tiles, (unsigned)splits), dim3(512), NSTAGE * STAGE_BYTES, st>>>(
        (const bf16_t*)A, (const bf16_t*)W, C, M, N, K, lda, ldw, ldc, (const bf16_t*)gate, (const bf16_t*)residual, ldr, rpb);
    return drn_launch_status();
}

// Called from gemm.hip.  splits == 1: C = epi(...) directly; splits > 1: fp32 slices into `partial` ([splits][M][N]), the
// caller runs gemm_splitk_epilogue_kernel afterwards.  Requirements (checked by the caller's rule, re-checked here): M % 256 == 0,
// N % 64 == 0, (K / 64) % splits == 0, operand byte offsets of a tile row panel < 4 GiB, 16-byte aligned pointers / strides.
int drn_gemm_tall_dispatch(const void* A, const void* W, void* C, float* partial, int64_t M, int64_t N, int64_t K, int64_t lda,
                           int64_t ldw, int64_t ldc, int epilogue, const void* gate, const void* residual, int64_t ldr,
                           int64_t rpb, int splits, void* stream) {
    if (M <= 0 || M % TM != 0 || N % TN != 0 || K % BK != 0 || splits < 1 || (K / BK) % splits != 0) return DRN_EINVAL;
    if (M >= (1ll << 31) || (M / TM) * (N / TN) >= (1ll << 31) || splits > 65535) return DRN_EINVAL;
    if (32 * lda * 2 + K * 2 >= (1ll << 32) || 8 * ldw * 2 + K * 2 >= (1ll << 32)) return DRN_EINVAL;   // 32-bit lane offsets
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
