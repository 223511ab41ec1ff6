// The streamed 256x256x64 MFMA main loop of gemm256s.hip as a set of macros over fixed local names (smem, lane, wave, nk,
// offa, offw, acc, af, wx, wy, dma_off + whatever the DMA macro in force reads), shared by the dense GEMM kernels
// (gemm256s.hip) and the implicit-GEMM convolution (conv256s.hip), which differ only in where a half-tile's rows come from
// (the DMA macro) and in the epilogue.  Schedule, wait-count arithmetic and LDS layout: see the header of gemm256s.hip.
#pragma once
#include "drn_common.h"

// (G256S_ABL, timing-only ablations with WRONG results: 16 = no counted DMA wait, 32 = no barrier, 256 = half of the fragment reads, 512 = cached operands)
#ifndef G256S_ABL
#define G256S_ABL 0
#endif
#define TB 256
#define BK 64
#define HALF_BYTES (128 * BK * 2)          // 16 KiB
#define STAGE_BYTES (4 * HALF_BYTES)       // A0 A1 W0 W1
// half-tile ids
#define H_A0 0
#define H_A1 1
#define H_W0 2
#define H_W1 3

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// workgroup index (dispatch order; XCD = index & 7) -> tile: XCD-contiguous chunks, GROUP tile rows per L2 band
static __device__ __forceinline__ void tile_of(int bid, int nwg, int tiles_m, int tiles_n, int GROUP, int64_t& m0, int64_t& n0) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int width = GROUP * tiles_n;
    const int group_id = pid / width;
    const int first_m = group_id * GROUP;
    const int gsz = min(tiles_m - first_m, GROUP);
    m0 = (int64_t)(first_m + (pid % width) % gsz) * TB;
    n0 = (int64_t)((pid % width) / gsz) * TB;
}

// this wave's 2 pieces (16 rows) of each half-tile [A0 A1 W0 W1] of the tile at (m0, n0)
#define SET_SRC(M0, N0)                                                                                                \
    do {                                                                                                               \
        _Pragma("unroll") for (int h = 0; h < 4; ++h)                                                                  \
            _Pragma("unroll") for (int p = 0; p < 2; ++p) {                                                            \
                const int r = (wave * 2 + p) * 8 + (lane >> 3);          /* row inside the half-tile, 0..127 */        \
                const int c = (lane & 7) ^ ((r >> 1) & 7);                                                             \
                if (h < 2) {                                                                                           \
                    int64_t row = (M0) + h * 128 + r;                                                                  \
                    if (row > M - 1) row = M - 1;                                                                      \
                    gsrc[h][p] = A + row * lda + c * 8;                                                                \
                } else {                                                                                               \
                    int64_t row = (N0) + (h - 2) * 128 + r;                                                            \
                    if (row > N - 1) row = N - 1;                                                                      \
                    gsrc[h][p] = W + row * ldw + c * 8;                                                                \
                }                                                                                                      \
            }                                                                                                          \
    } while (0)

// LDS regions.  Default: 2 stages x [A0 A1 W0 W1] x 16 KiB.  A kernel may redefine A_OFF / W_OFF before its body (gemm256s.hip's
// weight-streaming kernel keeps 2 A stages and 3 W stages: the cold operand gets the deeper ring).
#define A_OFF(S, I) ((S) * STAGE_BYTES + (I) * HALF_BYTES)
#define W_OFF(S, J) ((S) * STAGE_BYTES + (2 + (J)) * HALF_BYTES)
#define H_OFF(S, H) ((H) < 2 ? A_OFF(S, (H) & 1) : W_OFF(S, ((H) - 2) & 1))
// blocked operand layouts (drn_gemm_bf16_blocked), see gemm256s.hip
#define A_KOFF(KT) ((((int64_t)(KT) * BK) >> abc) * abs_ + (((int64_t)(KT) * BK) & ((1ll << abc) - 1)))
// half-tile H of K step KD (of the tile gsrc points at) into stage S
#define DMA(H, KD, S) GEMM_DMA(H, KD, S)
#define GEMM_DMA(H, KD, S)                                                                                             \
    do {                                                                                                               \
        const int kt_ = (G256S_ABL & 512) ? 0 : (int)(KD);   /* (512, timing only: every request re-reads K step 0 - L2 hits) */ \
        (void)0;                                                                                     \
        char* dst_ = smem + H_OFF(S, H) + dma_off;                                                                     \
        const int64_t ko_ = (H) < 2 ? A_KOFF(kt_) : (int64_t)kt_ * BK;                                                 \
        __builtin_amdgcn_global_load_lds((gptr_t)(gsrc[H][0] + ko_), (lptr_t)dst_, 16, 0, 0);                          \
        __builtin_amdgcn_global_load_lds((gptr_t)(gsrc[H][1] + ko_), (lptr_t)(dst_ + 1024), 16, 0, 0);                 \
    } while (0)

#define LD_A_(S, I, MT, KS) (*reinterpret_cast<const bf16x8_t*>(smem + A_OFF(S, I) + ((KS) ? (offa ^ 64) : offa) + (MT) * 2048))
#define LD_W_(S, J, NT, KS) (*reinterpret_cast<const bf16x8_t*>(smem + W_OFF(S, J) + ((KS) ? (offw ^ 64) : offw) + (NT) * 2048))
// (G256S_ABL & 256, timing only: every second A fragment and every second W fragment is NOT read - a third of the LDS read bytes gone,
//  everything else unchanged: what the fragment reads cost in time at the chip's power limit)
#define LD_A(S, I, MT, KS) (((G256S_ABL & 256) && ((MT) & 1)) ? af[(MT) & 2][KS] : LD_A_(S, I, MT, KS))
#define LD_W(S, J, NT, KS) (((G256S_ABL & 256) && ((NT) & 1)) ? wx[0][KS] : LD_W_(S, J, NT, KS))

#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define MM(I, MT, J, NT, KS, WF)                                                                                    \
    acc[I][MT][J][NT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[NT][KS], af[MT][KS], acc[I][MT][J][NT], 0, 0, 0)
// end of a group: the half-tile whose first read follows has landed (own pieces: 5 half-tiles = 10 younger pieces may fly;
// VM = 10, or 10 + the stores of an epilogue that were issued after that half-tile's request)
#define HANDOVER(VM)                                                                                                \
    do {                                                                                                            \
        FENCE();                                                                                                    \
        if (!(G256S_ABL & 16)) asm volatile("s_waitcnt vmcnt(" #VM ")" ::: "memory");                               \
        if (!(G256S_ABL & 32)) __builtin_amdgcn_s_barrier();                                                        \
        FENCE();                                                                                                    \
    } while (0)
// one (mt, ks) slot of a group: the two MFMAs (nt = 0, 1) that read af[MT][KS], then that slot's share of the prefetch
#define SLOT(I, J, MT, KS, WF, PREFETCH)                                                                            \
    do {                                                                                                            \
        MM(I, MT, J, 0, KS, WF);                                                                                    \
        MM(I, MT, J, 1, KS, WF);                                                                                    \
        PREFETCH;                                                                                                   \
        FENCE();                                                                                                    \
    } while (0)
// K step with its A half-tiles in A stage SA and its W half-tiles in W stage SW (literals; SAN / SWN = the stages of the NEXT
// step), W0 of this step in WA: groups 1..4; the DMA of a later step (k indices KDA / KDW of the tile gsrc points at) goes into
// the regions they free.  On exit af = A0 and WB = W0 of the next step.  V1..V4: the hand-over waits.
#define KSTEP2(SA, SAN, SW, SWN, WA, WB, KDA, KDW, V1, V2, V3, V4)                                                  \
    do {                                                                                                            \
        /* group 1 (A0,W0): request W1 -> WB; DMA W0(+2) */                                                         \
        SLOT(0, 0, 0, 0, WA, WB[0][0] = LD_W(SW, 1, 0, 0));                                                          \
        SLOT(0, 0, 1, 0, WA, WB[1][0] = LD_W(SW, 1, 1, 0));                                                          \
        SLOT(0, 0, 2, 0, WA, WB[0][1] = LD_W(SW, 1, 0, 1));                                                          \
        SLOT(0, 0, 3, 0, WA, WB[1][1] = LD_W(SW, 1, 1, 1));                                                          \
        DMA(H_W0, KDW, SW);                                                                                           \
        FENCE();                                                                                                    \
        SLOT(0, 0, 0, 1, WA, (void)0); SLOT(0, 0, 1, 1, WA, (void)0); SLOT(0, 0, 2, 1, WA, (void)0); SLOT(0, 0, 3, 1, WA, (void)0); \
        HANDOVER(V1);                    /* A1 of this step has landed */                                           \
        /* group 2 (A0,W1): request A1 -> af as its A0 entries die; DMA A0(+2) */                                   \
        SLOT(0, 1, 0, 0, WB, af[0][0] = LD_A(SA, 1, 0, 0));                                                          \
        SLOT(0, 1, 1, 0, WB, af[1][0] = LD_A(SA, 1, 1, 0));                                                          \
        SLOT(0, 1, 2, 0, WB, af[2][0] = LD_A(SA, 1, 2, 0));                                                          \
        SLOT(0, 1, 3, 0, WB, af[3][0] = LD_A(SA, 1, 3, 0));                                                          \
        DMA(H_A0, KDA, SA);                                                                                           \
        FENCE();                                                                                                    \
        SLOT(0, 1, 0, 1, WB, af[0][1] = LD_A(SA, 1, 0, 1));                                                          \
        SLOT(0, 1, 1, 1, WB, af[1][1] = LD_A(SA, 1, 1, 1));                                                          \
        SLOT(0, 1, 2, 1, WB, af[2][1] = LD_A(SA, 1, 2, 1));                                                          \
        SLOT(0, 1, 3, 1, WB, af[3][1] = LD_A(SA, 1, 3, 1));                                                          \
        HANDOVER(V2);                    /* W0 of the next step has landed */                                       \
        /* group 3 (A1,W1): request W0(+1) -> WB as it dies; DMA W1(+2) */                                          \
        SLOT(1, 1, 0, 0, WB, (void)0); SLOT(1, 1, 1, 0, WB, (void)0); SLOT(1, 1, 2, 0, WB, (void)0);                \
        SLOT(1, 1, 3, 0, WB, (WB[0][0] = LD_W(SWN, 0, 0, 0), WB[1][0] = LD_W(SWN, 0, 1, 0)));               \
        DMA(H_W1, KDW, SW);                                                                                           \
        FENCE();                                                                                                    \
        SLOT(1, 1, 0, 1, WB, (void)0); SLOT(1, 1, 1, 1, WB, (void)0); SLOT(1, 1, 2, 1, WB, (void)0);                \
        SLOT(1, 1, 3, 1, WB, (WB[0][1] = LD_W(SWN, 0, 0, 1), WB[1][1] = LD_W(SWN, 0, 1, 1)));               \
        HANDOVER(V3);                    /* A0 of the next step has landed */                                       \
        /* group 4 (A1,W0): request A0(+1) -> af as its A1 entries die; DMA A1(+2) */                               \
        SLOT(1, 0, 0, 0, WA, af[0][0] = LD_A(SAN, 0, 0, 0));                                                    \
        SLOT(1, 0, 1, 0, WA, af[1][0] = LD_A(SAN, 0, 1, 0));                                                    \
        SLOT(1, 0, 2, 0, WA, af[2][0] = LD_A(SAN, 0, 2, 0));                                                    \
        SLOT(1, 0, 3, 0, WA, af[3][0] = LD_A(SAN, 0, 3, 0));                                                    \
        DMA(H_A1, KDA, SA);                                                                                           \
        FENCE();                                                                                                    \
        SLOT(1, 0, 0, 1, WA, af[0][1] = LD_A(SAN, 0, 0, 1));                                                    \
        SLOT(1, 0, 1, 1, WA, af[1][1] = LD_A(SAN, 0, 1, 1));                                                    \
        SLOT(1, 0, 2, 1, WA, af[2][1] = LD_A(SAN, 0, 2, 1));                                                    \
        SLOT(1, 0, 3, 1, WA, af[3][1] = LD_A(SAN, 0, 3, 1));                                                    \
        HANDOVER(V4);                    /* W1 of the next step has landed */                                       \
    } while (0)

// the two-stage form: A and W of a step share stage S, the step two ahead is requested
#define KSTEP(S, WA, WB, KD, V1, V2, V3, V4) KSTEP2(S, (S) ^ 1, S, (S) ^ 1, WA, WB, KD, KD, V1, V2, V3, V4)

// (hook for a DMA macro that keeps a cursor instead of using the K index it is given: conv256s.hip)
#ifndef BETWEEN_PROLOGUE_STEPS
#define BETWEEN_PROLOGUE_STEPS() (void)0
#endif
// prologue of a cold start: K steps 0 and 1 requested in the steady-state order; W0(0) / A0(0) go to registers, W1(0) visible
#define PROLOGUE()                                                                                                  \
    do {                                                                                                            \
        DMA(H_W0, 0, 0); DMA(H_A0, 0, 0); DMA(H_W1, 0, 0); DMA(H_A1, 0, 0);                                         \
        BETWEEN_PROLOGUE_STEPS();                                                                                   \
        DMA(H_W0, k_second, 1); DMA(H_A0, k_second, 1); DMA(H_W1, k_second, 1); DMA(H_A1, k_second, 1);             \
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                  /* W0(0), A0(0) */                       \
        __builtin_amdgcn_s_barrier();                                                                               \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                            \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) wx[nt][ks] = LD_W(0, 0, nt, ks);                       \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                            \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) af[mt][ks] = LD_A(0, 0, mt, ks);                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                          \
        FENCE();                                                                                                    \
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                  /* W1(0): read in group 1 of K step 0 */ \
        __builtin_amdgcn_s_barrier();              /* (and every wave holds W0(0) / A0(0): their regions are free) */ \
        FENCE();                                                                                                    \
    } while (0)

#define ZERO_ACC()                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                   \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                            \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                           \
                _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) acc[i][mt][j][nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}

// per-thread constants of both kernels
#define THREAD_SETUP()                                                                                              \
    extern __shared__ __attribute__((aligned(1024))) char smem[];     /* 2 * STAGE_BYTES, the ONLY LDS object */     \
    const int tid = threadIdx.x;                                                                                    \
    const int lane = tid & 63;                                                                                      \
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                                                      \
    const int wr = (wave >> 1) & 1;            /* which 64 rows of each A half */                                   \
    const int wc = (wave & 1) | ((wave >> 2) << 1);    /* which 32 rows (output columns) of each W half: 0..3 */     \
    const int tiles_m = (int)((M + TB - 1) / TB);                                                                   \
    const int tiles_n = (int)((N + TB - 1) / TB);                                                                   \
    const int nwg = tiles_m * tiles_n;                                                                              \
    const int dma_off = wave * 2048;                                   /* this wave's 2 KiB inside a half-tile region */ \
    const int nk = (int)(K / BK);                                                                                   \
    /* fragment read offsets inside a half-tile region (k-substep 1 = offset ^ 64) */                               \
    const int fr = lane & 15, fq = lane >> 4;                                                                       \
    int offa, offw;                                                                                                 \
    {                                                                                                               \
        const int ra = wr * 64 + fr;                                   /* + 16 * mt */                              \
        const int rw = wc * 32 + fr;                                   /* + 16 * nt */                              \
        offa = ra * 128 + ((fq ^ ((ra >> 1) & 7)) << 4);                                                            \
        offw = rw * 128 + ((fq ^ ((rw >> 1) & 7)) << 4);                                                            \
    }                                                                                                               \
    const bf16_t* gsrc[4][2];        /* DMA sources [A0 A1 W0 W1][piece] */                                         \
    f32x4_t acc[2][4][2][2];         /* [i][mt][j][nt] */                                                           \
    bf16x8_t af[4][2], wx[2][2], wy[2][2]   /* A fragments [mt][ks]; the two W fragment buffers [nt][ks] (roles swap per K step) */

