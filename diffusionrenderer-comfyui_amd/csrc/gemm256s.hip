// 256x256x64 bf16 GEMM, "streamed" schedule (second generation of gemm256.hip):  C = epi(A[M,K] . W[N,K]^T)
//
// Measured on the first-generation kernel (tools/kbench.py, QKV shape, MI355X): MFMAs + barriers alone 0.92 ms, its memory
// pipeline alone (DMA + fragment reads + barriers) 0.96 ms, both together 1.50 ms - the fragment reads sit in bursts of 12 / 4 /
// 8 / 0 ds_read_b128 at the head of a phase, in front of a barrier, and a 12-read burst of four waves (48 KiB) outlasts the
// partner group's 16 MFMAs.  Here no wave ever stops to load:
//
//   * every wave runs ONE continuous MFMA stream, 4 groups of 16 v_mfma_f32_16x16x32_bf16 per K step - the quadrant pairs
//     (A0,W0) (A0,W1) (A1,W1) (A1,W0) as before - and requests the fragments of the group AFTER next in the issue shadow of
//     its own MFMAs: a fragment register is re-loaded right after the last MFMA that reads it (order [ks][mt][nt] inside a
//     group), so every fragment has a whole group (256+ cycles) of latency budget and no extra registers are needed:
//         group 1 (A0,W0): W1(k)   -> the free W buffer           (4 reads)
//         group 2 (A0,W1): A1(k)   -> af as its A0 entries die     (8 reads)
//         group 3 (A1,W1): W0(k+1) -> the W1 buffer as it dies     (4 reads; the W buffers swap roles every K step)
//         group 4 (A1,W0): A0(k+1) -> af as its A1 entries die     (8 reads)
//   * the two waves of a SIMD simply alternate on the matrix pipe; nothing is phase-locked, 4 barriers per K step (8 before)
//     only hand LDS regions over: a half-tile region is re-filled by DMA TWO groups after the group that read it last (its
//     reads are long complete by then - nothing waits for them)
//         group 1: W0(k+2)   group 2: A0(k+2)   group 3: W1(k+2)   group 4: A1(k+2)       (2 x 1 KiB pieces per wave each)
//     and read again 1.25 K steps later; with that order ONE counted wait, s_waitcnt vmcnt(10) in front of every barrier,
//     retires exactly the half-tile whose first read follows the barrier (5 half-tiles stay in flight per wave).
//
// Tried and rejected (round 2): (a) the same fragment reads as inline asm with hand-counted lgkmcnt waits instead of hipcc's
// lgkmcnt(0) after two of the four barriers: +-1 % (tools/kbench.py, interleaved A/B); (b) starting the first workgroup of
// every CU up to 31 x 0.5..4 us late so that the CUs' C-tile store bursts (256 x 128 KiB per round, ~6 % of a K = 4096 GEMM
// with the matrix pipes idle - the `G256S_ABL=8` build) do not coincide: 1.5-10 % SLOWER, monotonically in the stagger -
// tiles that run in lock step share their A / W panels in the XCD's L2, out of step they do not.
//
// Tile, LDS layout (2 stages x [A0 A1 W0 W1] x 16 KiB, 128-B rows, chunk ^ ((row>>1)&7), swizzle on the DMA source address),
// wave -> quadrant map, blocked operand layouts and the 16-byte epilogue stores are those of gemm256.hip.
#include <stdlib.h>
#include "drn_common.h"

// G256S_ABL: timing-only ablation (results WRONG; shipped with 0): 8 = no epilogue math / stores
#ifndef G256S_ABL
#define G256S_ABL 0
#endif
// G256S_ABL: timing-only ablation (results WRONG; shipped with 0): 8 = no epilogue math / stores
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

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm256s_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                          bf16_t* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                                                          int64_t ldw, int64_t ldc, const bf16_t* __restrict__ gate,
                                                          const bf16_t* R, int64_t ldr, int64_t rpb, int GROUP,
                                                          int abc, int64_t abs_, int cbc, int64_t cbs) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];     // 2 * STAGE_BYTES, the ONLY LDS object

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = (wave >> 1) & 1;            // which 64 rows of each A half
    const int wc = (wave & 1) | ((wave >> 2) << 1);    // which 32 rows (output columns) of each W half: 0..3

    const int tiles_m = (int)((M + TB - 1) / TB);
    const int tiles_n = (int)((N + TB - 1) / TB);
    const int nwg = tiles_m * tiles_n;
    int pid;
    {
        const int bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int width = GROUP * tiles_n;
    const int group_id = pid / width;
    const int first_m = group_id * GROUP;
    const int gsz = min(tiles_m - first_m, GROUP);
    const int tm = first_m + (pid % width) % gsz;
    const int tn = (pid % width) / gsz;
    const int64_t m0 = (int64_t)tm * TB, n0 = (int64_t)tn * TB;

    // ---- DMA source pointers: this wave's 2 pieces (16 rows) of each half-tile
    const bf16_t* gsrc[4][2];        // [A0 A1 W0 W1][piece]
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r = (wave * 2 + p) * 8 + (lane >> 3);           // row inside the half-tile, 0..127
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            if (h < 2) {
                int64_t row = m0 + h * 128 + r;
                if (row > M - 1) row = M - 1;
                gsrc[h][p] = A + row * lda + c * 8;
            } else {
                int64_t row = n0 + (h - 2) * 128 + r;
                if (row > N - 1) row = N - 1;
                gsrc[h][p] = W + row * ldw + c * 8;
            }
        }
    const int dma_off = wave * 2048;                                   // this wave's 2 KiB inside a half-tile region
    const int nk = (int)(K / BK);
    // blocked operand layouts (drn_gemm_bf16_blocked), see gemm256.hip
#define A_KOFF(KT) ((((int64_t)(KT) * BK) >> abc) * abs_ + (((int64_t)(KT) * BK) & ((1ll << abc) - 1)))
    const int64_t c_tile_off = (n0 >> cbc) * cbs + (n0 & ((1ll << cbc) - 1)) - n0;
    // half-tile H of K step KT (steps past the end re-request the last one into a region nobody reads: the wait counts stay uniform)
#define DMA(H, KT, S)                                                                                                  \
    do {                                                                                                               \
        const int kt_ = min((int)(KT), nk - 1);                                                                        \
        char* dst_ = smem + (S) * STAGE_BYTES + (H) * HALF_BYTES + dma_off;                                            \
        const int64_t ko_ = (H) < 2 ? A_KOFF(kt_) : (int64_t)kt_ * BK;                                                 \
        __builtin_amdgcn_global_load_lds((gptr_t)(gsrc[H][0] + ko_), (lptr_t)dst_, 16, 0, 0);                          \
        __builtin_amdgcn_global_load_lds((gptr_t)(gsrc[H][1] + ko_), (lptr_t)(dst_ + 1024), 16, 0, 0);                 \
    } while (0)

    // ---- fragment read offsets inside a half-tile region (k-substep 1 = offset ^ 64)
    const int fr = lane & 15, fq = lane >> 4;
    int offa, offw;
    {
        const int ra = wr * 64 + fr;                                   // + 16 * mt
        const int rw = wc * 32 + fr;                                   // + 16 * nt
        offa = ra * 128 + ((fq ^ ((ra >> 1) & 7)) << 4);
        offw = rw * 128 + ((fq ^ ((rw >> 1) & 7)) << 4);
    }
#define LD_A(S, I, MT, KS) (*reinterpret_cast<const bf16x8_t*>(smem + (S) * STAGE_BYTES + (I) * HALF_BYTES + ((KS) ? (offa ^ 64) : offa) + (MT) * 2048))
#define LD_W(S, J, NT, KS) (*reinterpret_cast<const bf16x8_t*>(smem + (S) * STAGE_BYTES + (2 + (J)) * HALF_BYTES + ((KS) ? (offw ^ 64) : offw) + (NT) * 2048))

    f32x4_t acc[2][4][2][2];       // [i][mt][j][nt]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[i][mt][j][nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    bf16x8_t af[4][2], wx[2][2], wy[2][2];   // A fragments [mt][ks]; the two W fragment buffers [nt][ks] (roles swap per K step)

#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define MM(I, MT, J, NT, KS, WF)                                                                                    \
    acc[I][MT][J][NT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[NT][KS], af[MT][KS], acc[I][MT][J][NT], 0, 0, 0)
    // end of a group: the half-tile whose first read follows has landed (own pieces: 5 half-tiles = 10 younger pieces may fly)
#define HANDOVER()                                                                                                  \
    do {                                                                                                            \
        FENCE();                                                                                                    \
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                                                           \
        __builtin_amdgcn_s_barrier();                                                                               \
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
    // K step KT (stage S = KT & 1, a literal) with W0(KT) in WA: groups 1..4.  On exit af = A0(KT+1), WB = W0(KT+1).
#define KSTEP(KT, S, WA, WB)                                                                                        \
    do {                                                                                                            \
        /* group 1 (A0,W0): request W1(KT) -> WB; DMA W0(KT+2) */                                                   \
        SLOT(0, 0, 0, 0, WA, WB[0][0] = LD_W(S, 1, 0, 0));                                                          \
        SLOT(0, 0, 1, 0, WA, WB[1][0] = LD_W(S, 1, 1, 0));                                                          \
        SLOT(0, 0, 2, 0, WA, WB[0][1] = LD_W(S, 1, 0, 1));                                                          \
        SLOT(0, 0, 3, 0, WA, WB[1][1] = LD_W(S, 1, 1, 1));                                                          \
        DMA(H_W0, (KT) + 2, S);                                                                                     \
        FENCE();                                                                                                    \
        SLOT(0, 0, 0, 1, WA, (void)0); SLOT(0, 0, 1, 1, WA, (void)0); SLOT(0, 0, 2, 1, WA, (void)0); SLOT(0, 0, 3, 1, WA, (void)0); \
        HANDOVER();                      /* A1(KT) has landed */                                                    \
        /* group 2 (A0,W1): request A1(KT) -> af as its A0 entries die; DMA A0(KT+2) */                             \
        SLOT(0, 1, 0, 0, WB, af[0][0] = LD_A(S, 1, 0, 0));                                                          \
        SLOT(0, 1, 1, 0, WB, af[1][0] = LD_A(S, 1, 1, 0));                                                          \
        SLOT(0, 1, 2, 0, WB, af[2][0] = LD_A(S, 1, 2, 0));                                                          \
        SLOT(0, 1, 3, 0, WB, af[3][0] = LD_A(S, 1, 3, 0));                                                          \
        DMA(H_A0, (KT) + 2, S);                                                                                     \
        FENCE();                                                                                                    \
        SLOT(0, 1, 0, 1, WB, af[0][1] = LD_A(S, 1, 0, 1));                                                          \
        SLOT(0, 1, 1, 1, WB, af[1][1] = LD_A(S, 1, 1, 1));                                                          \
        SLOT(0, 1, 2, 1, WB, af[2][1] = LD_A(S, 1, 2, 1));                                                          \
        SLOT(0, 1, 3, 1, WB, af[3][1] = LD_A(S, 1, 3, 1));                                                          \
        HANDOVER();                      /* W0(KT+1) has landed */                                                  \
        /* group 3 (A1,W1): request W0(KT+1) -> WB as it dies; DMA W1(KT+2) */                                      \
        SLOT(1, 1, 0, 0, WB, (void)0); SLOT(1, 1, 1, 0, WB, (void)0); SLOT(1, 1, 2, 0, WB, (void)0);                \
        SLOT(1, 1, 3, 0, WB, (WB[0][0] = LD_W((S) ^ 1, 0, 0, 0), WB[1][0] = LD_W((S) ^ 1, 0, 1, 0)));               \
        DMA(H_W1, (KT) + 2, S);                                                                                     \
        FENCE();                                                                                                    \
        SLOT(1, 1, 0, 1, WB, (void)0); SLOT(1, 1, 1, 1, WB, (void)0); SLOT(1, 1, 2, 1, WB, (void)0);                \
        SLOT(1, 1, 3, 1, WB, (WB[0][1] = LD_W((S) ^ 1, 0, 0, 1), WB[1][1] = LD_W((S) ^ 1, 0, 1, 1)));               \
        HANDOVER();                      /* A0(KT+1) has landed */                                                  \
        /* group 4 (A1,W0): request A0(KT+1) -> af as its A1 entries die; DMA A1(KT+2) */                           \
        SLOT(1, 0, 0, 0, WA, af[0][0] = LD_A((S) ^ 1, 0, 0, 0));                                                    \
        SLOT(1, 0, 1, 0, WA, af[1][0] = LD_A((S) ^ 1, 0, 1, 0));                                                    \
        SLOT(1, 0, 2, 0, WA, af[2][0] = LD_A((S) ^ 1, 0, 2, 0));                                                    \
        SLOT(1, 0, 3, 0, WA, af[3][0] = LD_A((S) ^ 1, 0, 3, 0));                                                    \
        DMA(H_A1, (KT) + 2, S);                                                                                     \
        FENCE();                                                                                                    \
        SLOT(1, 0, 0, 1, WA, af[0][1] = LD_A((S) ^ 1, 0, 0, 1));                                                    \
        SLOT(1, 0, 1, 1, WA, af[1][1] = LD_A((S) ^ 1, 0, 1, 1));                                                    \
        SLOT(1, 0, 2, 1, WA, af[2][1] = LD_A((S) ^ 1, 0, 2, 1));                                                    \
        SLOT(1, 0, 3, 1, WA, af[3][1] = LD_A((S) ^ 1, 0, 3, 1));                                                    \
        HANDOVER();                      /* W1(KT+1) has landed */                                                  \
    } while (0)

    // ---- prologue: K steps 0 and 1 requested in the steady-state order; W0(0) / A0(0) go to registers, W1(0) becomes visible
    DMA(H_W0, 0, 0); DMA(H_A0, 0, 0); DMA(H_W1, 0, 0); DMA(H_A1, 0, 0);
    DMA(H_W0, 1, 1); DMA(H_A0, 1, 1); DMA(H_W1, 1, 1); DMA(H_A1, 1, 1);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                  // W0(0), A0(0)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wx[nt][ks] = LD_W(0, 0, nt, ks);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) af[mt][ks] = LD_A(0, 0, mt, ks);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    FENCE();
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                  // W1(0): read in group 1 of K step 0
    __builtin_amdgcn_s_barrier();                                      // (and every wave holds W0(0) / A0(0): their regions are free)
    FENCE();

    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        KSTEP(kt, 0, wx, wy);
        KSTEP(kt + 1, 1, wy, wx);
    }
    if (kt < nk) KSTEP(kt, 0, wx, wy);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the re-requests past the last K step: nothing lands after exit

    if (G256S_ABL & 8) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) asm volatile("" :: "v"(acc[i][mt][j][nt]));
        return;
    }
    if (G256S_ABL & 8) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) asm volatile("" :: "v"(acc[i][mt][j][nt]));
        return;
    }
    // ---- epilogue (as gemm256.hip): column tiles nt = 0 / 1 exchanged between lane rows fq = 2k / 2k+1, 16-byte stores
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int64_t m_raw = m0 + i * 128 + wr * 64 + mt * 16 + fr;
            const bool m_ok = m_raw < M;
            const int64_t m = m_ok ? m_raw : M - 1;
            const int64_t b = (EPI == DRN_EPI_GATE_RES) ? m / rpb : 0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint2 o[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int64_t n = n0 + j * 128 + wc * 32 + nt * 16 + fq * 4;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = rbf(acc[i][mt][j][nt][r]);
                    if (EPI == DRN_EPI_GELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = gelu_erf_fast(v[r]);
                    } else if (EPI == DRN_EPI_GATE_RES) {
                        const uint2 g2 = *reinterpret_cast<const uint2*>(gate + b * N + n);
                        const uint2 r2 = *reinterpret_cast<const uint2*>(R + m * ldr + n);
                        const float g[4] = {bflo(g2.x), bfhi(g2.x), bflo(g2.y), bfhi(g2.y)};
                        const float x[4] = {bflo(r2.x), bfhi(r2.x), bflo(r2.y), bfhi(r2.y)};
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = x[r] + rbf(g[r] * v[r]);
                    }
                    o[nt].x = pack_bf2(v[0], v[1]);
                    o[nt].y = pack_bf2(v[2], v[3]);
                }
                const auto sx = __builtin_amdgcn_permlane16_swap(o[0].x, o[1].x, false, false);
                const auto sy = __builtin_amdgcn_permlane16_swap(o[0].y, o[1].y, false, false);
                const int64_t n8 = n0 + j * 128 + wc * 32 + (fq & 1) * 16 + (fq >> 1) * 8;
                if (m_ok && n8 < N)
                    *reinterpret_cast<uint4*>(C + m * ldc + n8 + c_tile_off) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
            }
        }
}

template <int EPI>
static int launch256s(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                      int64_t ldc, const void* gate, const void* residual, int64_t ldr, int64_t rpb, hipStream_t st,
                      const int64_t* blk) {
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256s_kernel<EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    const int64_t tiles = ((M + TB - 1) / TB) * ((N + TB - 1) / TB);
    if (tiles >= (1ll << 31)) return DRN_EINVAL;
    static int group = 0;
    if (group == 0) {
        const char* e = getenv("DRN_GEMM_GROUP");          // tile-rows per L2 band (A/B experiments)
        group = e ? atoi(e) : 4;
        if (group < 1) group = 4;
    }
    gemm256s_kernel<EPI><<<dim3((unsigned)tiles), dim3(512), 2 * STAGE_BYTES, st>>>(
        (const bf16_t*)A, (const bf16_t*)W, (bf16_t*)C, M, N, K, lda, ldw, ldc, (const bf16_t*)gate, (const bf16_t*)residual,
        ldr, rpb, group, (int)blk[0], blk[1], (int)blk[2], blk[3]);
    return drn_launch_status();
}

// called from gemm.hip (tile kernel 3); arguments already validated there
int drn_gemm256s_dispatch(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                          int64_t ldc, int epilogue, const void* gate, const void* residual, int64_t ldr, int64_t rpb,
                          void* stream, const int64_t* blk) {
    hipStream_t st = (hipStream_t)stream;
    switch (epilogue) {
        case DRN_EPI_NONE: return launch256s<DRN_EPI_NONE>(A, W, C, M, N, K, lda, ldw, ldc, gate, residual, ldr, rpb, st, blk);
        case DRN_EPI_GELU: return launch256s<DRN_EPI_GELU>(A, W, C, M, N, K, lda, ldw, ldc, gate, residual, ldr, rpb, st, blk);
        case DRN_EPI_GATE_RES: return launch256s<DRN_EPI_GATE_RES>(A, W, C, M, N, K, lda, ldw, ldc, gate, residual, ldr, rpb, st, blk);
        default: return DRN_EINVAL;
    }
}
