// 256x256x64 "ping-pong" bf16 GEMM for the large DiT linears (M >= 1024 tokens):  C = epi(A[M,K] . W[N,K]^T)
//
// One workgroup per CU: 8 waves = 2 groups of 4 (G0 = waves 0-3, G1 = waves 4-7); every SIMD hosts one wave of each
// group.  G1 runs ONE s_barrier behind G0, and every phase is  {LDS reads + next half-tile's global->LDS DMA} | barrier |
// {16 MFMAs} | barrier, so on each SIMD one wave feeds the matrix pipe while its partner issues memory work.
//
// Tile = 4 half-tiles per K step (A0, A1: 128 rows of A; W0, W1: 128 rows of W; 16 KiB each), two LDS stages (128 KiB).
// A wave owns the 64x32 output quadrant (i, j) of EVERY (A_i, W_j) pair:
//     rows  m0 + 128 i + 64 (wave&1..)   -> see ROW/COL below,      phases of a K step: (A0,W0) (A0,W1) (A1,W1) (A1,W0)
// so all waves touch the same half-tiles in the same phase and a half-tile region is DEAD once its phase has been read:
// A0 after phase 1, W1 after 2, A1 after 3, W0 after 4.  That lets the DMA run up to two K steps ahead with only two
// stages.  Per phase each wave issues the 2 x 1 KiB pieces of ONE half-tile:
//     phase 1: A1(k+1)   phase 2: W0(k+1)   phase 3: A0(k+2)   phase 4: W1(k+2)
// each >= 2 barrier intervals after the last read of the region it overwrites, and first read >= 3 phases later.
// One counted wait per K step: `s_waitcnt vmcnt(4)` in phase 4 (before its first barrier) retires everything up to
// W0(k+1), leaving A0(k+2) / W1(k+2) in flight across the K-step boundary.
//
// LDS rows are 128 B (64 bf16 of K); chunk index XOR (row>>1)&7 makes ds_read_b128 conflict-free; the swizzle is applied
// on the per-lane DMA source address (global_load_lds writes LDS lane-linearly).  MFMA issued as D = Wfrag x Afrag, so a
// lane owns 4 consecutive output features of one token (8-byte stores).
#include <stdlib.h>
#include "drn_common.h"

// G256_ABL: timing-only ablation builds (results are WRONG; shipped with 0): 1 = no DMA in the K loop (stale LDS), 2 = no fragment
// reads, 4 = no MFMAs, 8 = no epilogue math / stores (accumulators kept alive)
#ifndef G256_ABL
#define G256_ABL 0
#endif
#define TB 256
#define BK 64
#define HALF_BYTES (128 * BK * 2)          // 16 KiB
#define STAGE_BYTES (4 * HALF_BYTES)       // A0 A1 W0 W1
#define R_A0 0
#define R_A1 HALF_BYTES
#define R_W0 (2 * HALF_BYTES)
#define R_W1 (3 * HALF_BYTES)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ float gelu_erf256(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm256_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                         bf16_t* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                                                         int64_t ldw, int64_t ldc, const bf16_t* __restrict__ gate,
                                                         const bf16_t* R, int64_t ldr, int64_t rpb, int GROUP,
                                                         int abc, int64_t abs_, int cbc, int64_t cbs) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];     // 2 * STAGE_BYTES, the ONLY LDS object

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                 // 0: G0, 1: G1 (one barrier behind)
    const int wr = (wave >> 1) & 1;            // which 64 rows of each A half
    const int wc = (wave & 1) | (grp << 1);    // which 32 rows (output columns) of each W half: 0..3
    // (any bijection wave -> (wr, wc) works; groups differ in wc's high bit so both groups read every region)

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
    // blocked operand layouts (drn_gemm_bf16_blocked): K step kt of A starts at element A_KOFF(kt) of a row; the tile's
    // columns of C sit c_tile_off elements away from their plain position
    // (block widths are powers of two, passed as shifts; shift 62 = plain layout: the formulas then reduce to k / 0 without a branch)
#define A_KOFF(KT) ((((int64_t)(KT) * BK) >> abc) * abs_ + (((int64_t)(KT) * BK) & ((1ll << abc) - 1)))
    const int64_t c_tile_off = (n0 >> cbc) * cbs + (n0 & ((1ll << cbc) - 1)) - n0;
#define DMA(H, KT)                                                                                                     \
    do {                                                                                                               \
        if ((G256_ABL & 1) && (KT) > 1) break;                                                                         \
        char* dst_ = smem + ((KT) & 1) * STAGE_BYTES + (H) * HALF_BYTES + dma_off;                                     \
        const int64_t ko_ = (H) < 2 ? A_KOFF(KT) : (int64_t)(KT) * BK;                                                 \
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

    f32x4_t acc[2][4][2][2];       // [i][mt][j][nt]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[i][mt][j][nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    bf16x8_t af[4][2] = {}, wf0[2][2] = {}, wf1[2][2] = {};   // [mt][ks], W0 / W1 fragments [nt][ks] (W0 lives through phases 1..4)

#define READ_A(STAGE, I)                                                                                \
    do {                                                                                                \
        if (G256_ABL & 2) break;                                                                        \
        const char* b_ = smem + (STAGE) * STAGE_BYTES + (I) * HALF_BYTES;                               \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                              \
            af[mt][0] = *reinterpret_cast<const bf16x8_t*>(b_ + offa + mt * 2048);                      \
            af[mt][1] = *reinterpret_cast<const bf16x8_t*>(b_ + (offa ^ 64) + mt * 2048);               \
        }                                                                                               \
    } while (0)
#define READ_W(STAGE, J, WF)                                                                            \
    do {                                                                                                \
        if (G256_ABL & 2) break;                                                                        \
        const char* b_ = smem + (STAGE) * STAGE_BYTES + (2 + (J)) * HALF_BYTES;                         \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {                                              \
            WF[nt][0] = *reinterpret_cast<const bf16x8_t*>(b_ + offw + nt * 2048);                      \
            WF[nt][1] = *reinterpret_cast<const bf16x8_t*>(b_ + (offw ^ 64) + nt * 2048);               \
        }                                                                                               \
    } while (0)
#define MMA(I, J, WF)                                                                                   \
    do {                                                                                                \
        __builtin_amdgcn_s_setprio(1);                                                                  \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                \
            _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                            \
                _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                        \
                    if (G256_ABL & 4) { asm volatile("" :: "v"(WF[nt][ks]), "v"(af[mt][ks])); } else                        \
                    acc[I][mt][J][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[nt][ks], af[mt][ks], acc[I][mt][J][nt], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                  \
    } while (0)
#define SYNC_THEN_COMPUTE()                                                                             \
    do {                                                                                                \
        __builtin_amdgcn_s_barrier();                                                                   \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                              \
        __builtin_amdgcn_sched_barrier(0);                                                              \
    } while (0)
#define END_PHASE()                                                                                     \
    do {                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        __builtin_amdgcn_s_barrier();                                                                   \
    } while (0)

    const int nk = (int)(K / BK);
    // ---- prologue: K step 0 complete, A0/W1 of K step 1 in flight
    DMA(0, 0); DMA(1, 0); DMA(2, 0); DMA(3, 0);
    if (nk > 1) {
        DMA(0, 1); DMA(3, 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();          // G1 runs one barrier behind G0

    for (int kt = 0; kt < nk; ++kt) {
        const int st = kt & 1;
        // phase 1: (A0, W0)
        READ_W(st, 0, wf0);
        __builtin_amdgcn_sched_barrier(0);
        READ_A(st, 0);
        if (kt + 1 < nk) DMA(1, kt + 1);
        SYNC_THEN_COMPUTE();
        MMA(0, 0, wf0);
        END_PHASE();
        // phase 2: (A0, W1)
        READ_W(st, 1, wf1);
        if (kt + 1 < nk) DMA(2, kt + 1);
        SYNC_THEN_COMPUTE();
        MMA(0, 1, wf1);
        END_PHASE();
        // phase 3: (A1, W1)
        READ_A(st, 1);
        if (kt + 2 < nk) DMA(0, kt + 2);
        SYNC_THEN_COMPUTE();
        MMA(1, 1, wf1);
        END_PHASE();
        // phase 4: (A1, W0) - W0 fragments are still in registers from phase 1, no LDS reads here
        if (kt + 2 < nk) {
            DMA(3, kt + 2);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");     // everything up to W0(kt+1) has landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        SYNC_THEN_COMPUTE();
        MMA(1, 0, wf0);
        END_PHASE();
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();          // balance G1's extra barrier

    // ---- epilogue.  After the MFMAs lane (fr, fq) holds C[m = ..+fr][n..n+3] for n = 16 nt + 4 fq: stored as it stands that
    //      is 32 x 8 B per lane, 32 B contiguous per row and instruction (store-issue bound, partial lines).  The two column
    //      tiles nt = 0 / 1 are therefore exchanged between the lane rows fq = 2k and 2k+1 (v_permlane16_swap: odd rows of the
    //      first operand <-> even rows of the second), after which a lane owns 8 consecutive outputs:
    //      fq 0: n 0..7, fq 1: n 16..23, fq 2: n 8..15, fq 3: n 24..31  ->  16 x 16-B stores, 64 B contiguous per row.
    //      Partners share fr, i.e. the same output row, so row validity is identical on both sides of a swap; every lane
    //      executes the swaps (rows past M compute on a clamped row and are not stored).
    if (G256_ABL & 8) {
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
static int launch256(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                     int64_t ldc, const void* gate, const void* residual, int64_t ldr, int64_t rpb, hipStream_t st,
                     const int64_t* blk) {
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<EPI>),
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
    gemm256_kernel<EPI><<<dim3((unsigned)tiles), dim3(512), 2 * STAGE_BYTES, st>>>(
        (const bf16_t*)A, (const bf16_t*)W, (bf16_t*)C, M, N, K, lda, ldw, ldc, (const bf16_t*)gate, (const bf16_t*)residual,
        ldr, rpb, group, (int)blk[0], blk[1], (int)blk[2], blk[3]);
    return drn_launch_status();
}

// called from drn_gemm_bf16 (gemm.hip) for large problems; arguments already validated there
int drn_gemm256_dispatch(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                         int64_t ldc, int epilogue, const void* gate, const void* residual, int64_t ldr, int64_t rpb,
                         void* stream, const int64_t* blk) {
    hipStream_t st = (hipStream_t)stream;
    switch (epilogue) {
        case DRN_EPI_NONE: return launch256<DRN_EPI_NONE>(A, W, C, M, N, K, lda, ldw, ldc, gate, residual, ldr, rpb, st, blk);
        case DRN_EPI_GELU: return launch256<DRN_EPI_GELU>(A, W, C, M, N, K, lda, ldw, ldc, gate, residual, ldr, rpb, st, blk);
        case DRN_EPI_GATE_RES: return launch256<DRN_EPI_GATE_RES>(A, W, C, M, N, K, lda, ldw, ldc, gate, residual, ldr, rpb, st, blk);
        default: return DRN_EINVAL;
    }
}
