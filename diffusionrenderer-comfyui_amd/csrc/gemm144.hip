// 144x256x64 bf16 GEMM for the token bands of sequence parallelism:  C = epi(A[M,K] . W[N,K]^T),  M = 2304 k
//
// Why 144 rows: a rank of an 8-GPU job owns S/8 = 2304 tokens.  256-row tiles cut that into 9 tile rows, and 9 x (N/256) tiles
// never fill 256 CUs evenly (N = 4096: 144 tiles = 56 % of one round).  2304 = 16 x 144, so 144-row tiles give 16 x (N/256) =
// 256 / 768 / 1024 workgroups for N = 4096 / 12288 / 16384: whole rounds on 256 CUs, one workgroup per CU.
//
// One workgroup = 8 waves = 2 groups of 4; every SIMD hosts one wave of each group.  The groups split the K step: group g
// multiplies the k-substep g (32 of the 64 columns) of the WHOLE 144 x 256 tile, wave (g, wn) owning output columns
// 64 wn .. 64 wn + 63 -> 9 x 4 accumulator tiles (144 VGPRs), 13 ds_read_b128 per 36 MFMAs.  The two partial sums meet in
// LDS after the K loop.  As in gemm256s.hip the groups run ONE barrier apart and a phase is
// {fragment reads + global->LDS DMA} | barrier | {12 MFMAs} | barrier, three phases (48 rows each) per K step.
// Three LDS stages of 50 KiB (A 144 x 128 B, W 256 x 128 B): the DMA of K step k+2 overwrites the stage read in step k-1;
// one counted s_waitcnt vmcnt per K step.  XOR-swizzled rows as in gemm256s.hip.
#include <stdlib.h>
#include "drn_common.h"

#define TM 144
#define TN 256
#define BK 64
#define A_BYTES (TM * BK * 2)              // 18 KiB
#define W_BYTES (TN * BK * 2)              // 32 KiB
#define STAGE_BYTES (A_BYTES + W_BYTES)    // 50 KiB
#define NSTAGE 3

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// ONECLIP (gated-residual epilogue): the launcher vouches that no tile straddles two clips (rows_per_batch % 144 == 0 or one clip)
template <int EPI, bool ONECLIP = false>
__global__ __launch_bounds__(512, 2) void gemm144_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                         bf16_t* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                                                         int64_t ldw, int64_t ldc, const bf16_t* __restrict__ gate,
                                                         const bf16_t* R, int64_t ldr, int64_t rpb, int GROUP,
                                                         int abc, int64_t abs_, int cbc, int64_t cbs) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];     // NSTAGE * STAGE_BYTES, the ONLY LDS object

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                 // k-substep of this wave; G1 runs one barrier behind G0
    const int wn = wave & 3;                   // 64 output columns

    const int tiles_m = (int)((M + TM - 1) / TM);
    const int tiles_n = (int)((N + TN - 1) / TN);
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
    const int64_t m0 = (int64_t)tm * TM, n0 = (int64_t)tn * TN;

    // ---- DMA pieces (1 KiB = 8 rows x 128 B) of this wave: 4 of W, 2 of A, waves 0/1 one more of A (rows 128..143)
    const bool extra = wave < 2;
    const bf16_t* gw[4];
    const bf16_t* ga[3];
    int dw[4], da[3];
    {
        const int lr = lane >> 3;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int r = wave * 32 + p * 8 + lr;
            int64_t row = n0 + r;
            if (row > N - 1) row = N - 1;
            gw[p] = W + row * ldw + (((lane & 7) ^ ((r >> 1) & 7)) << 3);
            dw[p] = A_BYTES + (wave * 32 + p * 8) * 128;
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int r0 = p < 2 ? wave * 16 + p * 8 : 128 + (wave & 1) * 8;
            const int r = r0 + lr;
            int64_t row = m0 + r;
            if (row > M - 1) row = M - 1;
            ga[p] = A + row * lda + (((lane & 7) ^ ((r >> 1) & 7)) << 3);
            da[p] = r0 * 128;
        }
    }
#define DMA_W(P, SOFF, KT) __builtin_amdgcn_global_load_lds((gptr_t)(gw[P] + (int64_t)(KT) * BK), (lptr_t)(smem + (SOFF) + dw[P]), 16, 0, 0)
    // blocked operand layouts (drn_gemm_bf16_blocked): K step kt of A starts at element a_koff(kt) of a row; the tile's
    // columns of C sit c_tile_off elements away from their plain position
    // (block widths are powers of two, passed as shifts; shift 62 = plain layout: the formulas then reduce to k / 0 without a branch)
#define A_KOFF(KT) ((((int64_t)(KT) * BK) >> abc) * abs_ + (((int64_t)(KT) * BK) & ((1ll << abc) - 1)))
    const int64_t c_tile_off = (n0 >> cbc) * cbs + (n0 & ((1ll << cbc) - 1)) - n0;
#define DMA_A(P, SOFF, KT) __builtin_amdgcn_global_load_lds((gptr_t)(ga[P] + A_KOFF(KT)), (lptr_t)(smem + (SOFF) + da[P]), 16, 0, 0)
#define DMA_ALL(SOFF, KT)                                                           \
    do {                                                                            \
        DMA_W(0, SOFF, KT); DMA_W(1, SOFF, KT); DMA_W(2, SOFF, KT); DMA_W(3, SOFF, KT); \
        DMA_A(0, SOFF, KT); DMA_A(1, SOFF, KT);                                     \
        if (extra) DMA_A(2, SOFF, KT);                                              \
    } while (0)
    // wait until only the newest K step's pieces of this wave are in flight
#define WAIT_PREV_STEP()                                                            \
    do {                                                                            \
        if (extra) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");                 \
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                       \
    } while (0)

    // ---- fragment read offsets (bytes inside a stage); k-substep = grp
    const int fr = lane & 15, fq = lane >> 4;
    const int sw = (((grp << 2) | fq) ^ ((fr >> 1) & 7)) << 4;
    const int offa = fr * 128 + sw;                                    // + 2048 * mt
    const int offw = A_BYTES + (wn * 64 + fr) * 128 + sw;              // + 2048 * nt

    f32x4_t acc[9][4];
#pragma unroll
    for (int mt = 0; mt < 9; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    bf16x8_t wf[4], af[3];

#define READ_W(SOFF)                                                                                    \
    do {                                                                                                \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                \
            wf[nt] = *reinterpret_cast<const bf16x8_t*>(smem + (SOFF) + offw + nt * 2048);              \
    } while (0)
#define READ_A(SOFF, P)                                                                                 \
    do {                                                                                                \
        _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                   \
            af[i] = *reinterpret_cast<const bf16x8_t*>(smem + (SOFF) + offa + (3 * (P) + i) * 2048);    \
    } while (0)
#define MMA(P)                                                                                          \
    do {                                                                                                \
        __builtin_amdgcn_s_setprio(1);                                                                  \
        _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                   \
            _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                            \
                acc[3 * (P) + i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[i], acc[3 * (P) + i][nt], 0, 0, 0); \
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
    // ---- prologue: K step 0 landed, K step 1 in flight
    DMA_ALL(0, 0);
    if (nk > 1) {
        DMA_ALL(STAGE_BYTES, 1);
        WAIT_PREV_STEP();
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();          // G1 runs one barrier behind G0

    int soff = 0;                                        // stage of K step kt
    for (int kt = 0; kt < nk; ++kt) {
        const int noff = soff == 0 ? 2 * STAGE_BYTES : soff - STAGE_BYTES;   // stage of K step kt+2 (= the one read in kt-1)
        const bool more = kt + 2 < nk;
        // phase 0: rows 0..47
        READ_W(soff);
        READ_A(soff, 0);
        SYNC_THEN_COMPUTE();
        MMA(0);
        END_PHASE();
        // phase 1: rows 48..95; the stage of step kt-1 is dead for every wave by now
        READ_A(soff, 1);
        if (more) { DMA_W(0, noff, kt + 2); DMA_W(1, noff, kt + 2); DMA_W(2, noff, kt + 2); DMA_W(3, noff, kt + 2); }
        SYNC_THEN_COMPUTE();
        MMA(1);
        END_PHASE();
        // phase 2: rows 96..143
        READ_A(soff, 2);
        if (more) {
            DMA_A(0, noff, kt + 2); DMA_A(1, noff, kt + 2);
            if (extra) DMA_A(2, noff, kt + 2);
            WAIT_PREV_STEP();                            // K step kt+1 has landed (this wave's pieces)
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        SYNC_THEN_COMPUTE();
        MMA(2);
        END_PHASE();
        soff = soff == 2 * STAGE_BYTES ? 0 : soff + STAGE_BYTES;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();          // balance G1's extra barrier
    __syncthreads();                                     // every wave is past its last LDS read

    // ---- the two k-substep partial sums meet in LDS: G0 finishes columns nt 0,1 of its 64, G1 columns nt 2,3.
    //      (static accumulator indices only - a run-time choice between acc[..][h] and acc[..][2+h] would send acc to scratch: G1 swaps
    //      its halves under a uniform branch instead, see below)
    char* mine = smem + ((wn * 2 + grp) * 18) * 1024 + lane * 16;          // partner's contribution to MY tiles
    char* theirs = smem + ((wn * 2 + (grp ^ 1)) * 18) * 1024 + lane * 16;  // my contribution to the partner's tiles
#define GIVE()                                                                                          \
    do {                                                                                                \
        _Pragma("unroll") for (int mt = 0; mt < 9; ++mt)                                                \
            _Pragma("unroll") for (int h = 0; h < 2; ++h)                                               \
                *reinterpret_cast<f32x4_t*>(theirs + (mt * 2 + h) * 1024) = acc[mt][2 + h];       \
    } while (0)
#define TAKE()                                                                                          \
    do {                                                                                                \
        _Pragma("unroll") for (int mt = 0; mt < 9; ++mt) {                                              \
            _Pragma("unroll") for (int h = 0; h < 2; ++h)                                               \
                acc[mt][h] += *reinterpret_cast<const f32x4_t*>(mine + (mt * 2 + h) * 1024);    \
            if (mt % 2 == 1 || mt == 8) {     /* 4 reads in flight, not 18 (the residual registers are live): the sums are formed HERE */ \
                _Pragma("unroll") for (int m3 = (mt == 8 ? 8 : mt - 1); m3 <= mt; ++m3)                 \
                    asm volatile("" : "+v"(acc[m3][0]), "+v"(acc[m3][1]));                  \
                __builtin_amdgcn_sched_barrier(0);                                                      \
            }                                                                                           \
        }                                                                                               \
    } while (0)
    // epilogue of the owned tiles: lane holds C[m][n..n+3] of both; the two column tiles are exchanged between lane rows
    // fq = 2k / 2k+1 (v_permlane16_swap) so that a lane stores 8 consecutive outputs, 16 B (see gemm256s.hip)
    // gate / residual of the owned tiles: all 9 row tiles' loads in flight at once, requested right after this wave gave its other
    // half away (those 72 accumulator registers are free) so that their latency runs under the barrier and the LDS exchange; issued
    // per row tile in front of its store they were 9 dependent memory round trips (the residual may alias C: hipcc cannot move a
    // load above the previous store).  Every element is loaded and stored by lanes of the same wave (the permlane16 partner).
    uint2 g1_[2], r2_[9][2];
    const int64_t b_tile = (EPI == DRN_EPI_GATE_RES) ? (int64_t)((uint32_t)m0 / (uint32_t)rpb) : 0;     // launcher: M < 2^31
    constexpr bool one_clip = ONECLIP;                                // the tile lies inside one clip: one gate row
#define LOAD_RES(OWN)                                                                                   \
    do {                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        if (EPI == DRN_EPI_GATE_RES && one_clip) {                                                      \
            _Pragma("unroll") for (int h = 0; h < 2; ++h)                                               \
                g1_[h] = *reinterpret_cast<const uint2*>(gate + b_tile * N + n0 + wn * 64 + ((OWN) + h) * 16 + fq * 4); \
            /* a uniform base + 32-bit lane offsets (launcher: 144 rows x ldr x 2 B < 2^31): no 64-bit address per row tile */ \
            const char* rb_ = reinterpret_cast<const char*>(R + m0 * ldr + n0 + wn * 64 + (OWN) * 16);  \
            const int last_ = (int)min((int64_t)(TM - 1), M - 1 - m0);                                  \
            _Pragma("unroll") for (int mt = 0; mt < 9; ++mt) {                                          \
                const uint32_t ro_ = (uint32_t)((min(mt * 16 + fr, last_) * (int)ldr + fq * 4) * 2);    \
                _Pragma("unroll") for (int h = 0; h < 2; ++h)                                           \
                    r2_[mt][h] = *reinterpret_cast<const uint2*>(rb_ + ro_ + h * 32);                   \
            }                                                                                           \
        }                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                              \
    } while (0)
    // PRE = 1: gate / residual come from g1_ / r2_ (tile inside one clip); 0: loaded per row tile (a tile that straddles two clips)
#define STORE_TILES(OWN, PRE)                                                                           \
    do {                                                                                                \
        _Pragma("unroll") for (int mt = 0; mt < 9; ++mt) {                                              \
            const int64_t m_raw = m0 + mt * 16 + fr;                                                    \
            const bool m_ok = m_raw < M;                                                                \
            const int64_t m = m_ok ? m_raw : M - 1;                                                     \
            uint2 o[2];                                                                                 \
            _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                             \
                float v[4];                                                                             \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) v[r] = rbf(acc[mt][h][r]);        \
                if (EPI == DRN_EPI_GELU) {                                                              \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r) v[r] = gelu_erf_fast(v[r]);           \
                } else if (EPI == DRN_EPI_GATE_RES) {                                                   \
                    const int64_t n_ = n0 + wn * 64 + ((OWN) + h) * 16 + fq * 4;                        \
                    const uint2 g2 = (PRE) ? g1_[h] : *reinterpret_cast<const uint2*>(gate + (int64_t)((uint32_t)m / (uint32_t)rpb) * N + n_); \
                    const uint2 r2 = (PRE) ? r2_[mt][h] : *reinterpret_cast<const uint2*>(R + m * ldr + n_); \
                    const float g[4] = {bflo(g2.x), bfhi(g2.x), bflo(g2.y), bfhi(g2.y)};                \
                    const float x[4] = {bflo(r2.x), bfhi(r2.x), bflo(r2.y), bfhi(r2.y)};                \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r) v[r] = x[r] + rbf(g[r] * v[r]);       \
                }                                                                                       \
                o[h].x = pack_bf2(v[0], v[1]);                                                          \
                o[h].y = pack_bf2(v[2], v[3]);                                                          \
            }                                                                                           \
            const auto sx = __builtin_amdgcn_permlane16_swap(o[0].x, o[1].x, false, false);             \
            const auto sy = __builtin_amdgcn_permlane16_swap(o[0].y, o[1].y, false, false);             \
            const int64_t n8 = n0 + wn * 64 + (OWN) * 16 + (fq & 1) * 16 + (fq >> 1) * 8;               \
            if (m_ok && n8 < N)                                                                         \
                *reinterpret_cast<uint4*>(C + m * ldc + n8 + c_tile_off) = make_uint4(sx[0], sy[0], sx[1], sy[1]); \
        }                                                                                               \
    } while (0)
    // ONE code path for both groups: G1 swaps its halves first (72 v_swap, uniform branch), so that "own" is acc[..][0..1] and "given"
    // acc[..][2..3] for every wave; which column tiles those are (own) only enters the addresses.  (Two copies of the epilogue, one per
    // group, left the 72 registers of the half a wave has given away unused and spilled the residual instead.)
    if (grp == 1) {
#pragma unroll
        for (int mt = 0; mt < 9; ++mt)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4_t t = acc[mt][h];
                acc[mt][h] = acc[mt][2 + h];
                acc[mt][2 + h] = t;
            }
    }
    const int own = grp * 2;
    GIVE();
    LOAD_RES(own);
    __syncthreads();
    TAKE();
    STORE_TILES(own, ONECLIP);
}

template <int EPI, bool ONECLIP>
static int launch144_k(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                       int64_t ldc, const void* gate, const void* residual, int64_t ldr, int64_t rpb, hipStream_t st,
                       const int64_t* blk, int64_t tiles, int group) {
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm144_kernel<EPI, ONECLIP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * STAGE_BYTES);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    gemm144_kernel<EPI, ONECLIP><<<dim3((unsigned)tiles), dim3(512), NSTAGE * STAGE_BYTES, st>>>(
        (const bf16_t*)A, (const bf16_t*)W, (bf16_t*)C, M, N, K, lda, ldw, ldc, (const bf16_t*)gate, (const bf16_t*)residual,
        ldr, rpb, group, (int)blk[0], blk[1], (int)blk[2], blk[3]);
    return drn_launch_status();
}

template <int EPI>
static int launch144(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                     int64_t ldc, const void* gate, const void* residual, int64_t ldr, int64_t rpb, hipStream_t st,
                     const int64_t* blk) {
    const int64_t tiles = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
    if (tiles >= (1ll << 31) || M >= (1ll << 31)) return DRN_EINVAL;
    if (rpb > M || rpb <= 0) rpb = M;                      // (32-bit row / rows_per_batch arithmetic in the epilogue)
    static int group = 0;
    if (group == 0) {
        const char* e = getenv("DRN_GEMM_GROUP");          // tile-rows per L2 band (A/B experiments)
        group = e ? atoi(e) : 4;
        if (group < 1) group = 4;
    }
    if constexpr (EPI == DRN_EPI_GATE_RES) {
        // every tile inside one clip and 32-bit residual offsets: the epilogue that requests the residual before the exchange
        if ((rpb == M || rpb % TM == 0) && residual && (int64_t)TM * ldr * 2 < (1ll << 31))
            return launch144_k<EPI, true>(A, W, C, M, N, K, lda, ldw, ldc, gate, residual, ldr, rpb, st, blk, tiles, group);
    }
    return launch144_k<EPI, false>(A, W, C, M, N, K, lda, ldw, ldc, gate, residual, ldr, rpb, st, blk, tiles, group);
}

// called from drn_gemm_bf16 (gemm.hip) when 144-row tiles fill the CUs better than 256- or 128-row tiles
int drn_gemm144_dispatch(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                         int64_t ldc, int epilogue, const void* gate, const void* residual, int64_t ldr, int64_t rpb,
                         void* stream, const int64_t* blk) {
    hipStream_t st = (hipStream_t)stream;
    switch (epilogue) {
        case DRN_EPI_NONE: return launch144<DRN_EPI_NONE>(A, W, C, M, N, K, lda, ldw, ldc, gate, residual, ldr, rpb, st, blk);
        case DRN_EPI_GELU: return launch144<DRN_EPI_GELU>(A, W, C, M, N, K, lda, ldw, ldc, gate, residual, ldr, rpb, st, blk);
        case DRN_EPI_GATE_RES: return launch144<DRN_EPI_GATE_RES>(A, W, C, M, N, K, lda, ldw, ldc, gate, residual, ldr, rpb, st, blk);
        default: return DRN_EINVAL;
    }
}
