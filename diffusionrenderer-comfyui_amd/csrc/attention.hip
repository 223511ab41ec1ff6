// Non-causal flash attention forward for head_dim 128 on gfx950 (replaces F.scaled_dot_product_attention +
// the sbhd<->bhsd permutes + the head flatten, CleanGeneralDIT.py:181-203, :299-304).
//
// Workgroup = 8 waves = 256 query rows (32 per wave), KV tile = 64 keys.  K tiles triple-, V tiles quadruple-buffered in LDS,
// filled by global->LDS DMA (global_load_lds_dwordx4, swizzle on the source address) two tiles ahead.
//
// Per wave and KV tile (v_mfma_f32_32x32x16_bf16 only):
//   S^T[key][q]  = K . Q^T          A = K rows (ds_read_b128, XOR-swizzled 256-B rows), B = Q (registers)
//       -> the query sits on the LANE, its 64 scores in 2x16 registers of lanes l and l^32:
//          row max / row sum are in-register reductions + one cross-lane exchange.
//   O^T[d][q]   += V^T . P^T        B = P^T taken straight from the S^T accumulator registers (cvt to bf16, no
//       lane movement; the k order inside a 16-step is permuted: element j of lane half h is key
//       16s + 8(j>>2) + 4h + (j&3)), A = V^T read with ds_read_b64_tr_b16 in that same key order.
//   O^T keeps the query on the lane too, so the online-softmax rescale is one per-lane scalar.
// Softmax in fp32 (exp2 with the scale folded in), P rounded to bf16 for the PV MFMA, row sum from the fp32 P.
//
// Schedule (measured on MI355X: with both waves of a SIMD free to run QK / softmax / PV as they come, the matrix work
// (2.8 ms per launch at cfg 3) and everything else (softmax VALU, LDS / DMA issue: 2.8 ms) added up to 5.0 ms - the two
// waves meet in their MFMA phases, contend for the one matrix pipe, then contend for the VALU).  So the tile loop is a
// PING-PONG of two segments, the wave groups G0 = waves 0-3 and G1 = waves 4-7 (one wave of each per SIMD) half a tile apart:
//     M(t) = PV(t-1) . QK(t)     32 MFMAs back to back; fragment reads issued between them, a block (8 MFMAs) ahead
//     S(t) = softmax(t)          ~160 VALU + the tile prefetch DMA, no MFMA
//     interval 2t:   G0 runs M(t),  G1 runs S(t-1)   | barrier |   interval 2t+1:  G0 runs S(t),  G1 runs M(t)   | barrier
// so on every SIMD one wave feeds the matrix pipe while its partner's VALU work fills the issue slots between its MFMAs.
#include <stdlib.h>
#include "drn_common.h"

// Build switches (A/B timing of variants in one process: tools/kbench.py --lib; the shipped library uses the defaults):
//   ATT_EPI_LDS 1: O is transposed through LDS and stored as whole 256-B rows (16 B per lane); 0: 8-B row-strided stores.
//   ATT_PRIO    1: a wave runs its MFMA segment at s_setprio 1 (its MFMAs win the issue arbitration against the partner's VALU).
#ifndef ATT_EPI_LDS
#define ATT_EPI_LDS 1
#endif
#ifndef ATT_PRIO
#define ATT_PRIO 1
#endif
// ATT_ABL: timing-only ablation builds (results are WRONG; the shipped library is built with 0): 1 = no exp / row sum (softmax
// VALU reduced to the max + bf16 pack), 2 = no barriers, 4 = no PV MFMAs, 8 = no QK MFMAs, 16 = no K/V DMA
#ifndef ATT_ABL
#define ATT_ABL 0
#endif
// ATT_DIAG 1: diagnostic build (tools/kbench.py attn --diag): every wave sums the shader cycles (s_memtime) it spends in M work /
// waiting at the end of M (DMA + barrier) / in S work / waiting at the end of S, plus loop cycles and 100 MHz ticks (the clock
// the chip held), and lane 0 stores the eight sums to the split-KV workspace.  Stamps drain the LDS prefetch: read SHARES only.
#ifndef ATT_DIAG
#define ATT_DIAG 0
#endif
// ATT_DMA_IN_S: how many of a wave's 4 DMA pieces per tile are requested in the softmax segment instead of the MFMA segment
// (0..4; a piece costs its wave ~60-80 cycles of issue)
#ifndef ATT_DMA_IN_S
#define ATT_DMA_IN_S 0
#endif
// (pieces requested in S(t) would use the row offsets that M_QK(t) has already clamped for the LAST tile while they still belong
//  to a full tile: the variant needs its own unclamped offsets before it may be built again)
static_assert(ATT_DMA_IN_S == 0, "ATT_DMA_IN_S > 0 reads clamped K/V rows for tile nt-2: fix S_REQ's offsets first");
// ATT_PTR 1: the DMA source of a tile is a running scalar pointer advanced once per tile (0: recomputed per piece from the tile
// index - ~15 scalar instructions of 64-bit multiply per piece in front of the DMA, inside the MFMA stream)
#ifndef ATT_PTR
#define ATT_PTR 1
#endif
// ATT_TIE 1: a counted LDS wait ties its fragment's registers ("+v": hipcc then pads the following MFMA with an s_nop);
// 0: the wait is a bare s_waitcnt, its place is held by the sched_barrier fences alone
#ifndef ATT_TIE
#define ATT_TIE 1
#endif
// ATT_BAR1 1: ONE barrier per tile.  G0 runs [M(t) S(t)] | barrier, G1 runs [S(t-1) M(t)] | barrier: inside an interval the two
// groups still alternate on the matrix pipe (M beside S, then S beside M) but nothing stops them in the middle; the K / V rings
// (3 / 4 tiles) are deep enough that a tile's buffer is never re-filled in the interval that reads it.  0: barrier after every segment.
#ifndef ATT_BAR1
#define ATT_BAR1 0
#endif
#define QROWS 256        // query rows per workgroup
#define KVT 64           // keys per tile
#define KBYTES (KVT * 256)
#ifndef NKB
#define NKB 3            // K tile buffers
#endif
#define NVB 4            // V tile buffers
#ifndef RESCALE_THR
#define RESCALE_THR 6.0f   // log2 units
#endif

typedef __attribute__((address_space(3))) bf16x4_t* lds_b64_ptr;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;



__global__ __launch_bounds__(512, 2) void attention_fwd_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kp, const bf16_t* __restrict__ Vp, bf16_t* __restrict__ O,
    int heads, int64_t Sq, int64_t Sk_total, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t bsq, int64_t bsk,
    int64_t bsv, int64_t bso, float scale_log2e, int nqb, int total, int nsplit, int64_t kv_chunk,
    float* __restrict__ Opart, float* __restrict__ MLpart) {
    __shared__ __attribute__((aligned(1024))) char smem[(NKB + NVB) * KBYTES];   // K0..K2 V0..V3 - the ONLY LDS object

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;

    // XCD-aware order: each XCD walks consecutive (batch, head) pairs so the K/V stream of a head is L2-shared
    int pid;
    {
        const int bid = blockIdx.x;
        const int q = total >> 3, r = total & 7, xcd = bid & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    // pid -> ((batch*heads + head) * nsplit + split) * nqb + qblock: the q-blocks sharing one K/V chunk are adjacent
    const int qb = pid % nqb;
    const int rest = pid / nqb;
    const int split = rest % nsplit;
    const int bh = rest / nsplit;
    const int b = bh / heads, head = bh - b * heads;
    const int64_t q0 = (int64_t)qb * QROWS + wave * 32;
    // split-KV (flash-decoding style): this workgroup sees keys [kv_begin, kv_begin + Sk) only and, when nsplit > 1, emits an
    // un-normalised partial (O, m, l) that drn_attention_combine merges.  Used to fill the chip when (q-blocks x heads) is a
    // poor multiple of the 256 CUs, e.g. the 2304-query bands of 8-way sequence parallelism.
    const int64_t kv_begin = (int64_t)split * kv_chunk;
    const int64_t Sk = min(kv_chunk, Sk_total - kv_begin);

    const bf16_t* Qb = Q + b * bsq + (int64_t)head * 128;
    const bf16_t* Kb = Kp + b * bsk + kv_begin * ldk + (int64_t)head * 128;
    const bf16_t* Vb = Vp + b * bsv + kv_begin * ldv + (int64_t)head * 128;

    // ---- Q fragments: lane (lr, lh) holds Q[q0+lr][16*ks + 8*lh .. +7], ks = 0..7
    bf16x8_t qf[8];
    {
        int64_t qrow = q0 + lr;
        if (qrow > Sq - 1) qrow = Sq - 1;
        const bf16_t* qp = Qb + qrow * ldq + 8 * lh;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + 16 * ks);
        // make the Q loads complete HERE: with a DMA in flight hipcc waits vmcnt(0) at the first use of any ordinary load
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(qf[ks]));
    }

    // ---- staging: a K or V tile is 16 pieces of 1 KiB (4 rows x 256 B); wave w copies pieces 2w and 2w+1 of both by
    //      global->LDS DMA.  The DMA writes LDS lane-linearly (lane i -> byte 16 i of the piece), so the swizzle sits on the
    //      SOURCE address: the lane that fills chunk position cp of row r fetches chunk  cp ^ (r & 15)  (K, read back by
    //      ds_read_b128) or  (((cp >> 2) ^ (r & 3)) << 2) | (cp & 3)  (V, read back by ds_read_b64_tr_b16); both are
    //      involutions inside one 256-B row, so every row is still fetched as whole lines.
    int st_row[2], st_kc[2], st_vc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        st_row[i] = 4 * (2 * wave + i) + (lane >> 4);
        const int cp = lane & 15;
        st_kc[i] = (cp ^ (st_row[i] & 15)) * 8;
        st_vc[i] = ((((cp >> 2) ^ (st_row[i] & 3)) << 2) | (cp & 3)) * 8;
    }
    const int dma_off = wave * 2048;
    // per-lane BYTE offsets inside a tile (32-bit: 64 rows x ld); rows of the last, partial tile are clamped to its last key
    // (their scores are masked in QK_MASK) - recomputed once, when that tile is about to be requested
    uint32_t kso[2], vso[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        kso[i] = 2u * (uint32_t)(st_row[i] * (int)ldk + st_kc[i]);
        vso[i] = 2u * (uint32_t)(st_row[i] * (int)ldv + st_vc[i]);
    }
    const int last_rows = (int)(Sk - (int64_t)((Sk - 1) / KVT) * KVT);        // keys in the last tile, 1..64
#define CLAMP_LAST_TILE()                                                                     \
    do {                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                       \
            const int r_ = min(st_row[i], last_rows - 1);                                     \
            kso[i] = 2u * (uint32_t)(r_ * (int)ldk + st_kc[i]);                               \
            vso[i] = 2u * (uint32_t)(r_ * (int)ldv + st_vc[i]);                               \
        }                                                                                     \
    } while (0)
#define DMA16(SRC, DST) __builtin_amdgcn_global_load_lds((gptr_t)(SRC), (lptr_t)(DST), 16, 0, 0)
    // piece P (0..3: K0 V0 K1 V1) of tile TILE into K buffer KBUF / V buffer VBUF
#define DMA_PIECE(P, TILE, KBUF, VBUF)                                                        \
    do {                                                                                      \
        if ((P) & 1) {                                                                        \
            const char* vt_ = reinterpret_cast<const char*>(Vb + (int64_t)(TILE) * KVT * ldv); \
            DMA16(vt_ + vso[(P) >> 1], smem + (NKB + (VBUF)) * KBYTES + dma_off + ((P) >> 1) * 1024); \
        } else {                                                                              \
            const char* kt_ = reinterpret_cast<const char*>(Kb + (int64_t)(TILE) * KVT * ldk); \
            DMA16(kt_ + kso[(P) >> 1], smem + (KBUF) * KBYTES + dma_off + ((P) >> 1) * 1024); \
        }                                                                                     \
    } while (0)
    // the same with the tile's K / V base taken from the running pointers kreq_p / vreq_p (tile min(t + 2, nt - 1) during M(t))
#define DMA_PIECE_P(P, KBUF, VBUF)                                                            \
    do {                                                                                      \
        if ((P) & 1) DMA16(vreq_p + vso[(P) >> 1], smem + (NKB + (VBUF)) * KBYTES + dma_off + ((P) >> 1) * 1024); \
        else DMA16(kreq_p + kso[(P) >> 1], smem + (KBUF) * KBYTES + dma_off + ((P) >> 1) * 1024); \
    } while (0)
#define STAGE_TILE(TILE, KBUF, VBUF)                                                          \
    do {                                                                                      \
        DMA_PIECE(0, TILE, KBUF, VBUF); DMA_PIECE(1, TILE, KBUF, VBUF);                       \
        DMA_PIECE(2, TILE, KBUF, VBUF); DMA_PIECE(3, TILE, KBUF, VBUF);                       \
    } while (0)
    // this wave's DMA pieces have landed (the only vector-memory traffic inside the loop); the barrier that follows makes
    // every wave's pieces visible.  Raw s_barrier: __syncthreads() would add lgkmcnt(0) and drain the fragment prefetch.
#define DMA_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory")
#define BARRIER()                                                                             \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        if (!(ATT_ABL & 2)) __builtin_amdgcn_s_barrier();                                     \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    } while (0)

    // ---- LDS read addresses, kept in a few MUTABLE registers that step from buffer to buffer (DS instructions take
    //      VGPR + immediate only; left to hipcc, the dynamic buffer index turns into one hoisted address set per buffer)
    const uint32_t lds0 = (uint32_t)(uintptr_t)((lptr_t)smem);
    // K (A operand of S^T): row = 32*kt2 + lr, 16-B chunk (2*ks + lh) ^ (row & 15); kt2 = 1 is +8192 (immediate)
    uint32_t ka[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) ka[ks] = lds0 + lr * 256 + (((2 * ks + lh) ^ (lr & 15)) << 4);
    // V^T (A operand of O^T) via ds_read_b64_tr_b16: 16-lane group g reads a 4-key x 16-d block; lane i of the group
    // supplies row (i>>2), columns 4*(i&3)..+3 and receives column i.  Byte column inside the 256-B row for d-tile dt:
    // 64*dt + 32*vdh + 8*vp, swizzled to segment (dt ^ (key&3)); key&3 == vq for every block (block bases are % 4 == 0).
    uint32_t va[4];
    {
        const int vi = lane & 15;
        const int vq = vi >> 2, vp = vi & 3;
        const int vdh = (lane >> 4) & 1;          // which 16-wide half of the 32-d tile
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            va[dt] = lds0 + NKB * KBYTES + (4 * lh + vq) * 256 + ((dt ^ vq) << 6) + 32 * vdh + 8 * vp;   // + (32*kt2 + 16*s [+8]) * 256
    }

    f32x16_t acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    f32x16_t s[2];
    bf16x8_t pb[2][2] = {};

    // ---- phases, cut into fenced blocks: the fragment reads of one key half (kt2) go into a named block well ahead of the 8
    //      MFMAs that consume it.  A K block and a V block of the same half have disjoint live ranges (32 shared VGPRs).
#define FENCE() __builtin_amdgcn_sched_barrier(0)
    // All LDS reads of the loop are inline asm with hand-counted lgkmcnt waits: with a DMA in flight hipcc puts a
    // vmcnt(0) in front of every ds_read_b64_tr_b16 it knows about (the LDS-DMA vs LDS-read rule of its waitcnt pass), i.e.
    // it would drain the tile prefetch.  LDS returns data in order, so "the fragment requested N reads ago is complete"
    // is lgkmcnt(N) (field range 0..15).  A wait ties the fragment's registers ("+v") so that no consumer can be scheduled
    // above it; sched_barrier fences keep the MFMAs (register-only, not ordered by a memory clobber) in their slots.
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define RD_K(KT2, KS, DST)                                                                                         \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ka[KS]), "i"((KT2) * 8192) : "memory")
    // V fragment F = 4*sidx + dt of key half KT2: two transposed 4-key blocks (keys +0..3 and +8..11 of the 16-key step)
#define RD_V(KT2, F, DST)                                                                                          \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                      \
                 : "=&v"(DST.lo), "=&v"(DST.hi)                                                                    \
                 : "v"(va[(F) & 3]), "i"((32 * (KT2) + 16 * ((F) >> 2)) * 256), "i"((32 * (KT2) + 16 * ((F) >> 2) + 8) * 256) \
                 : "memory")
#if ATT_TIE
#define WAIT_K(N, X) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(X) : "i"(N) : "memory")
#define WAIT_V(N, X) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(X.lo), "+v"(X.hi) : "i"(N) : "memory")
#else
#define WAIT_K(N, X) asm volatile("s_waitcnt lgkmcnt(%0)" :: "i"(N) : "memory")
#define WAIT_V(N, X) asm volatile("s_waitcnt lgkmcnt(%0)" :: "i"(N) : "memory")
#endif
#define JOIN(X) __builtin_shufflevector(X.lo, X.hi, 0, 1, 2, 3, 4, 5, 6, 7)
#define MFMA(A, B, C) ((ATT_ABL & 4) ? (C) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0))
#define K_STEP(DELTA) do { _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) ka[ks] += (DELTA); } while (0)
#define V_STEP(DELTA) do { _Pragma("unroll") for (int dt = 0; dt < 4; ++dt) va[dt] += (DELTA); } while (0)
    // mask keys past Sk (last tile only)
#define QK_MASK(T)                                                                                                 \
    do {                                                                                                           \
        if ((int64_t)((T) + 1) * KVT > Sk) {                                                                       \
            const int64_t kbase = (int64_t)(T) * KVT + 4 * lh;                                                     \
            _Pragma("unroll") for (int kt2 = 0; kt2 < 2; ++kt2)                                                    \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                   \
                    const int64_t key = kbase + 32 * kt2 + (r & 3) + 8 * (r >> 2);                                 \
                    if (key >= Sk) s[kt2][r] = -INFINITY;                                                          \
                }                                                                                                  \
        }                                                                                                          \
    } while (0)

    // online softmax (query on the lane; partner lane^32 holds the other 32 keys) -> bf16 P^T fragments pb
#define SOFTMAX_PHASE()                                                                                            \
    do {                                                                                                           \
        float mx4[4];                                                                                              \
        _Pragma("unroll") for (int c = 0; c < 4; ++c) mx4[c] = fmaxf(s[0][c], s[1][c]);                            \
        _Pragma("unroll") for (int r = 4; r < 16; ++r) mx4[r & 3] = fmaxf(mx4[r & 3], fmaxf(s[0][r], s[1][r]));    \
        float mx = fmaxf(fmaxf(mx4[0], mx4[1]), fmaxf(mx4[2], mx4[3]));                                            \
        {                                                                                                          \
            const uint32_t mb = __float_as_uint(mx);                                                               \
            const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false); /* lane ^ 32, no LDS */        \
            mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));                                            \
        }                                                                                                          \
        /* deferred rescale: keep the old reference max while the new one is within 2^RESCALE_THR of it (P <= 2^THR, */ \
        /* exact in the final O / l ratio); most tiles skip the 64-register O rescale.  Wave-uniform decision.      */ \
        if (__any((mx - m_run) * scale_log2e > RESCALE_THR)) {                                                     \
            const float m_new = fmaxf(m_run, mx);                                                                  \
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);                             \
            m_run = m_new;                                                                                         \
            l_run *= alpha;                                                                                        \
            _Pragma("unroll") for (int dt = 0; dt < 4; ++dt)                                                       \
                _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[dt][r] *= alpha;                                \
        }                                                                                                          \
        const float mc = m_run * scale_log2e;                                                                      \
        float ps4[4] = {0.f, 0.f, 0.f, 0.f};                                                                       \
        _Pragma("unroll") for (int kt2 = 0; kt2 < 2; ++kt2) {                                                      \
            float p[16];                                                                                           \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                       \
                if (ATT_ABL & 1) { p[r] = s[kt2][r]; } else if (ATT_ABL & 64) {                                    \
                p[r] = s[kt2][r] * scale_log2e - mc; p[r] = p[r] * p[r] + mc;                                      \
                ps4[r & 3] += p[r]; } else {                                                                       \
                p[r] = __builtin_amdgcn_exp2f(s[kt2][r] * scale_log2e - mc);                                       \
                ps4[r & 3] += p[r]; }                                                                              \
            }                                                                                                      \
            _Pragma("unroll") for (int sidx = 0; sidx < 2; ++sidx) {                                               \
                union { bf16x8_t v; uint32_t u[4]; } cv;                                                           \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) cv.u[i] = pack_bf2(p[8 * sidx + 2 * i], p[8 * sidx + 2 * i + 1]); \
                pb[kt2][sidx] = cv.v;                                                                              \
            }                                                                                                      \
        }                                                                                                          \
        l_run += (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);                                                            \
    } while (0)

    const int nt = (int)((Sk + KVT - 1) / KVT);
    const int grp = wave >> 2;
    // ---- prologue: tiles 0 and 1 landed and visible
    if (nt == 1 && last_rows < KVT) CLAMP_LAST_TILE();
    STAGE_TILE(0, 0, 0);
    if (nt > 1) {
        if (nt == 2 && last_rows < KVT) CLAMP_LAST_TILE();
        STAGE_TILE(1, 1, 1);
    }
    DMA_WAIT(0);
    BARRIER();
    if (nt == 3 && last_rows < KVT) CLAMP_LAST_TILE();                // M(0) requests tile 2
    if (!ATT_BAR1 && grp == 1) BARRIER();          // G1 runs one interval behind G0
    const int64_t ktile_bytes = (int64_t)KVT * ldk * 2, vtile_bytes = (int64_t)KVT * ldv * 2;
    const char* kreq_p = reinterpret_cast<const char*>(Kb) + (int64_t)min(2, nt - 1) * ktile_bytes;   // tile that M(0) requests
    const char* vreq_p = reinterpret_cast<const char*>(Vb) + (int64_t)min(2, nt - 1) * vtile_bytes;

    bf16x8_t ka_[8], kb_[8];                       // K fragments of key half 0 / 1
    struct vfrag_t { bf16x4_t lo, hi; } va_[8], vb_[8];   // V fragments of key half 0 / 1 (halves as the transposed reads deliver them)
    bf16x8_t vc_[8];                               // first half of V[t], joined, carried from the end of S(t) into M(t+1)
    int kbuf = 0, vbuf = 0;                        // t % NKB, t % NVB
    int kreq = 2 % NKB, vreq = 2 % NVB;            // buffers of the tile that M(t) requests (t + 2)
    int kreq_s = kreq, vreq_s = vreq;              // the same for the pieces requested in S(t)

    // ---- M(t) = PV(t-1) . QK(t): four blocks of 8 MFMAs.  Every MFMA is followed, in its own issue shadow, by the fragment
    //      read(s) that the MFMA one block later needs (a full block = 256 cycles of latency budget), and is preceded by the
    //      counted wait for its own fragment: reads issued after fragment f of the running block = the rest of that block's
    //      reads + what this block has requested so far.  va points at V[t-1] during PV0, at V[t] from QK1 on (stepped
    //      during PV1); ka is stepped to K[t+1] during QK1.  One DMA piece of tile t+2 per block.
#define STEP_PV0(F)                                                                           \
    do {                                                                                      \
        acc[(F) & 3] = MFMA(vc_[F], pb[0][(F) >> 2], acc[(F) & 3]);                           \
        RD_V(1, F, vb_[F]);                                                                   \
        FENCE();                                                                              \
    } while (0)
#define STEP_PV1(F, WITH_K)                                                                   \
    do {                                                                                      \
        WAIT_V((WITH_K) ? 14 - (F) : 14 - 2 * (F), vb_[F]);                                   \
        FENCE();                                                                              \
        acc[(F) & 3] = MFMA(JOIN(vb_[F]), pb[1][(F) >> 2], acc[(F) & 3]);                     \
        if (WITH_K) RD_K(0, F, ka_[F]);                                                       \
        FENCE();                                                                              \
    } while (0)
#define STEP_QK0(KS)                                                                          \
    do {                                                                                      \
        WAIT_K(7, ka_[KS]);                                                                   \
        FENCE();                                                                              \
        s[0] = (ATT_ABL & 8) ? s[0] : __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka_[KS], qf[KS], s[0], 0, 0, 0); \
        RD_K(1, KS, kb_[KS]);                                                                 \
        FENCE();                                                                              \
    } while (0)
#define STEP_QK1(KS)                                                                          \
    do {                                                                                      \
        WAIT_K(7 + (KS), kb_[KS]);                                                            \
        FENCE();                                                                              \
        s[1] = (ATT_ABL & 8) ? s[1] : __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb_[KS], qf[KS], s[1], 0, 0, 0); \
        RD_V(0, KS, va_[KS]);                                                                 \
        FENCE();                                                                              \
    } while (0)
#define ZERO_S(KT2) do { _Pragma("unroll") for (int r = 0; r < 16; ++r) s[KT2][r] = 0.f; } while (0)
#define REQ_PIECE_NOW(P, T)                                                                   \
    do {                                                                                      \
        if (!(ATT_ABL & 16)) {                                                                \
            if (ATT_PTR) DMA_PIECE_P(P, kreq, vreq);                                          \
            else DMA_PIECE(P, min((T) + 2, nt - 1), kreq, vreq);                              \
        }                                                                                     \
        FENCE();                                                                              \
    } while (0)
    // pieces 0 .. ATT_DMA_IN_S-1 wait for the softmax segment (S_REQ), the others are requested between the MFMA blocks
#define REQ_PIECE(P, T) do { if ((P) >= ATT_DMA_IN_S) REQ_PIECE_NOW(P, T); } while (0)
#define S_REQ(T)                                                                              \
    do {                                                                                      \
        _Pragma("unroll") for (int p_ = 0; p_ < ATT_DMA_IN_S; ++p_)                           \
            if (!(ATT_ABL & 16)) DMA_PIECE(p_, min((T) + 2, nt - 1), kreq_s, vreq_s);         \
    } while (0)
    // QK(t) blocks + end of M(t): the DMA of tile t+1 (requested in M(t-1), older than this segment's 4 pieces) has landed
#define M_QK(T)                                                                               \
    do {                                                                                      \
        ZERO_S(0);                                                                            \
        STEP_QK0(0); STEP_QK0(1); STEP_QK0(2); STEP_QK0(3);                                   \
        REQ_PIECE(2, T);                                                                      \
        STEP_QK0(4); STEP_QK0(5); STEP_QK0(6); STEP_QK0(7);                                   \
        K_STEP(kbuf == NKB - 1 ? -(NKB - 1) * KBYTES : KBYTES);          /* K[t] -> K[t+1] */ \
        kbuf = kbuf == NKB - 1 ? 0 : kbuf + 1;                                                \
        ZERO_S(1);                                                                            \
        STEP_QK1(0); STEP_QK1(1); STEP_QK1(2); STEP_QK1(3);                                   \
        REQ_PIECE(3, T);                                                                      \
        STEP_QK1(4); STEP_QK1(5); STEP_QK1(6); STEP_QK1(7);                                   \
        QK_MASK(T);                                                                           \
        kreq_s = kreq; vreq_s = vreq;                                                         \
        kreq = kreq == NKB - 1 ? 0 : kreq + 1;                                                \
        vreq = vreq == NVB - 1 ? 0 : vreq + 1;                                                \
        if ((T) + 3 == nt - 1 && last_rows < KVT) CLAMP_LAST_TILE();    /* before M(t+1) requests the last tile */ \
        if (ATT_PTR && (T) + 3 <= nt - 1) { kreq_p += ktile_bytes; vreq_p += vtile_bytes; }   /* M(t+1) requests tile min(t+3, nt-1) */ \
        FENCE();                                                                              \
        if (ATT_PRIO) __builtin_amdgcn_s_setprio(0);                                          \
        STAMP(0);                        /* M work */                                         \
        if (!ATT_BAR1 || grp == 1) {                                                          \
            DMA_WAIT(4 - ATT_DMA_IN_S);  /* tile T+1: only this segment's pieces of tile T+2 are younger */ \
            STAMP(1);                    /* DMA wait */                                       \
            BARRIER();                                                                        \
        }                                                                                     \
        STAMP(2);                        /* barrier at the end of M */                        \
    } while (0)
    // ---- S(t): softmax only; the first V half of tile t (requested during QK1) is complete at its end
#define S_SEGMENT(T)                                                                          \
    do {                                                                                      \
        S_REQ(T);                                                                             \
        if (!(ATT_ABL & 32)) SOFTMAX_PHASE();                                                 \
        FENCE();                                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)"                                                   \
                     : "+v"(va_[0].lo), "+v"(va_[0].hi), "+v"(va_[1].lo), "+v"(va_[1].hi), "+v"(va_[2].lo), "+v"(va_[2].hi), \
                       "+v"(va_[3].lo), "+v"(va_[3].hi), "+v"(va_[4].lo), "+v"(va_[4].hi), "+v"(va_[5].lo), "+v"(va_[5].hi), \
                       "+v"(va_[6].lo), "+v"(va_[6].hi), "+v"(va_[7].lo), "+v"(va_[7].hi),     \
                       "+v"(pb[0][0]), "+v"(pb[0][1]), "+v"(pb[1][0]), "+v"(pb[1][1]), "+v"(l_run) :: "memory"); \
        /* (pb / l_run tied: the softmax is complete HERE - with a conditional barrier behind it hipcc otherwise sinks the exp / \
           pack half of the softmax into the block after the barrier) */                      \
        _Pragma("unroll") for (int f = 0; f < 8; ++f) vc_[f] = JOIN(va_[f]);                  \
        STAMP(3);                        /* S work */                                         \
        if (!ATT_BAR1) BARRIER();                                                             \
        else if (grp == 0) { DMA_WAIT(4); BARRIER(); }   /* (its M(t) requests are a whole segment old: nothing to wait for) */ \
        STAMP(4);                        /* barrier at the end of S */                        \
    } while (0)

#if ATT_DIAG
    unsigned long long dg_[6] = {0, 0, 0, 0, 0, 0}, dg_t = 0, dg_n = 0, dg_r0 = 0;
#define STAMP(SLOT)                                                                           \
    do {                                                                                      \
        FENCE();                                                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dg_n) :: "memory");       \
        FENCE();                                                                              \
        dg_[SLOT] += dg_n - dg_t;                                                             \
        dg_t = dg_n;                                                                          \
    } while (0)
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dg_r0) :: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dg_t) :: "memory");
    const unsigned long long dg_c0 = dg_t;
#else
#define STAMP(SLOT) do { } while (0)
#endif
    // tile 0: M(0) = QK(0) only (peeled: the loop body then has ONE fragment history, no merge)
    if (ATT_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) RD_K(0, ks, ka_[ks]);
    FENCE();
    REQ_PIECE(0, 0);
    REQ_PIECE(1, 0);
    M_QK(0);
    S_SEGMENT(0);
    for (int t = 1; t < nt; ++t) {
        if (ATT_PRIO) __builtin_amdgcn_s_setprio(1);
        STEP_PV0(0); STEP_PV0(1); STEP_PV0(2); STEP_PV0(3);
        REQ_PIECE(0, t);
        STEP_PV0(4); STEP_PV0(5); STEP_PV0(6); STEP_PV0(7);
        V_STEP(vbuf == NVB - 1 ? -(NVB - 1) * KBYTES : KBYTES);         // V[t-1] -> V[t] (all reads of V[t-1] are issued)
        vbuf = vbuf == NVB - 1 ? 0 : vbuf + 1;
        STEP_PV1(0, 1); STEP_PV1(1, 1); STEP_PV1(2, 1); STEP_PV1(3, 1);
        REQ_PIECE(1, t);
        STEP_PV1(4, 1); STEP_PV1(5, 1); STEP_PV1(6, 1); STEP_PV1(7, 1);
        M_QK(t);
        S_SEGMENT(t);
    }
    // M(nt) = PV(nt-1)
    if (ATT_PRIO) __builtin_amdgcn_s_setprio(1);
    STEP_PV0(0); STEP_PV0(1); STEP_PV0(2); STEP_PV0(3); STEP_PV0(4); STEP_PV0(5); STEP_PV0(6); STEP_PV0(7);
    STEP_PV1(0, 0); STEP_PV1(1, 0); STEP_PV1(2, 0); STEP_PV1(3, 0); STEP_PV1(4, 0); STEP_PV1(5, 0); STEP_PV1(6, 0); STEP_PV1(7, 0);
    if (ATT_PRIO) __builtin_amdgcn_s_setprio(0);
    DMA_WAIT(0);                                   // (the clamped re-requests of the last tile: nothing may land after the epilogue took the LDS)
    if (!ATT_BAR1 && grp == 0) BARRIER();          // balance G1's extra barrier
#if ATT_DIAG
    {
        unsigned long long dg_r1, dg_c1;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dg_r1) :: "memory");
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dg_c1) :: "memory");
        if (Opart && nsplit == 1 && lane == 0) {
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(Opart) + ((int64_t)blockIdx.x * 8 + wave) * 8;
            dst[0] = dg_[0]; dst[1] = dg_[1]; dst[2] = dg_[2]; dst[3] = dg_[3]; dst[4] = dg_[4];
            dst[5] = dg_c1 - dg_c0; dst[6] = dg_r1 - dg_r0; dst[7] = (unsigned long long)nt;
        }
    }
#endif

    // ---- epilogue: O[q][head*128 + d] = O^T[d][q] / l   (or the un-normalised partial when the keys are split)
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const int64_t qrow = q0 + lr;
    if (nsplit > 1) {
        if (qrow < Sq) {
            // partial layout: Opart[split][batch][q][heads*128] fp32, MLpart[split][batch][q][heads][2] = (m, l)
            const int nbatch = total / (nqb * heads * nsplit);
            const int64_t rowid = ((int64_t)split * nbatch + b) * Sq + qrow;
            float* op = Opart + rowid * ((int64_t)heads * 128) + (int64_t)head * 128 + 4 * lh;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    *reinterpret_cast<float4*>(op + 32 * dt + 8 * rg) =
                        make_float4(acc[dt][4 * rg + 0], acc[dt][4 * rg + 1], acc[dt][4 * rg + 2], acc[dt][4 * rg + 3]);
            if (lh == 0) {
                float2* ml = reinterpret_cast<float2*>(MLpart) + rowid * heads + head;
                *ml = make_float2(m_run, l_tot);
            }
        }
        return;
    }
    const float inv = 1.0f / l_tot;
#if ATT_EPI_LDS
    // The lane owns 4 consecutive d of ONE query row per (dt, rg): stored directly that is 16 x 8 B per lane at a row
    // stride (32 rows x 16 B per store instruction, partial lines).  Instead the wave's 32 x 128 tile goes through its own
    // 8 KiB of the (now dead) K/V buffers - 16-byte chunk c of row r at chunk position c ^ (r & 15) - and leaves as whole
    // rows: 16 lanes x 16 B = one 256-B row, 4 rows per store instruction.
    __syncthreads();                               // group 1's last PV still read a V buffer
    {
        char* ob = smem + wave * 8192;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                uint2 o;
                o.x = pack_bf2(acc[dt][4 * rg + 0] * inv, acc[dt][4 * rg + 1] * inv);
                o.y = pack_bf2(acc[dt][4 * rg + 2] * inv, acc[dt][4 * rg + 3] * inv);
                *reinterpret_cast<uint2*>(ob + lr * 256 + (((4 * dt + rg) ^ (lr & 15)) << 4) + 8 * lh) = o;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave's own LDS writes, read back by other lanes
        const int oc = lane & 15;
        bf16_t* op = O + b * bso + (int64_t)head * 128 + oc * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = 4 * j + (lane >> 4);
            const uint4 v = *reinterpret_cast<const uint4*>(ob + row * 256 + ((oc ^ (row & 15)) << 4));
            const int64_t qr = q0 + row;
            if (qr < Sq) *reinterpret_cast<uint4*>(op + qr * ldo) = v;
        }
    }
#else
    if (qrow < Sq) {
        bf16_t* op = O + b * bso + qrow * ldo + (int64_t)head * 128 + 4 * lh;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                uint2 o;
                o.x = pack_bf2(acc[dt][4 * rg + 0] * inv, acc[dt][4 * rg + 1] * inv);
                o.y = pack_bf2(acc[dt][4 * rg + 2] * inv, acc[dt][4 * rg + 3] * inv);
                *reinterpret_cast<uint2*>(op + 32 * dt + 8 * rg) = o;
            }
    }
#endif
}

// merge the nsplit partials of one (row, head): O = sum_s w_s O_s / sum_s w_s l_s,  w_s = 2^((m_s - max m) * scale*log2e)
__global__ __launch_bounds__(256) void attention_combine_kernel(const float* __restrict__ Opart, const float* __restrict__ MLpart,
                                                                bf16_t* __restrict__ O, int nsplit, int batch, int heads, int64_t Sq,
                                                                int64_t ldo, int64_t bso, float scale_log2e) {
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);      // (batch, row, head)
    const int64_t items = (int64_t)batch * Sq * heads;
    if (item >= items) return;
    const int head = (int)(item % heads);
    const int64_t brow = item / heads;                                       // batch * Sq + row
    const int64_t b = brow / Sq, row = brow - b * Sq;
    const int64_t split_stride = (int64_t)batch * Sq;
    const float2* ml = reinterpret_cast<const float2*>(MLpart);
    float mmax = -INFINITY;
    for (int s = 0; s < nsplit; ++s) mmax = fmaxf(mmax, ml[(s * split_stride + brow) * heads + head].x);
    float den = 0.f, o0 = 0.f, o1 = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float2 v = ml[(s * split_stride + brow) * heads + head];
        const float w = __builtin_amdgcn_exp2f((v.x - mmax) * scale_log2e);
        den += w * v.y;
        const float2 o = *reinterpret_cast<const float2*>(Opart + ((s * split_stride + brow) * heads + head) * 128 + 2 * lane);
        o0 += w * o.x;
        o1 += w * o.y;
    }
    const float inv = 1.0f / den;
    *reinterpret_cast<uint32_t*>(O + b * bso + row * ldo + (int64_t)head * 128 + 2 * lane) = pack_bf2(o0 * inv, o1 * inv);
}

// attention16.hip: the same kernel on v_mfma_f32_16x16x32_bf16 (same grid, same arguments)
void drn_attention16_launch(const void* q, const void* k, const void* v, void* o, int heads, int64_t Sq, int64_t Sk, int64_t ldq,
                            int64_t ldk, int64_t ldv, int64_t ldo, int64_t bsq, int64_t bsk, int64_t bsv, int64_t bso,
                            float scale_log2e, int nqb, int64_t total, int nsplit, int64_t kv_chunk, float* opart, float* mlpart,
                            hipStream_t st);
// default since round 3: the 16x16x32 body (tools/kbench.py attn --shapes 0,1: 4.21-4.33 vs 4.46-4.62 ms at cfg 3, in the model
// 121.8 vs 124.6 ms of attention per step; both bodies pass the same tests).  DRN_ATT16=0 selects the 32x32x16 body below.
#ifndef ATT_DEFAULT_SHAPE16
#define ATT_DEFAULT_SHAPE16 1
#endif
static int g_att16 = -1;       // -1: DRN_ATT16 from the environment (default ATT_DEFAULT_SHAPE16); 0 / 1 forced (tests, A/B)
extern "C" void drn_attention_force_shape16(int on) { g_att16 = on; }

static int attention_launch(const void* q, const void* k, const void* v, void* o, int batch, int heads, int64_t Sq, int64_t Sk,
                            int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t bsq, int64_t bsk, int64_t bsv,
                            int64_t bso, float scale, int nsplit, void* workspace, void* stream) {
    DRN_CHECK_ARG(q && k && v && o && batch > 0 && heads > 0 && Sq >= 0 && Sk > 0 && nsplit >= 1);
    DRN_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0);
    DRN_CHECK_ARG(bsq % 8 == 0 && bsk % 8 == 0 && bsv % 8 == 0 && bso % 8 == 0);
    DRN_CHECK_ARG(((uintptr_t)q & 15) == 0 && ((uintptr_t)k & 15) == 0 && ((uintptr_t)v & 15) == 0 && ((uintptr_t)o & 15) == 0);
    // the tile DMA addresses a K / V row as a 32-bit byte offset from the tile's first row (64 rows x ld x 2 B)
    DRN_CHECK_ARG(ldk > 0 && ldv > 0 && 64 * ldk * 2 < (1ll << 32) && 64 * ldv * 2 < (1ll << 32));
    if (Sq == 0) return DRN_OK;
    int64_t kv_chunk = Sk;
    if (nsplit > 1) {
        DRN_CHECK_ARG(workspace && ((uintptr_t)workspace & 15) == 0 && batch <= 65535);
        kv_chunk = ((Sk + nsplit - 1) / nsplit + KVT - 1) / KVT * KVT;
        nsplit = (int)((Sk + kv_chunk - 1) / kv_chunk);                       // no empty chunk
    }
    const int64_t nqb = (Sq + QROWS - 1) / QROWS;
    const int64_t total = nqb * heads * batch * nsplit;
    DRN_CHECK_ARG(total < (1ll << 31));
    const float scale_log2e = scale * 1.44269504088896340736f;
    float* opart = (float*)workspace;
    float* mlpart = opart ? opart + (int64_t)nsplit * batch * Sq * heads * 128 : nullptr;
    hipStream_t st = (hipStream_t)stream;
    static int env16 = -1;
    if (env16 < 0) {
        const char* e = getenv("DRN_ATT16");
        env16 = e ? (e[0] != '0') : ATT_DEFAULT_SHAPE16;
    }
    if (g_att16 >= 0 ? g_att16 : env16) {
        drn_attention16_launch(q, k, v, o, heads, Sq, Sk, ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso, scale_log2e, (int)nqb, total, nsplit,
                               kv_chunk, opart, mlpart, st);
    } else
    attention_fwd_kernel<<<dim3((unsigned)total, 1, 1), dim3(512), 0, st>>>(
        (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, heads, Sq, Sk, ldq, ldk, ldv, ldo, bsq, bsk,
        bsv, bso, scale_log2e, (int)nqb, (int)total, nsplit, kv_chunk, opart, mlpart);
    if (nsplit > 1) {
        const int64_t items = (int64_t)batch * Sq * heads;
        attention_combine_kernel<<<dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st>>>(opart, mlpart, (bf16_t*)o, nsplit, batch,
                                                                                            heads, Sq, ldo, bso, scale_log2e);
    }
    return drn_launch_status();
}

extern "C" int drn_attention_bf16(const void* q, const void* k, const void* v, void* o, int batch, int heads, int64_t Sq,
                                  int64_t Sk, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t bsq,
                                  int64_t bsk, int64_t bsv, int64_t bso, float scale, void* stream) {
    return attention_launch(q, k, v, o, batch, heads, Sq, Sk, ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso, scale, 1, nullptr, stream);
}

extern "C" int64_t drn_attention_splitkv_workspace_bytes(int batch, int heads, int64_t Sq, int nsplit) {
    return (int64_t)nsplit * batch * Sq * heads * (128 + 2) * (int64_t)sizeof(float);
}

extern "C" int drn_attention_splitkv_bf16(const void* q, const void* k, const void* v, void* o, int batch, int heads, int64_t Sq,
                                          int64_t Sk, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t bsq,
                                          int64_t bsk, int64_t bsv, int64_t bso, float scale, int nsplit, void* workspace,
                                          void* stream) {
    return attention_launch(q, k, v, o, batch, heads, Sq, Sk, ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso, scale, nsplit, workspace,
                            stream);
}
