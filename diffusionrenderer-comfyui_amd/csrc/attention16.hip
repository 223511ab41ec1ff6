// Flash attention forward for head_dim 128 on v_mfma_f32_16x16x32_bf16 (the second body of drn_attention_bf16; attention.hip holds
// the 32x32x16 one and the launcher).  Same workgroup (8 waves x 32 queries, KV tile 64), same K / V rings filled by LDS-DMA, same
// ping-pong of M(t) = PV(t-1) . QK(t) and S(t) = softmax(t) between the wave groups G0 / G1 - only the matrix shape differs.
// Why: tools/ubench/port_share.hip (profiles/r03_ubench_port_share.txt) - the same 1024 matrix-pipe cycles run at 2.08 GHz as
// 16x16x32 and at 1.73 GHz as 32x32x16 (the chip is power-limited under bf16 MFMA), while every MFMA *instruction* takes 8 cycles
// of VALU issue from the SIMD's other wave (64 instead of 32 per segment: the softmax partner pays +256 cycles per tile).
//
// Lane l = (c = l & 15, g = l >> 4).  A wave's 32 queries are two 16-query tiles qt; a 64-key tile is four 16-key tiles kt.
//   S^T[key][q] = K . Q^T :  A = K rows (ds_read_b128; MFMA row m of key tile kt is key 16 kt + pi(m), pi swaps bits 2 and 3 of m,
//                            i.e. lane group g holds keys kappa(g) + i, kappa = {0, 8, 4, 12}), B = Q (registers), 4 d-steps of 32
//                            -> sacc[kt][qt][i] = score of query 16 qt + c with key 16 kt + kappa(g) + i
//   O^T[d][q]  += V^T . P^T : B = P^T straight from the sacc registers (k index (g, j) of key step s = key 32 s + 16 (j >> 2) +
//                            kappa(g) + (j & 3)), A = V^T by two ds_read_b64_tr_b16 per fragment in that key order; the two 4-key
//                            blocks of a 32-lane half are 8 rows apart (that is what pi is for) and the V image XORs the chunk
//                            index with ((row & 3) << 1) | (((row >> 3) & 1) << 3): conflict-free
//                            -> acc[dt][qt][i] = O^T[16 dt + 4 g + i][16 qt + c]
// A query's 64 scores of a tile sit in the 4 lanes (c, g = 0..3): row max = in-lane tree + permlane32_swap + permlane16_swap.
//
// M segment = 32 fragments in order V(s, dt) x 16, K(kt, ks) x 16, two MFMAs (qt = 0, 1) per fragment; every fragment is read 8
// fragments (16 MFMAs = 256 matrix-pipe cycles) ahead into a ring of 8 slots, the read being issued in the shadow of the MFMAs that
// free the slot; hand-counted s_waitcnt lgkmcnt (V fragment = 2 LDS reads, K fragment = 1).  The last 8 reads of M(t) fetch the
// first V key step of tile t for M(t+1) and stay in flight across S(t).
#include "drn_common.h"

#define QROWS 256
#define KVT 64
#define KBYTES (KVT * 256)
#define NKB 3
#define NVB 4
#ifndef RESCALE_THR
#define RESCALE_THR 6.0f   // log2 units
#endif

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__global__ __launch_bounds__(512, 2) void attention16_fwd_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kp, const bf16_t* __restrict__ Vp, bf16_t* __restrict__ O,
    int heads, int64_t Sq, int64_t Sk_total, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t bsq, int64_t bsk,
    int64_t bsv, int64_t bso, float scale_log2e, int nqb, int total, int nsplit, int64_t kv_chunk,
    float* __restrict__ Opart, float* __restrict__ MLpart) {
    __shared__ __attribute__((aligned(1024))) char smem[(NKB + NVB) * KBYTES];   // K0..K2 V0..V3 - the ONLY LDS object

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;

    int pid;
    {
        const int bid = blockIdx.x;
        const int q = total >> 3, r = total & 7, xcd = bid & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int qb = pid % nqb;
    const int rest = pid / nqb;
    const int split = rest % nsplit;
    const int bh = rest / nsplit;
    const int b = bh / heads, head = bh - b * heads;
    const int64_t q0 = (int64_t)qb * QROWS + wave * 32;
    const int64_t kv_begin = (int64_t)split * kv_chunk;
    const int64_t Sk = min(kv_chunk, Sk_total - kv_begin);

    const bf16_t* Qb = Q + b * bsq + (int64_t)head * 128;
    const bf16_t* Kb = Kp + b * bsk + kv_begin * ldk + (int64_t)head * 128;
    const bf16_t* Vb = Vp + b * bsv + kv_begin * ldv + (int64_t)head * 128;

    // ---- Q fragments (B operand): lane (c, g) holds Q[q0 + 16 qt + c][32 ks + 8 g .. + 7]
    bf16x8_t qf[2][4];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        int64_t qrow = q0 + 16 * qt + c;
        if (qrow > Sq - 1) qrow = Sq - 1;
        const bf16_t* qp = Qb + qrow * ldq + 8 * g;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[qt][ks] = *reinterpret_cast<const bf16x8_t*>(qp + 32 * ks);
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+v"(qf[qt][ks]));      // the Q loads complete HERE (see attention.hip)

    // ---- staging (as attention.hip): a tile is 16 pieces of 1 KiB (4 rows x 256 B), wave w copies pieces 2w, 2w+1 of K and of V;
    //      the swizzle sits on the SOURCE address.  K: chunk ^ (row & 15).  V: chunk ^ (((row & 3) << 1) | (((row >> 3) & 1) << 3)).
    int st_row[2], st_kc[2], st_vc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        st_row[i] = 4 * (2 * wave + i) + (lane >> 4);
        const int cp = lane & 15;
        st_kc[i] = (cp ^ (st_row[i] & 15)) * 8;
        st_vc[i] = (cp ^ (((st_row[i] & 3) << 1) | (((st_row[i] >> 3) & 1) << 3))) * 8;
    }
    const int dma_off = wave * 2048;
    uint32_t kso[2], vso[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        kso[i] = 2u * (uint32_t)(st_row[i] * (int)ldk + st_kc[i]);
        vso[i] = 2u * (uint32_t)(st_row[i] * (int)ldv + st_vc[i]);
    }
    const int last_rows = (int)(Sk - (int64_t)((Sk - 1) / KVT) * KVT);
#define CLAMP_LAST_TILE()                                                                     \
    do {                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                       \
            const int r_ = min(st_row[i], last_rows - 1);                                     \
            kso[i] = 2u * (uint32_t)(r_ * (int)ldk + st_kc[i]);                               \
            vso[i] = 2u * (uint32_t)(r_ * (int)ldv + st_vc[i]);                               \
        }                                                                                     \
    } while (0)
#define DMA16(SRC, DST) __builtin_amdgcn_global_load_lds((gptr_t)(SRC), (lptr_t)(DST), 16, 0, 0)
#define DMA_PIECE_AT(P, KPTR, VPTR, KBUF, VBUF)                                               \
    do {                                                                                      \
        if ((P) & 1) DMA16((VPTR) + vso[(P) >> 1], smem + (NKB + (VBUF)) * KBYTES + dma_off + ((P) >> 1) * 1024); \
        else DMA16((KPTR) + kso[(P) >> 1], smem + (KBUF) * KBYTES + dma_off + ((P) >> 1) * 1024); \
    } while (0)
#define DMA_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory")
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define BARRIER()                                                                             \
    do {                                                                                      \
        FENCE();                                                                              \
        __builtin_amdgcn_s_barrier();                                                         \
        FENCE();                                                                              \
    } while (0)

    // ---- LDS read addresses (mutable: they step from ring buffer to ring buffer)
    const uint32_t lds0 = (uint32_t)(uintptr_t)((lptr_t)smem);
    const int pc = (c & 3) | ((c & 4) << 1) | ((c & 8) >> 1);                   // pi(c): the key row inside a 16-key tile this lane's MFMA row is
    uint32_t ka[4];                                                             // K fragment (kt, ks): ka[ks] + kt * 4096
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ka[ks] = lds0 + pc * 256 + (((4 * ks + g) ^ pc) << 4);
    const int vq = c >> 2, vp = c & 3;
    const int kap = ((g & 1) << 3) | ((g & 2) << 1);                            // kappa(g) = {0, 8, 4, 12}
    uint32_t va[8];                                                             // V fragment (s, dt), block u: va[dt] + (32 s + 16 u) * 256
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
        va[dt] = lds0 + NKB * KBYTES + (kap + vq) * 256 + 32 * (dt ^ (vq | ((g & 1) << 2))) + 16 * (vp >> 1) + 8 * (vp & 1);

    f32x4_t acc[8][2];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) acc[dt][qt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
    f32x4_t lacc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};       // ATT16_LSUM_MFMA: ones . P^T (every row = the row sum)
    bf16x8_t ones_frag;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones_frag[i] = (short)0x3f80;
    asm volatile("" : "+v"(ones_frag));                              // (a register operand, not eight literals per MFMA)
    f32x4_t sacc[4][2];
    bf16x8_t pb[2][2] = {};

    // ---- fragment rings (8 slots; a K and a V slot of the same index are never live together)
    bf16x8_t kring[8];
    struct vfrag_t { bf16x4_t lo, hi; } vring[8];
#define RD_K(F16, SLOT)   /* K fragment F16 = 4 kt + ks */                                                          \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kring[SLOT]) : "v"(ka[(F16) & 3]), "i"(((F16) >> 2) * 4096) : "memory")
#define RD_V(F16, SLOT)   /* V fragment F16 = 8 s + dt: blocks u = 0, 1 */                                           \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                         \
                 : "=&v"(vring[SLOT].lo), "=&v"(vring[SLOT].hi)                                                       \
                 : "v"(va[(F16) & 7]), "i"((32 * ((F16) >> 3)) * 256), "i"((32 * ((F16) >> 3) + 16) * 256) : "memory")
// ATT16_TIE 1: a counted wait ties its fragment's registers ("+v": hipcc then pads the following MFMA with an s_nop - one of the
// four issue slots a fragment step has between its two 16-cycle MFMAs); 0: bare s_waitcnt, held in place by the fences alone
#ifndef ATT16_TIE
#define ATT16_TIE 0
#endif
#if ATT16_TIE
#define WAIT_K(N, SLOT) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(kring[SLOT]) : "i"(N) : "memory")
#define WAIT_V(N, SLOT) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(vring[SLOT].lo), "+v"(vring[SLOT].hi) : "i"(N) : "memory")
#else
#define WAIT_K(N, SLOT) asm volatile("s_waitcnt lgkmcnt(%0)" :: "i"(N) : "memory")
#define WAIT_V(N, SLOT) asm volatile("s_waitcnt lgkmcnt(%0)" :: "i"(N) : "memory")
#endif
#define JOIN(X) __builtin_shufflevector(X.lo, X.hi, 0, 1, 2, 3, 4, 5, 6, 7)
#define MFMA16(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, C, 0, 0, 0)
    // fragment F of the M segment's order (0..15: V(F), 16..31: K(F - 16), 32..39: next segment's V(F - 32)) into slot F & 7
#define ISSUE(F)                                                                              \
    do {                                                                                      \
        if ((F) < 16) RD_V(F, (F) & 7);                                                       \
        else if ((F) < 32) RD_K((F) - 16, (F) & 7);                                           \
        else RD_V((F) - 32, (F) & 7);                                                         \
    } while (0)
    // LDS reads outstanding behind fragment F while fragments F+1 .. F+7 are in flight (V = 2 reads, K = 1)
#define NV_(F) ((F) < 16 ? 2 : ((F) < 32 ? 1 : 2))
#define PEND(F) (NV_((F) + 1) + NV_((F) + 2) + NV_((F) + 3) + NV_((F) + 4) + NV_((F) + 5) + NV_((F) + 6) + NV_((F) + 7))
    // PV step F (0..15): s = F >> 3, dt = F & 7.  NEXT: whether fragment F + 8 is to be requested (always in the loop)
// ATT16_PAIR 1: one counted wait per PAIR of fragments, placed in front of the even one and counted for the odd one (a wait is an
// issue slot; the look-ahead shrinks from 8 fragments to 7 for the even ones)
#ifndef ATT16_PAIR
#define ATT16_PAIR 0
#endif
// ATT16_DMA_IN_S 1: a wave requests its 4 DMA pieces of tile t + 2 at the start of S(t) (a VALU-only stretch) instead of between
// the MFMAs of M(t) (where a piece costs the tightly packed MFMA stream ~60-80 cycles and the partner's VALU as much)
#ifndef ATT16_DMA_IN_S
#define ATT16_DMA_IN_S 0
#endif
// (tried and rejected, kbench A/B in one process: the four waves of a group requesting a piece at four different fragment steps
//  instead of all at once: 4.33 -> 4.62 ms; the row sum by v_dot2c_f32_bf16 on the packed P: 4.32 -> 4.61 ms)
// ATT16_LSUM_MFMA 1: the softmax denominator comes off the matrix pipe - one extra MFMA per key step and query tile multiplies the
// bf16 P^T fragment by a fragment of ones (no LDS read), instead of 32 v_add_f32 per tile in the softmax segment, which is the
// longer of the two segments in this body; the sum is then that of the ROUNDED probabilities the PV product uses, and every lane of
// a query holds it (no cross-lane step in the epilogue).
#ifndef ATT16_LSUM_MFMA
#define ATT16_LSUM_MFMA 1
#endif
// ATT16_PRIO: s_setprio of the M segment (default 1: its MFMAs win the issue arbitration against the partner's softmax); S runs at 0
#ifndef ATT16_PRIO
#define ATT16_PRIO 1
#endif
#ifndef ATT16_SPRIO
#define ATT16_SPRIO 0
#endif
// ATT16_BAR1 1: one barrier per tile (G0: M S |, G1: S M |) instead of one after every segment (LDS-safe on the 3 + 4 tile rings);
// measured 4.236 -> 4.269 ms: not the lever
#ifndef ATT16_BAR1
#define ATT16_BAR1 0
#endif
// ATT16_ABL: timing-only ablations (results WRONG; shipped with 0): 1 = no exp (softmax VALU minus the 32 transcendentals),
// 2 = no softmax at all (M segments alone), 4 = no K/V DMA inside the loop, 8 = every DMA request re-reads tile 2 (always cached)
#ifndef ATT16_ABL
#define ATT16_ABL 0
#endif
// (tried: the row sum from the bf16-rounded P pairs with v_dot2c_f32_bf16, 16 instructions instead of 32 adds: 4.32 -> 4.61 ms)
#define PAIRED(F, PENDING) (ATT16_PAIR && !ATT16_TIE ? (((F) & 1) ? -1 : (PENDING) - NV_((F) + 1)) : (PENDING))
#define STEP_PV(F, PENDING, NEXT)                                                             \
    do {                                                                                      \
        if (PAIRED(F, PENDING) >= 0) WAIT_V(PAIRED(F, PENDING), (F) & 7);                     \
        FENCE();                                                                              \
        {                                                                                     \
            const bf16x8_t vf_ = JOIN(vring[(F) & 7]);                                        \
            acc[(F) & 7][0] = MFMA16(vf_, pb[(F) >> 3][0], acc[(F) & 7][0]);                  \
            acc[(F) & 7][1] = MFMA16(vf_, pb[(F) >> 3][1], acc[(F) & 7][1]);                  \
            if (ATT16_LSUM_MFMA && ((F) & 7) == 4) {       /* row sums of key step s (any step of it would do) */ \
                lacc[0] = MFMA16(ones_frag, pb[(F) >> 3][0], lacc[0]);                        \
                lacc[1] = MFMA16(ones_frag, pb[(F) >> 3][1], lacc[1]);                        \
            }                                                                                 \
        }                                                                                     \
        if (NEXT) ISSUE((F) + 8);                                                             \
        FENCE();                                                                              \
    } while (0)
    // QK step F (16..31): kt = (F - 16) >> 2, ks = (F - 16) & 3; the first d-step of a key tile starts from a zero accumulator
#define STEP_QK(F, PENDING)                                                                   \
    do {                                                                                      \
        if (PAIRED(F, PENDING) >= 0) WAIT_K(PAIRED(F, PENDING), (F) & 7);                     \
        FENCE();                                                                              \
        if ((((F) - 16) & 3) == 0) {                                                          \
            sacc[((F) - 16) >> 2][0] = MFMA16(kring[(F) & 7], qf[0][0], (f32x4_t{0.f, 0.f, 0.f, 0.f})); \
            sacc[((F) - 16) >> 2][1] = MFMA16(kring[(F) & 7], qf[1][0], (f32x4_t{0.f, 0.f, 0.f, 0.f})); \
        } else {                                                                              \
            sacc[((F) - 16) >> 2][0] = MFMA16(kring[(F) & 7], qf[0][((F) - 16) & 3], sacc[((F) - 16) >> 2][0]); \
            sacc[((F) - 16) >> 2][1] = MFMA16(kring[(F) & 7], qf[1][((F) - 16) & 3], sacc[((F) - 16) >> 2][1]); \
        }                                                                                     \
        ISSUE((F) + 8);                                                                       \
        FENCE();                                                                              \
    } while (0)
#define K_STEP(DELTA) do { _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) ka[ks] += (DELTA); } while (0)
#define V_STEP(DELTA) do { _Pragma("unroll") for (int dt = 0; dt < 8; ++dt) va[dt] += (DELTA); } while (0)

    // mask keys past Sk (last tile only): sacc[kt][qt][i] is key 64 T + 16 kt + kappa(g) + i
#define QK_MASK(T)                                                                            \
    do {                                                                                      \
        if ((int64_t)((T) + 1) * KVT > Sk) {                                                  \
            const int64_t kbase = (int64_t)(T) * KVT + kap;                                   \
            _Pragma("unroll") for (int kt = 0; kt < 4; ++kt)                                  \
                _Pragma("unroll") for (int i = 0; i < 4; ++i)                                 \
                    if (kbase + 16 * kt + i >= Sk) { sacc[kt][0][i] = -INFINITY; sacc[kt][1][i] = -INFINITY; } \
        }                                                                                     \
    } while (0)

    // online softmax: a query's 64 scores live in the 4 lanes (c, g); -> bf16 P^T fragments pb[s][qt]
#define MAX3(A, B, C) __builtin_fmaxf(__builtin_fmaxf(A, B), C)      /* one v_max3_f32 */
#define SOFTMAX_PHASE()                                                                       \
    do {                                                                                      \
        float mx[2];                                                                          \
        {   /* 16 scores -> 1 in 8 three-input steps per query tile; the two tiles' cross-lane steps interleaved (each swap wants \
               two wait states after the VALU write of its operand: the other tile's instruction sits there) */ \
            float a0[2], a2[2];                                                               \
            _Pragma("unroll") for (int qt = 0; qt < 2; ++qt) {                                \
                float b0 = MAX3(sacc[0][qt][0], sacc[0][qt][1], sacc[0][qt][2]);              \
                float b1 = MAX3(sacc[0][qt][3], sacc[1][qt][0], sacc[1][qt][1]);              \
                float b2 = MAX3(sacc[1][qt][2], sacc[1][qt][3], sacc[2][qt][0]);              \
                float b3 = MAX3(sacc[2][qt][1], sacc[2][qt][2], sacc[2][qt][3]);              \
                float b4 = MAX3(sacc[3][qt][0], sacc[3][qt][1], sacc[3][qt][2]);              \
                a0[qt] = MAX3(b0, b1, sacc[3][qt][3]);                                        \
                a2[qt] = MAX3(b2, b3, b4);                                                    \
            }                                                                                 \
            uint32_t u0 = __float_as_uint(fmaxf(a0[0], a2[0])), u1 = __float_as_uint(fmaxf(a0[1], a2[1])); \
            auto s0 = __builtin_amdgcn_permlane32_swap(u0, u0, false, false);      /* lanes l ^ 32 */ \
            auto s1 = __builtin_amdgcn_permlane32_swap(u1, u1, false, false);                 \
            u0 = __float_as_uint(fmaxf(__uint_as_float(s0[0]), __uint_as_float(s0[1])));      \
            u1 = __float_as_uint(fmaxf(__uint_as_float(s1[0]), __uint_as_float(s1[1])));      \
            s0 = __builtin_amdgcn_permlane16_swap(u0, u0, false, false);           /* lanes l ^ 16 */ \
            s1 = __builtin_amdgcn_permlane16_swap(u1, u1, false, false);                      \
            mx[0] = fmaxf(__uint_as_float(s0[0]), __uint_as_float(s0[1]));                    \
            mx[1] = fmaxf(__uint_as_float(s1[0]), __uint_as_float(s1[1]));                    \
        }                                                                                     \
        /* deferred rescale (as attention.hip): keep the old reference max while the new one is within 2^RESCALE_THR of it; \
           m_thr = m_run + the threshold in raw score units, mc = m_run in exp2 units: both kept, updated on a rescale only */ \
        if (__any(mx[0] > m_thr[0] || mx[1] > m_thr[1])) {                                    \
            _Pragma("unroll") for (int qt = 0; qt < 2; ++qt) {                                \
                const float m_new = fmaxf(m_run[qt], mx[qt]);                                 \
                const float alpha = __builtin_amdgcn_exp2f((m_run[qt] - m_new) * scale_log2e); \
                m_run[qt] = m_new;                                                            \
                m_thr[qt] = m_new + thr_raw;                                                  \
                mcs[qt] = m_new * scale_log2e;                                                \
                l_run[qt] *= alpha;                                                           \
                lacc[qt] *= alpha;                                                            \
                _Pragma("unroll") for (int dt = 0; dt < 8; ++dt) acc[dt][qt] *= alpha;        \
            }                                                                                 \
        }                                                                                     \
        _Pragma("unroll") for (int qt = 0; qt < 2; ++qt) {                                    \
            const float mc = mcs[qt];                                                         \
            float ps[2] = {0.f, 0.f};                                                         \
            _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                \
                float p[8];                                                                   \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) {                               \
                    if (ATT16_ABL & 1) p[j] = sacc[2 * s_ + (j >> 2)][qt][j & 3] * scale_log2e - mc;      \
                    else p[j] = __builtin_amdgcn_exp2f(sacc[2 * s_ + (j >> 2)][qt][j & 3] * scale_log2e - mc); \
                    if (!ATT16_LSUM_MFMA) ps[j & 1] += p[j];                                  \
                }                                                                             \
                union { bf16x8_t v; uint32_t u[4]; } cv;                                      \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) cv.u[i] = pack_bf2(p[2 * i], p[2 * i + 1]); \
                pb[s_][qt] = cv.v;                                                            \
            }                                                                                 \
            if (!ATT16_LSUM_MFMA) l_run[qt] += ps[0] + ps[1];                                 \
        }                                                                                     \
    } while (0)

    const float thr_raw = RESCALE_THR / scale_log2e;               // the rescale threshold in raw score units (scale > 0)
    float m_thr[2] = {-INFINITY, -INFINITY}, mcs[2] = {0.f, 0.f};  // m_run + thr_raw, m_run * scale_log2e
    const int nt = (int)((Sk + KVT - 1) / KVT);
    const int grp = wave >> 2;
    // ---- prologue: tiles 0 and 1 landed and visible
    if (nt == 1 && last_rows < KVT) CLAMP_LAST_TILE();
    {
        const char* k0_ = reinterpret_cast<const char*>(Kb);
        const char* v0_ = reinterpret_cast<const char*>(Vb);
        DMA_PIECE_AT(0, k0_, v0_, 0, 0); DMA_PIECE_AT(1, k0_, v0_, 0, 0); DMA_PIECE_AT(2, k0_, v0_, 0, 0); DMA_PIECE_AT(3, k0_, v0_, 0, 0);
    }
    const int64_t ktile_bytes = (int64_t)KVT * ldk * 2, vtile_bytes = (int64_t)KVT * ldv * 2;
    if (nt > 1) {
        if (nt == 2 && last_rows < KVT) CLAMP_LAST_TILE();
        const char* k1_ = reinterpret_cast<const char*>(Kb) + ktile_bytes;
        const char* v1_ = reinterpret_cast<const char*>(Vb) + vtile_bytes;
        DMA_PIECE_AT(0, k1_, v1_, 1, 1); DMA_PIECE_AT(1, k1_, v1_, 1, 1); DMA_PIECE_AT(2, k1_, v1_, 1, 1); DMA_PIECE_AT(3, k1_, v1_, 1, 1);
    }
    DMA_WAIT(0);
    BARRIER();
    if (nt == 3 && last_rows < KVT) CLAMP_LAST_TILE();                // M(0) requests tile 2
    if (!ATT16_BAR1 && grp == 1) BARRIER();                           // G1 runs one interval behind G0
    const char* kreq_p = reinterpret_cast<const char*>(Kb) + (int64_t)min(2, nt - 1) * ktile_bytes;   // tile that M(0) requests
    const char* vreq_p = reinterpret_cast<const char*>(Vb) + (int64_t)min(2, nt - 1) * vtile_bytes;
    int kbuf = 0, vbuf = 0;                        // t % NKB, (t - 1) % NVB: the buffers ka / va point at
    int kreq = 2 % NKB, vreq = 2 % NVB;            // buffers of the tile that M(t) requests (t + 2)
#define REQ_PIECE_NOW(P)                                                                      \
    do {                                                                                      \
        if (!(ATT16_ABL & 4)) DMA_PIECE_AT(P, kreq_p, vreq_p, kreq, vreq);                    \
        FENCE();                                                                              \
    } while (0)
#define REQ_PIECE(P) do { if (!ATT16_DMA_IN_S) REQ_PIECE_NOW(P); } while (0)
    // the ring / pointer / clamp bookkeeping that follows the requests of tile T + 2
#define AFTER_REQUESTS(T)                                                                     \
    do {                                                                                      \
        kreq = kreq == NKB - 1 ? 0 : kreq + 1;                                                \
        vreq = vreq == NVB - 1 ? 0 : vreq + 1;                                                \
        if ((T) + 3 == nt - 1 && last_rows < KVT) CLAMP_LAST_TILE();    /* before the last tile is requested */ \
        if (!(ATT16_ABL & 8) && (T) + 3 <= nt - 1) { kreq_p += ktile_bytes; vreq_p += vtile_bytes; }   /* (ABL 8: every request re-reads tile 2: cache hits) */ \
    } while (0)
    // QK(t) = steps 16..31 (+ the end of M(t)): K addresses move on to K[t+1] once the last K fragment is requested (step 23)
#define M_QK(T)                                                                               \
    do {                                                                                      \
        STEP_QK(16, PEND(16)); STEP_QK(17, PEND(17)); STEP_QK(18, PEND(18)); STEP_QK(19, PEND(19)); \
        REQ_PIECE(2);                                                                         \
        STEP_QK(20, PEND(20)); STEP_QK(21, PEND(21)); STEP_QK(22, PEND(22)); STEP_QK(23, PEND(23)); \
        K_STEP(kbuf == NKB - 1 ? -(NKB - 1) * KBYTES : KBYTES);                               \
        kbuf = kbuf == NKB - 1 ? 0 : kbuf + 1;                                                \
        STEP_QK(24, PEND(24)); STEP_QK(25, PEND(25)); STEP_QK(26, PEND(26)); STEP_QK(27, PEND(27)); \
        REQ_PIECE(3);                                                                         \
        STEP_QK(28, PEND(28)); STEP_QK(29, PEND(29)); STEP_QK(30, PEND(30)); STEP_QK(31, PEND(31)); \
        QK_MASK(T);                                                                           \
        if (!ATT16_DMA_IN_S) AFTER_REQUESTS(T);                                               \
        FENCE();                                                                              \
        __builtin_amdgcn_s_setprio(ATT16_SPRIO);                                              \
        /* tile T+1 has landed: only this segment's pieces of tile T+2 are younger (requests in S: nothing is younger) */ \
        if (!ATT16_BAR1 || grp == 1) {                                                        \
            if (ATT16_DMA_IN_S) DMA_WAIT(0); else DMA_WAIT(4);                                \
            BARRIER();                                                                        \
        }                                                                                     \
    } while (0)
#define S_SEGMENT(T)                                                                          \
    do {                                                                                      \
        if (ATT16_DMA_IN_S) {                                                                 \
            REQ_PIECE_NOW(0); REQ_PIECE_NOW(1); REQ_PIECE_NOW(2); REQ_PIECE_NOW(3);           \
            AFTER_REQUESTS(T);                                                                \
        }                                                                                     \
        if (!(ATT16_ABL & 2)) SOFTMAX_PHASE();                                                \
        else { _Pragma("unroll") for (int kt = 0; kt < 4; ++kt) asm volatile("" :: "v"(sacc[kt][0]), "v"(sacc[kt][1])); } \
        FENCE();                                                                              \
        /* the softmax is complete HERE (a conditional block behind it must not pull half of it down) */ \
        asm volatile("" : "+v"(pb[0][0]), "+v"(pb[0][1]), "+v"(pb[1][0]), "+v"(pb[1][1]), "+v"(l_run[0]), "+v"(l_run[1]) :: "memory"); \
        if (!ATT16_BAR1) BARRIER();                                                           \
        else if (grp == 0) { DMA_WAIT(4); BARRIER(); }                                        \
    } while (0)

    // ---- tile 0: M(0) = QK(0) only.  va points at V[0] from the start (M(0)'s steps 24..31 read V[0]'s first key step)
    __builtin_amdgcn_s_setprio(ATT16_PRIO);
    ISSUE(16); ISSUE(17); ISSUE(18); ISSUE(19); ISSUE(20); ISSUE(21); ISSUE(22); ISSUE(23);
    FENCE();
    REQ_PIECE(0);
    REQ_PIECE(1);
    M_QK(0);
    S_SEGMENT(0);
    for (int t = 1; t < nt; ++t) {
        __builtin_amdgcn_s_setprio(ATT16_PRIO);
        // PV(t-1): key step 0 (fragments 0..7, read at the end of M(t-1)) and key step 1 (8..15, requested here) of V[t-1]
        STEP_PV(0, PEND(0), 1); STEP_PV(1, PEND(1), 1); STEP_PV(2, PEND(2), 1); STEP_PV(3, PEND(3), 1);
        REQ_PIECE(0);
        STEP_PV(4, PEND(4), 1); STEP_PV(5, PEND(5), 1); STEP_PV(6, PEND(6), 1); STEP_PV(7, PEND(7), 1);
        V_STEP(vbuf == NVB - 1 ? -(NVB - 1) * KBYTES : KBYTES);         // V[t-1] -> V[t] (every read of V[t-1] is issued)
        vbuf = vbuf == NVB - 1 ? 0 : vbuf + 1;
        STEP_PV(8, PEND(8), 1); STEP_PV(9, PEND(9), 1); STEP_PV(10, PEND(10), 1); STEP_PV(11, PEND(11), 1);
        REQ_PIECE(1);
        STEP_PV(12, PEND(12), 1); STEP_PV(13, PEND(13), 1); STEP_PV(14, PEND(14), 1); STEP_PV(15, PEND(15), 1);
        M_QK(t);
        S_SEGMENT(t);
    }
    // ---- M(nt) = PV(nt-1): nothing to request beyond fragment 15
    __builtin_amdgcn_s_setprio(ATT16_PRIO);
    STEP_PV(0, 14, 1); STEP_PV(1, 14, 1); STEP_PV(2, 14, 1); STEP_PV(3, 14, 1);
    STEP_PV(4, 14, 1); STEP_PV(5, 14, 1); STEP_PV(6, 14, 1); STEP_PV(7, 14, 1);
    STEP_PV(8, 14, 0); STEP_PV(9, 12, 0); STEP_PV(10, 10, 0); STEP_PV(11, 8, 0);
    STEP_PV(12, 6, 0); STEP_PV(13, 4, 0); STEP_PV(14, 2, 0); STEP_PV(15, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    DMA_WAIT(0);                                   // (the clamped re-requests of the last tile: nothing may land after the epilogue took the LDS)
    if (!ATT16_BAR1 && grp == 0) BARRIER();        // balance G1's extra barrier

    // ---- epilogue: O[q][head*128 + d] = O^T[d][q] / l   (or the un-normalised partial when the keys are split)
    float l_tot[2];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        if (ATT16_LSUM_MFMA) {
            l_tot[qt] = lacc[qt][0];
        } else {
            float l = l_run[qt] + __shfl_xor(l_run[qt], 16, 64);
            l_tot[qt] = l + __shfl_xor(l, 32, 64);
        }
    }
    if (nsplit > 1) {
        const int nbatch = total / (nqb * heads * nsplit);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const int64_t qrow = q0 + 16 * qt + c;
            if (qrow < Sq) {
                const int64_t rowid = ((int64_t)split * nbatch + b) * Sq + qrow;
                float* op = Opart + rowid * ((int64_t)heads * 128) + (int64_t)head * 128 + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) *reinterpret_cast<f32x4_t*>(op + 16 * dt) = acc[dt][qt];
                if (g == 0) {
                    float2* ml = reinterpret_cast<float2*>(MLpart) + rowid * heads + head;
                    *ml = make_float2(m_run[qt], l_tot[qt]);
                }
            }
        }
        return;
    }
    // the wave's 32 x 128 tile through its own 8 KiB of the (now dead) K / V buffers: 16-byte chunk ch of row r at chunk position
    // ch ^ (r & 15); a lane owns 4 consecutive d (8 B) of one query row per (dt, qt); leaves as whole 256-B rows, 16 B per lane
    __syncthreads();                               // group 1's last PV still read a V buffer
    {
        char* ob = smem + wave * 8192;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const float inv = 1.0f / l_tot[qt];
            const int r = 16 * qt + c;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                uint2 o;
                o.x = pack_bf2(acc[dt][qt][0] * inv, acc[dt][qt][1] * inv);
                o.y = pack_bf2(acc[dt][qt][2] * inv, acc[dt][qt][3] * inv);
                *reinterpret_cast<uint2*>(ob + r * 256 + (((2 * dt + (g >> 1)) ^ (r & 15)) << 4) + 8 * (g & 1)) = o;
            }
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
}

// called by attention.hip's launcher (same grid, same arguments as attention_fwd_kernel)
void drn_attention16_launch(const void* q, const void* k, const void* v, void* o, int heads, int64_t Sq, int64_t Sk, int64_t ldq,
                            int64_t ldk, int64_t ldv, int64_t ldo, int64_t bsq, int64_t bsk, int64_t bsv, int64_t bso,
                            float scale_log2e, int nqb, int64_t total, int nsplit, int64_t kv_chunk, float* opart, float* mlpart,
                            hipStream_t st) {
    attention16_fwd_kernel<<<dim3((unsigned)total, 1, 1), dim3(512), 0, st>>>(
        (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, heads, Sq, Sk, ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso,
        scale_log2e, nqb, (int)total, nsplit, kv_chunk, opart, mlpart);
}
