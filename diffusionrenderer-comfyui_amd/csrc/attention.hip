// Non-causal flash attention forward for head_dim 128 on gfx950 (replaces F.scaled_dot_product_attention +
// the sbhd<->bhsd permutes + the head flatten, CleanGeneralDIT.py:181-203, :299-304).
//
// Workgroup = 8 waves = 256 query rows (32 per wave), KV tile = 64 keys, K and V tiles double-buffered in LDS
// (register-staged: the next tile's global loads are issued before the current tile's MFMAs and written to LDS
// after them, one barrier per tile).
//
// Per wave and KV tile (v_mfma_f32_32x32x16_bf16 only):
//   S^T[key][q]  = K . Q^T          A = K rows (ds_read_b128, XOR-swizzled 256-B rows), B = Q (registers)
//       -> the query sits on the LANE, its 64 scores in 2x16 registers of lanes l and l^32:
//          row max / row sum are in-register reductions + one cross-lane exchange.
//   O^T[d][q]   += V^T . P^T        B = P^T taken straight from the S^T accumulator registers (cvt to bf16, no
//       lane movement; the k order inside a 16-step is permuted: element j of lane half h is key
//       16s + 8(j>>2) + 4h + (j&3)), A = V^T read with ds_read_b64_tr_b16 in that same key order.
//   O^T keeps the query on the lane too, so the online-softmax rescale is one per-lane scalar.
//
// Softmax in fp32 (exp2 with the scale folded in), P rounded to bf16 for the PV MFMA, row sum from the fp32 P.
#include "drn_common.h"

// EXPV: ablation builds behind the stall budget quoted in DESIGN.md (1: no barrier, 2: no K/V staging, 4: no exp) -
// timing only, results are wrong; the shipped library is built with 0.
#ifndef EXPV
#define EXPV 0
#endif
#define QROWS 256        // query rows per workgroup
#define KVT 64           // keys per tile
#define KBYTES (KVT * 256)
#ifndef RESCALE_THR
#define RESCALE_THR 6.0f   // log2 units
#endif

typedef __attribute__((address_space(3))) bf16x4_t* lds_b64_ptr;

__device__ __forceinline__ bf16x4_t ds_read_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64_ptr)(p));
}

__global__ __launch_bounds__(512, 2) void attention_fwd_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kp, const bf16_t* __restrict__ Vp, bf16_t* __restrict__ O,
    int heads, int64_t Sq, int64_t Sk_total, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t bsq, int64_t bsk,
    int64_t bsv, int64_t bso, float scale_log2e, int nqb, int total, int nsplit, int64_t kv_chunk,
    float* __restrict__ Opart, float* __restrict__ MLpart) {
    __shared__ __attribute__((aligned(1024))) char smem[5 * KBYTES];   // K0 K1 V0 V1 V2

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
        // make the Q loads complete HERE: otherwise hipcc keeps a decreasing vmcnt ladder in front of the QK^T MFMAs of
        // every iteration (first-iteration hazard), which also drains the next tile's K/V loads far too early
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(qf[ks]));
    }

    // ---- staging map: 1024 16-byte chunks per tile, two per thread per operand
    int st_row[2], st_c[2], st_koff[2], st_voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = tid + 512 * i;
        st_row[i] = id >> 4;
        st_c[i] = id & 15;
        st_koff[i] = st_row[i] * 256 + ((st_c[i] ^ (st_row[i] & 15)) << 4);
        st_voff[i] = st_row[i] * 256 + ((((st_c[i] >> 2) ^ (st_row[i] & 3)) << 6) | ((st_c[i] & 3) << 4));
    }
    // named registers (no arrays / lambdas: keeps the in-flight tile in VGPRs, not in a promoted alloca)
    u32x4_t kreg0, kreg1, vreg0, vreg1;
    const bf16_t* kp0 = Kb + (int64_t)st_row[0] * ldk + st_c[0] * 8;      // advanced by one tile per iteration
    const bf16_t* kp1 = Kb + (int64_t)st_row[1] * ldk + st_c[1] * 8;
    const bf16_t* vp0 = Vb + (int64_t)st_row[0] * ldv + st_c[0] * 8;
    const bf16_t* vp1 = Vb + (int64_t)st_row[1] * ldv + st_c[1] * 8;
    const int64_t kstep = (int64_t)KVT * ldk, vstep = (int64_t)KVT * ldv;
#define LOAD_TILE(KV0)                                                                        \
    do {                                                                                      \
        if ((KV0) + KVT <= Sk) {                                                              \
            kreg0 = *reinterpret_cast<const u32x4_t*>(kp0);                                   \
            vreg0 = *reinterpret_cast<const u32x4_t*>(vp0);                                   \
            kreg1 = *reinterpret_cast<const u32x4_t*>(kp1);                                   \
            vreg1 = *reinterpret_cast<const u32x4_t*>(vp1);                                   \
        } else { /* last, partial tile: clamp rows (masked below) */                          \
            int64_t r0_ = (KV0) + st_row[0], r1_ = (KV0) + st_row[1];                         \
            if (r0_ > Sk - 1) r0_ = Sk - 1;                                                   \
            if (r1_ > Sk - 1) r1_ = Sk - 1;                                                   \
            kreg0 = *reinterpret_cast<const u32x4_t*>(Kb + r0_ * ldk + st_c[0] * 8);          \
            vreg0 = *reinterpret_cast<const u32x4_t*>(Vb + r0_ * ldv + st_c[0] * 8);          \
            kreg1 = *reinterpret_cast<const u32x4_t*>(Kb + r1_ * ldk + st_c[1] * 8);          \
            vreg1 = *reinterpret_cast<const u32x4_t*>(Vb + r1_ * ldv + st_c[1] * 8);          \
        }                                                                                     \
        kp0 += kstep; kp1 += kstep; vp0 += vstep; vp1 += vstep;                               \
    } while (0)
#define WRITE_TILE(KBUF, VBUF)                                                                \
    do {                                                                                      \
        char* ks__ = smem + (KBUF) * KBYTES;                                                  \
        char* vs__ = smem + (2 + (VBUF)) * KBYTES;                                            \
        *reinterpret_cast<u32x4_t*>(ks__ + st_koff[0]) = kreg0;                               \
        *reinterpret_cast<u32x4_t*>(vs__ + st_voff[0]) = vreg0;                               \
        *reinterpret_cast<u32x4_t*>(ks__ + st_koff[1]) = kreg1;                               \
        *reinterpret_cast<u32x4_t*>(vs__ + st_voff[1]) = vreg1;                               \
    } while (0)

    // ---- LDS read offsets
    // K (A operand of S^T): row = 32*kt2 + lr, chunk = 2*ks + lh
    int koff[2];
#pragma unroll
    for (int kt2 = 0; kt2 < 2; ++kt2) koff[kt2] = (32 * kt2 + lr) * 256;
    const int kx = lr & 15;          // swizzle key: (row & 15) == (lr & 15) for both sub-tiles
    // V^T (A operand of O^T) via ds_read_b64_tr_b16: 16-lane group g reads a 4-key x 16-d block;
    // lane i of the group supplies row (i>>2), columns 4*(i&3)..+3 and receives column i.
    const int vi = lane & 15;
    const int vq = vi >> 2, vp = vi & 3;
    const int vdh = (lane >> 4) & 1;          // which 16-wide half of the 32-d tile
    // byte column inside the 256-B row for d-tile dt: (32*dt + 16*vdh + 4*vp) * 2 = 64*dt + 32*vdh + 8*vp
    // swizzled: segment (dt ^ (key&3)) * 64 + 32*vdh + 8*vp ; key&3 == vq for every block (block bases are % 4 == 0)
    int vcol[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vcol[dt] = ((dt ^ vq) << 6) + 32 * vdh + 8 * vp;
    const int vrow0 = (4 * lh + vq) * 256;    // + (32*kt2 + 16*s [+8]) * 256

    f32x16_t acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    f32x16_t s[2];
    bf16x8_t pb[2][2];

    // S^T = K . Q^T from K buffer KBUF, then mask keys past Sk (last tile only)
#define QK_PHASE(KBUF, T)                                                                                          \
    do {                                                                                                           \
        const char* ks_ = smem + (KBUF) * KBYTES;                                                                  \
        _Pragma("unroll") for (int kt2 = 0; kt2 < 2; ++kt2) {                                                      \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) s[kt2][r] = 0.f;                                        \
            _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) {                                                     \
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(ks_ + koff[kt2] + (((2 * ks + lh) ^ kx) << 4)); \
                s[kt2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kt2], 0, 0, 0);                     \
            }                                                                                                      \
        }                                                                                                          \
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
                p[r] = (EXPV & 4) ? (s[kt2][r] * scale_log2e - mc) : __builtin_amdgcn_exp2f(s[kt2][r] * scale_log2e - mc);                                       \
                ps4[r & 3] += p[r];                                                                                \
            }                                                                                                      \
            _Pragma("unroll") for (int sidx = 0; sidx < 2; ++sidx) {                                               \
                union { bf16x8_t v; uint32_t u[4]; } cv;                                                           \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) cv.u[i] = pack_bf2(p[8 * sidx + 2 * i], p[8 * sidx + 2 * i + 1]); \
                pb[kt2][sidx] = cv.v;                                                                              \
            }                                                                                                      \
        }                                                                                                          \
        l_run += (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);                                                            \
    } while (0)

    // O^T += V^T . P^T from V buffer VBUF
#define PV_PHASE(VBUF)                                                                                             \
    do {                                                                                                           \
        const char* vs_ = smem + (2 + (VBUF)) * KBYTES;                                                            \
        _Pragma("unroll") for (int kt2 = 0; kt2 < 2; ++kt2)                                                        \
            _Pragma("unroll") for (int sidx = 0; sidx < 2; ++sidx) {                                               \
                const char* vb = vs_ + vrow0 + (32 * kt2 + 16 * sidx) * 256;                                       \
                _Pragma("unroll") for (int dt = 0; dt < 4; ++dt) {                                                 \
                    const bf16x4_t lo = ds_read_tr16(vb + vcol[dt]);                                               \
                    const bf16x4_t hi = ds_read_tr16(vb + 8 * 256 + vcol[dt]);                                     \
                    bf16x8_t vf;                                                                                   \
                    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];                                    \
                    vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];                                    \
                    acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[kt2][sidx], acc[dt], 0, 0, 0);        \
                }                                                                                                  \
            }                                                                                                      \
    } while (0)

    const int nt = (int)((Sk + KVT - 1) / KVT);
    LOAD_TILE((int64_t)0);
    WRITE_TILE(0, 0);
    __syncthreads();

    // The two waves that share a SIMD (w and w+4) run the SAME tile between two barriers but in rotated order:
    //   group 0:  QK(t) . softmax(t) . PV(t)            group 1:  PV(t-1) . QK(t) . softmax(t)
    // so after the barrier one wave's softmax (VALU) overlaps the other's MFMA phases instead of both contending for
    // the matrix pipe and then both for the VALU.  V is triple-buffered (group 1 still reads V[t-1] while V[t+1] lands).
    int vcur = 0;                                  // t % 3
    if (wave < 4) {
        for (int t = 0; t < nt; ++t) {
            const int vnext = vcur == 2 ? 0 : vcur + 1;
            if (!(EXPV & 2) && t + 1 < nt) LOAD_TILE((int64_t)(t + 1) * KVT);
            QK_PHASE(t & 1, t);
            SOFTMAX_PHASE();
            PV_PHASE(vcur);
            if (!(EXPV & 2) && t + 1 < nt) WRITE_TILE((t + 1) & 1, vnext);
            if (!(EXPV & 1)) __syncthreads();
            vcur = vnext;
        }
    } else {
        int vprev = 2;
        for (int t = 0; t < nt; ++t) {
            const int vnext = vcur == 2 ? 0 : vcur + 1;
            if (!(EXPV & 2) && t + 1 < nt) LOAD_TILE((int64_t)(t + 1) * KVT);
            if (t > 0) PV_PHASE(vprev);
            QK_PHASE(t & 1, t);
            SOFTMAX_PHASE();
            if (!(EXPV & 2) && t + 1 < nt) WRITE_TILE((t + 1) & 1, vnext);
            if (!(EXPV & 1)) __syncthreads();
            vprev = vcur;
            vcur = vnext;
        }
        PV_PHASE(vprev);
    }

    // ---- epilogue: O[q][head*128 + d] = O^T[d][q] / l   (or the un-normalised partial when the keys are split)
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const int64_t qrow = q0 + lr;
    if (qrow < Sq) {
        if (nsplit > 1) {
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
        } else {
            const float inv = 1.0f / l_tot;
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
    }
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

static int attention_launch(const void* q, const void* k, const void* v, void* o, int batch, int heads, int64_t Sq, int64_t Sk,
                            int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int64_t bsq, int64_t bsk, int64_t bsv,
                            int64_t bso, float scale, int nsplit, void* workspace, void* stream) {
    DRN_CHECK_ARG(q && k && v && o && batch > 0 && heads > 0 && Sq >= 0 && Sk > 0 && nsplit >= 1);
    DRN_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0);
    DRN_CHECK_ARG(bsq % 8 == 0 && bsk % 8 == 0 && bsv % 8 == 0 && bso % 4 == 0);
    DRN_CHECK_ARG(((uintptr_t)q & 15) == 0 && ((uintptr_t)k & 15) == 0 && ((uintptr_t)v & 15) == 0 && ((uintptr_t)o & 7) == 0);
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
    // gridDim.y carries the batch count for the partial layout (blocks are indexed by x only)
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
