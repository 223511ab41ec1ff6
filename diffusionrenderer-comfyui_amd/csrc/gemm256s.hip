// 256x256x64 bf16 GEMM, "streamed" schedule:  C = epi(A[M,K] . W[N,K]^T)
//
// Measured on the first-generation kernel of round 1 (a ping-pong of two wave groups, removed in round 3; QKV shape, MI355X): MFMAs + barriers alone 0.92 ms, its memory
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
// A persistent form (one workgroup per CU walking its tiles, DMA stream crossing the tile boundary) was built and measured in
// round 2 (out-proj +2.6 %, MLP +-0.3 %, QKV -3.9 %; it also spilled 8 B per lane) and removed in round 3: no net gain.  Launch +
// cold prologue between tiles cost the one-workgroup-per-tile kernel nothing measurable (the next workgroup's prologue already
// overlaps the previous one's store drain); non-temporal C stores (G256S_NT): +-0.5 %.
//
// (c) `nt` (aux = 2) or `sc0` (aux = 1) cache-policy bits on the W or A LDS-DMA loads: 0.5-5 % SLOWER on every shape - both
// operands are re-read from L2 / the Infinity Cache by the other tiles of the band (MI355X_MICROARCH.md 'nt-weights' says the same).
// (d) an L2 'touch' - one global_load_dword per wave and K step over the 128-byte lines that the DMA will ask for 3-6 K steps
// later (hand-over waits 12 / 12 / 11 / 11), meant to turn the ~19 % of operand bytes that miss the XCD's L2 into hits: 15-27 %
// SLOWER on every shape.  A 64-line load costs the CU's vector-memory path more than the two 1 KiB DMA pieces it stands beside;
// that path (one DMA piece per ~37 cycles per CU) is what the operand side of this kernel is bound by.
//
// Tile, LDS layout (2 stages x [A0 A1 W0 W1] x 16 KiB, 128-B rows, chunk ^ ((row>>1)&7), swizzle on the DMA source address),
// wave -> quadrant map and blocked operand layouts: gemm256s_core.h.
#include <stdlib.h>
#include "drn_common.h"

// G256S_ABL: timing-only ablations (results WRONG; shipped with 0): 8 = no epilogue math / stores, 64 = no gate / residual loads,
// 128 = no C stores (16 / 32: the hand-over waits, gemm256s_core.h)
#ifndef G256S_ABL
#define G256S_ABL 0
#endif
// G256S_NT: C tiles stored non-temporal (a round of 256 tiles is 32 MB, the eight L2s hold 32 MB)
#ifndef G256S_NT
#define G256S_NT 0
#endif
#include "gemm256s_core.h"

#define EPI_PARTIAL 3      // internal: split-K slice (blockIdx.y = slice), fp32 tile to the workspace, no epilogue

// epilogue: column tiles nt = 0 / 1 exchanged between lane rows fq = 2k / 2k+1, 16-byte stores (16 per lane)
// PRE (gated-residual epilogue of whole tiles): the residual tile is already in LDS - row r of the tile at r * 512 B, its 16-byte
// chunk c at chunk position c ^ (r & 15) - requested by the last two K steps in place of their re-requests (see the kernel); so
// are the tile's 256 gate values (gate_lds: this wave's own copy)
template <int EPI, bool PRE = false>
static __device__ __forceinline__ void store_tile(f32x4_t (&acc)[2][4][2][2], bf16_t* C, int64_t M, int64_t N, int64_t ldc,
                                                  const bf16_t* __restrict__ gate, const bf16_t* R, int64_t ldr, int64_t rpb,
                                                  int cbc, int64_t cbs, int64_t m0, int64_t n0, int wr, int wc, int fr, int fq,
                                                  const char* res_lds = nullptr, const char* gate_lds = nullptr) {
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
    if (EPI == EPI_PARTIAL) {
        // split-K slice: the fp32 tile as it is, [split][M][N] (gemm_splitk_epilogue_kernel sums the slices and applies the
        // epilogue); a lane owns 4 consecutive columns of a row: 16-byte stores
        float* part = reinterpret_cast<float*>(C);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int64_t m = m0 + i * 128 + wr * 64 + mt * 16 + fr;
                if (m >= M) continue;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const int64_t n = n0 + j * 128 + wc * 32 + nt * 16 + fq * 4;
                        if (n < N) *reinterpret_cast<f32x4_t*>(part + m * ldc + n) = acc[i][mt][j][nt];
                    }
            }
        return;
    }
    const int64_t c_tile_off = (n0 >> cbc) * cbs + (n0 & ((1ll << cbc) - 1)) - n0;
    // every address below derives from these two: defined HERE, so that none of the address arithmetic is hoisted out of the
    // persistent kernel's tile loop into registers that stay live through the K loop (it spilled there)
    asm volatile("" : "+v"(fr), "+v"(fq));
    const int64_t b_tile = (EPI == DRN_EPI_GATE_RES) ? (int64_t)((uint32_t)m0 / (uint32_t)rpb) : 0;      // launcher: M < 2^31
    const bool one_clip = m0 + TB <= (b_tile + 1) * rpb;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        // gate / residual of this half first, all 32 loads in flight at once: issued one (mt, j) unit at a time in front of that
        // unit's store they were 16 dependent memory round trips per tile (the residual may alias C - in-place x += g * y - so
        // hipcc cannot move a load above the previous unit's store by itself).  Every element is loaded and stored by lanes of
        // the same wave (the permlane16 partner), and a wave's loads of a half all precede its stores of that half.
        uint2 g2[4][2][2], r2[4][2][2];
        if (EPI == DRN_EPI_GATE_RES && (G256S_ABL & 64)) {           // timing only: no gate / residual loads
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) g2[mt][j][nt] = r2[mt][j][nt] = make_uint2(0x3f803f80u, 0x3f803f80u);
        } else if (EPI == DRN_EPI_GATE_RES && PRE) {
            // whole tile, residual in LDS: (chunk ^ fr) for nt = 1 is that of nt = 0 with bit 1 flipped (bit 1 of wc*4 + (fq>>1) is 0)
            const uint32_t ra = (uint32_t)((wr * 64 + fr) * 512 + (((wc * 4 + (fq >> 1)) ^ fr) << 4) + (fq & 1) * 8);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        r2[mt][j][nt] = *reinterpret_cast<const uint2*>(res_lds + i * 65536 + mt * 8192 + j * 256 + (ra ^ (nt * 32)));
                        if (mt == 0) {                               // (a whole tile lies inside one clip: launcher)
                            g2[0][j][nt] = *reinterpret_cast<const uint2*>(gate_lds + (j * 128 + wc * 32 + nt * 16 + fq * 4) * 2);
                        } else {
                            g2[mt][j][nt] = g2[0][j][nt];
                        }
                    }
        } else if (EPI == DRN_EPI_GATE_RES) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int64_t m_raw = m0 + i * 128 + wr * 64 + mt * 16 + fr;
                const int64_t m = m_raw < M ? m_raw : M - 1;
                // the tile's clip: one scalar division per tile unless the tile straddles two clips (64-bit vector divisions
                // are ~100 instructions each)
                const int64_t b = one_clip ? b_tile : (int64_t)((uint32_t)m / (uint32_t)rpb);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const int64_t n = n0 + j * 128 + wc * 32 + nt * 16 + fq * 4;
                        g2[mt][j][nt] = *reinterpret_cast<const uint2*>(gate + b * N + n);
                        r2[mt][j][nt] = *reinterpret_cast<const uint2*>(R + m * ldr + n);
                    }
            }
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int64_t m_raw = m0 + i * 128 + wr * 64 + mt * 16 + fr;
            const bool m_ok = m_raw < M;
            const int64_t m = m_ok ? m_raw : M - 1;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint2 o[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = rbf(acc[i][mt][j][nt][r]);
                    if (EPI == DRN_EPI_GELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = gelu_erf_fast(v[r]);
                    } else if (EPI == DRN_EPI_GATE_RES) {
                        const uint2 gg = g2[mt][j][nt], rr = r2[mt][j][nt];
                        const float g[4] = {bflo(gg.x), bfhi(gg.x), bflo(gg.y), bfhi(gg.y)};
                        const float x[4] = {bflo(rr.x), bfhi(rr.x), bflo(rr.y), bfhi(rr.y)};
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = x[r] + rbf(g[r] * v[r]);
                    }
                    o[nt].x = pack_bf2(v[0], v[1]);
                    o[nt].y = pack_bf2(v[2], v[3]);
                }
                const auto sx = __builtin_amdgcn_permlane16_swap(o[0].x, o[1].x, false, false);
                const auto sy = __builtin_amdgcn_permlane16_swap(o[0].y, o[1].y, false, false);
                const int64_t n8 = n0 + j * 128 + wc * 32 + (fq & 1) * 16 + (fq >> 1) * 8;
                if (G256S_ABL & 128) {                              // timing only: no stores
                    asm volatile("" :: "v"(sx[0]), "v"(sy[0]), "v"(sx[1]), "v"(sy[1]));
                } else if (m_ok && n8 < N) {
                    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
                    const u32x4_t val = {sx[0], sy[0], sx[1], sy[1]};
                    u32x4_t* dst = reinterpret_cast<u32x4_t*>(C + m * ldc + n8 + c_tile_off);
                    if (G256S_NT) __builtin_nontemporal_store(val, dst);
                    else *dst = val;
                }
            }
        }
    }
}
#define STORES_PER_TILE 16          // vector-memory stores per lane in store_tile (whole tiles): part of the vmcnt arithmetic

template <int EPI, bool PRE = false>
__global__ __launch_bounds__(512, 2) void gemm256s_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                          bf16_t* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                                                          int64_t ldw, int64_t ldc, const bf16_t* __restrict__ gate,
                                                          const bf16_t* R, int64_t ldr, int64_t rpb, int GROUP,
                                                          int abc, int64_t abs_, int cbc, int64_t cbs) {
    if (EPI == EPI_PARTIAL) {
        // slice blockIdx.y of gridDim.y: K / gridDim.y columns of A and W (plain layouts), its own fp32 slab of the workspace
        K /= gridDim.y;
        A += (int64_t)blockIdx.y * K;
        W += (int64_t)blockIdx.y * K;
        C = reinterpret_cast<bf16_t*>(reinterpret_cast<float*>(C) + (int64_t)blockIdx.y * M * ldc);
    }
    THREAD_SETUP();
    int64_t m0, n0;
    tile_of(blockIdx.x, nwg, tiles_m, tiles_n, GROUP, m0, n0);
    const int k_second = min(1, nk - 1);
    int kt = 0;
    if (PRE) {
        // Gated-residual epilogue of a whole tile (launcher: M, N multiples of 256, every tile inside one clip, nk even and >= 4).
        // The last two K steps have nothing left to request, and the eight 16 KiB regions they free (stage 0 during step nk-2,
        // stage 1 during step nk-1) are exactly one 256 x 256 bf16 tile: they request the RESIDUAL tile instead - same two 1 KiB
        // pieces per wave and region, same wait counts - so that the epilogue finds it in LDS instead of starting 2 x 16 row-strided
        // 8-byte loads per lane after the loop (measured: those loads were 11 % of the out-projection, 3.8 % of MLP-down;
        // `G256S_ABL=64`).  Region q = 4 * stage + half-tile id holds rows 32 q .. 32 q + 31 of the tile (512 B each), a piece = 2
        // rows, lane l = row l >> 5, chunk position l & 31.
        // Whole tiles need no row clamp, so the operand pieces are addressed as a uniform half-tile base + ONE 32-bit lane offset per
        // operand and piece (4 registers instead of the 8 lane pointers of the general kernel: the loop has none to spare).
        (void)gsrc;
        const char* abase = reinterpret_cast<const char*>(A + m0 * lda);
        const char* wbase = reinterpret_cast<const char*>(W + n0 * ldw);
        uint32_t voff[2][2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r = (wave * 2 + p) * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            voff[0][p] = (uint32_t)((r * (int)lda + c * 8) * 2);
            voff[1][p] = (uint32_t)((r * (int)ldw + c * 8) * 2);
        }
#undef DMA
#define DMA(H, KD, S)                                                                                                  \
    do {                                                                                                               \
        const int kt_ = (int)(KD);                                                                                     \
        char* dst_ = smem + H_OFF(S, H) + dma_off;                                                                     \
        const char* sb_ = (H) < 2 ? abase + (((int64_t)((H) & 1) * 128 * lda + A_KOFF(kt_)) << 1)                      \
                                  : wbase + (((int64_t)((H) & 1) * 128 * ldw + (int64_t)kt_ * BK) << 1);               \
        __builtin_amdgcn_global_load_lds((gptr_t)(sb_ + voff[(H) >> 1][0]), (lptr_t)dst_, 16, 0, 0);                   \
        __builtin_amdgcn_global_load_lds((gptr_t)(sb_ + voff[(H) >> 1][1]), (lptr_t)(dst_ + 1024), 16, 0, 0);          \
    } while (0)
        ZERO_ACC();
        PROLOGUE();
        for (; kt + 2 < nk; kt += 2) {
            KSTEP(0, wx, wy, kt + 2, 10, 10, 10, 10);
            KSTEP(1, wy, wx, kt + 3, 10, 10, 10, 10);
        }
        const char* rres = reinterpret_cast<const char*>(R + m0 * ldr + n0);
        uint32_t roff[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int rr = wave * 4 + p * 2 + (lane >> 5);           // row inside the region (its low 4 bits = those of the tile row)
            roff[p] = (uint32_t)((rr * (int)ldr + (((lane & 31) ^ (rr & 15)) << 3)) * 2);
        }
#undef DMA
#define DMA(H, KD, S)                                                                                                  \
    do {                                                                                                               \
        const char* rb_ = rres + (int64_t)(((S) * 4 + (H)) * 32) * ldr * 2;                                            \
        char* dst_ = smem + ((S) * 4 + (H)) * HALF_BYTES + dma_off;                                                    \
        __builtin_amdgcn_global_load_lds((gptr_t)(rb_ + roff[0]), (lptr_t)dst_, 16, 0, 0);                             \
        __builtin_amdgcn_global_load_lds((gptr_t)(rb_ + roff[1]), (lptr_t)(dst_ + 1024), 16, 0, 0);                    \
    } while (0)
        KSTEP(0, wx, wy, 0, 10, 10, 10, 10);
        // the tile's 256 gate values (512 B; every wave fetches its own copy, lanes 32..63 a second one: a piece is 1 KiB) ride along
        // with the first request of the last step: one more piece in flight, so that step's hand-over waits count 11
        const char* gbase = reinterpret_cast<const char*>(gate + (int64_t)((uint32_t)m0 / (uint32_t)rpb) * N + n0);
        const uint32_t goff = (uint32_t)((lane & 31) * 16);
        char* gate_lds = smem + 2 * STAGE_BYTES + wave * 1024;
#undef DMA
#define DMA(H, KD, S)                                                                                                  \
    do {                                                                                                               \
        const char* rb_ = rres + (int64_t)(((S) * 4 + (H)) * 32) * ldr * 2;                                            \
        char* dst_ = smem + ((S) * 4 + (H)) * HALF_BYTES + dma_off;                                                    \
        if ((H) == H_W0) __builtin_amdgcn_global_load_lds((gptr_t)(gbase + goff), (lptr_t)gate_lds, 16, 0, 0);         \
        __builtin_amdgcn_global_load_lds((gptr_t)(rb_ + roff[0]), (lptr_t)dst_, 16, 0, 0);                             \
        __builtin_amdgcn_global_load_lds((gptr_t)(rb_ + roff[1]), (lptr_t)(dst_ + 1024), 16, 0, 0);                    \
    } while (0)
        KSTEP(1, wy, wx, 0, 11, 11, 11, 11);
#undef DMA
#define DMA(H, KD, S) GEMM_DMA(H, KD, S)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the residual tile has landed (this wave's pieces) ...
        __builtin_amdgcn_s_barrier();                                  // ... and everybody else's
        store_tile<EPI, true>(acc, C, M, N, ldc, gate, R, ldr, rpb, cbc, cbs, m0, n0, wr, wc, fr, fq, smem, gate_lds);
        return;
    }
    SET_SRC(m0, n0);
    ZERO_ACC();
    PROLOGUE();
    // steps past the end re-request the last one into a region nobody reads: the wait counts stay uniform
    for (; kt + 1 < nk; kt += 2) {
        KSTEP(0, wx, wy, min(kt + 2, nk - 1), 10, 10, 10, 10);
        KSTEP(1, wy, wx, min(kt + 3, nk - 1), 10, 10, 10, 10);
    }
    if (kt < nk) KSTEP(0, wx, wy, nk - 1, 10, 10, 10, 10);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the re-requests past the last K step: nothing lands after exit
    store_tile<EPI>(acc, C, M, N, ldc, gate, R, ldr, rpb, cbc, cbs, m0, n0, wr, wc, fr, fq);
}

// ---- weight-streaming form of the split-K slice kernel (few tokens: the weights come cold from HBM, once).
// With 2 + 2 stages the DMA of a W half-tile is ~1.25 K steps ahead of its first read: 40 KB of W per CU in flight, which at an
// HBM round trip under load of ~3 us is 24 GB/s per CU - the slices ran latency-bound (43 us for 16 K steps).  The A slice of a
// few-token product is shared by every workgroup of its K slice and comes from L2; only W needs depth.  So the whole 160 KiB
// of LDS becomes A: 2 stages x 32 KiB (as before) + W: 3 stages x 32 KiB, and W is requested THREE steps ahead:
//     group 1: DMA W0(k+3)   group 2: DMA A0(k+2)   group 3: DMA W1(k+3)   group 4: DMA A1(k+2)
// In-order vmcnt arithmetic (calls of 2 pieces each, queue per step Q(k) = [W0(k+3) A0(k+2) W1(k+3) A1(k+2)]):
//     after group 1 needs A1(k)   = last of Q(k-2):   Q(k-1) + 1 call  younger = 10 pieces
//     after group 2 needs W0(k+1) = first of Q(k-2):  3 + 4 + 2 calls  younger = 18
//     after group 3 needs A0(k+1) = second of Q(k-1): 2 + 3 calls      younger = 10
//     after group 4 needs W1(k+1) = third of Q(k-2):  1 + 4 + 4 calls  younger = 18
// The prologue issues W0(0) W1(0), then the virtual steps Q(-2) = [W0(1) A0(0) W1(1) A1(0)] and Q(-1) = [W0(2) A0(1) W1(2)
// A1(1)], so that the counts hold from step 0 on.  Same MFMA order per output element as gemm256s_kernel<EPI_PARTIAL>:
// bit-identical slices.
#undef A_OFF
#undef W_OFF
#define A_OFF(S, I) ((S) * 2 * HALF_BYTES + (I) * HALF_BYTES)                       // 2 stages x [A0 A1]
#define W_OFF(S, J) (4 * HALF_BYTES + (S) * 2 * HALF_BYTES + (J) * HALF_BYTES)      // 3 stages x [W0 W1]
#define WS_LDS_BYTES (10 * HALF_BYTES)                                              // 160 KiB
__global__ __launch_bounds__(512, 2) void gemm256w_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                          float* __restrict__ partial, int64_t M, int64_t N, int64_t K,
                                                          int64_t lda, int64_t ldw, int GROUP) {
    const int abc = 62, cbc = 62;
    const int64_t abs_ = 0, cbs = 0;
    K /= gridDim.y;
    A += (int64_t)blockIdx.y * K;
    W += (int64_t)blockIdx.y * K;
    float* part = partial + (int64_t)blockIdx.y * M * N;
    THREAD_SETUP();
    int64_t m0, n0;
    tile_of(blockIdx.x, nwg, tiles_m, tiles_n, GROUP, m0, n0);
    SET_SRC(m0, n0);
    ZERO_ACC();
#define KCL(k) min((int)(k), nk - 1)
    // ---- prologue (see above)
    DMA(H_W0, 0, 0); DMA(H_W1, 0, 0);
    DMA(H_W0, KCL(1), 1); DMA(H_A0, 0, 0); DMA(H_W1, KCL(1), 1); DMA(H_A1, 0, 0);
    DMA(H_W0, KCL(2), 2); DMA(H_A0, KCL(1), 1); DMA(H_W1, KCL(2), 2); DMA(H_A1, KCL(1), 1);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                  // W0(0), W1(0), W0(1), A0(0) have landed
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
    __builtin_amdgcn_s_barrier();                                      // every wave holds W0(0) / A0(0): their regions are free
    FENCE();
    // ---- K loop: the A stage (k % 2) and the roles of the two W fragment buffers are literals of a body unrolled by two; the W
    //      stage (k % 3) is a scalar that walks 0 1 2 (a body unrolled by six with early exits made hipcc merge six fragment
    //      histories at the loop exit: 1100 spilled registers)
    int sw = 0;
#define WSTEP(KT, SA, WA, WB)                                                                                         \
    do {                                                                                                              \
        const int swn_ = sw == 2 ? 0 : sw + 1;                                                                        \
        KSTEP2(SA, (SA) ^ 1, sw, swn_, WA, WB, KCL((KT) + 2), KCL((KT) + 3), 10, 18, 10, 18);                         \
        sw = swn_;                                                                                                    \
    } while (0)
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        WSTEP(kt, 0, wx, wy);
        WSTEP(kt + 1, 1, wy, wx);
    }
    if (kt < nk) WSTEP(kt, 0, wx, wy);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the re-requests past the last K step
    store_tile<EPI_PARTIAL>(acc, reinterpret_cast<bf16_t*>(part), M, N, /*ldc=*/N, nullptr, nullptr, 0, M, cbc, cbs, m0, n0, wr, wc,
                            fr, fq);
}
#undef WSTEP
#undef KCL
#undef A_OFF
#undef W_OFF
#define A_OFF(S, I) ((S) * STAGE_BYTES + (I) * HALF_BYTES)
#define W_OFF(S, J) ((S) * STAGE_BYTES + (2 + (J)) * HALF_BYTES)

#ifndef G256S_PRE_DEFAULT
#define G256S_PRE_DEFAULT 1
#endif
static int g_res_prefetch = G256S_PRE_DEFAULT;     // drn_gemm_force_res_prefetch (tests, A/B runs; tools/build_variants.py)
extern "C" int drn_gemm_force_res_prefetch(int on) {
    const int was = g_res_prefetch;
    if (on >= 0) g_res_prefetch = on ? 1 : 0;
    return was;
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
    if (tiles >= (1ll << 31) || M >= (1ll << 31)) return DRN_EINVAL;
    if (rpb > M || rpb <= 0) rpb = M;                      // (32-bit row / rows_per_batch arithmetic in the epilogue)
    static int group = 0;
    if (group == 0) {
        const char* e = getenv("DRN_GEMM_GROUP");          // tile-rows per L2 band (A/B experiments)
        group = e ? atoi(e) : 4;
        if (group < 1) group = 4;
    }
    if constexpr (EPI == DRN_EPI_GATE_RES) {
        // whole tiles, each inside one clip, an even number (>= 4) of K steps, 16-byte residual rows: the residual tile comes in
        // through LDS under the last two K steps (same bits: only where the epilogue reads it from changes)
        static int pre = -1;
        if (pre < 0) {
            const char* e = getenv("DRN_GEMM_RES_PREFETCH");   // 0: the epilogue loads the residual itself (A/B runs, tests)
            pre = (e && e[0] == '0') ? 0 : 1;
            if (pre && hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256s_kernel<EPI, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES + 8192) != hipSuccess) {
                (void)hipGetLastError();
                pre = 0;
            }
        }
        const int64_t nk = K / BK;
        if (pre && g_res_prefetch && M % TB == 0 && N % TB == 0 && rpb % TB == 0 && nk >= 4 && nk % 2 == 0 && residual && gate &&
            ldr % 8 == 0 && ((uintptr_t)residual & 15) == 0 && ((uintptr_t)gate & 15) == 0 && 128 * ldr * 2 < (1ll << 31) && 128 * lda * 2 < (1ll << 31) &&
            128 * ldw * 2 < (1ll << 31)) {
            gemm256s_kernel<EPI, true><<<dim3((unsigned)tiles), dim3(512), 2 * STAGE_BYTES + 8192, st>>>(
                (const bf16_t*)A, (const bf16_t*)W, (bf16_t*)C, M, N, K, lda, ldw, ldc, (const bf16_t*)gate, (const bf16_t*)residual,
                ldr, rpb, group, (int)blk[0], blk[1], (int)blk[2], blk[3]);
            return drn_launch_status();
        }
    }
    gemm256s_kernel<EPI><<<dim3((unsigned)tiles), dim3(512), 2 * STAGE_BYTES, st>>>(
        (const bf16_t*)A, (const bf16_t*)W, (bf16_t*)C, M, N, K, lda, ldw, ldc, (const bf16_t*)gate, (const bf16_t*)residual,
        ldr, rpb, group, (int)blk[0], blk[1], (int)blk[2], blk[3]);
    return drn_launch_status();
}

// split-K slices of the streamed kernel (small M: gemm.hip drn_gemm_bf16_splitk): partial[split][M][N] fp32.
// Measured at S = 256 (cfg 1): QKV / MLP-up / MLP-down 52 -> 43 us per launch (48 x 4, 64 x 4, 16 x 16 workgroups of 16 K steps),
// 7.39 -> 6.97 ms per step.  A workgroup takes in 1 MB in those 43 us = 24 GB/s per CU against ~80 GB/s in the big GEMMs: the
// weights come cold from HBM and only ~1.25 K steps (80 KB per CU, 15-20 MB chip-wide) are in flight.  Tried: the weights
// stored K-step-major ([K/64][N][64]: every half-tile one contiguous 16 KiB run instead of 128 runs of 128 B at an 8 KiB
// stride) - bit-identical, NOT faster (48.5 vs 50.7 us QKV incl. the reduce): DRAM page locality is not the limit, bytes in
// flight are.  A deeper W ring needs a different LDS budget (A 2 x 32 KiB + W 3 x 32 KiB = the whole 160 KiB).
int drn_gemm256s_partial(const void* A, const void* W, float* partial, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                         int splits, void* stream, bool wring_ok) {
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256s_kernel<EPI_PARTIAL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    const int64_t tiles = ((M + TB - 1) / TB) * ((N + TB - 1) / TB);
    if (tiles >= 65536 || splits < 1 || splits > 64 || (K / BK) % splits != 0 || M >= (1ll << 31)) return DRN_EINVAL;
    static int wring = -1;
    if (wring < 0) {
        const char* e = getenv("DRN_SPLITK_WRING");        // 0: the 2 + 2 stage kernel (A/B runs)
        wring = (e && e[0] == '0') ? 0 : 1;
        if (wring && hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256w_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES) != hipSuccess) {
            (void)hipGetLastError();
            wring = 0;                                      // (a device that does not grant 160 KiB per workgroup)
        }
    }
    if (wring && wring_ok && K / splits / BK >= 3) {
        gemm256w_kernel<<<dim3((unsigned)tiles, (unsigned)splits), dim3(512), WS_LDS_BYTES, (hipStream_t)stream>>>(
            (const bf16_t*)A, (const bf16_t*)W, partial, M, N, K, lda, ldw, 4);
        return drn_launch_status();
    }
    static const int64_t plain[4] = {62, 0, 62, 0};
    gemm256s_kernel<EPI_PARTIAL><<<dim3((unsigned)tiles, (unsigned)splits), dim3(512), 2 * STAGE_BYTES, (hipStream_t)stream>>>(
        (const bf16_t*)A, (const bf16_t*)W, (bf16_t*)partial, M, N, K, lda, ldw, /*ldc=*/N, nullptr, nullptr, 0, M, 4,
        (int)plain[0], plain[1], (int)plain[2], plain[3]);
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
