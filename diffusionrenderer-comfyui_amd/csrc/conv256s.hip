// Causal 3-D convolution as an implicit GEMM on the streamed 256x256x64 schedule of gemm256s.hip (round 2).
//
// conv_igemm.hip's 128x128 kernel waits for its one prefetched K step at the top of every iteration (vmcnt(0) + __syncthreads):
// 0.88-0.98 PFLOP/s on the tokenizer's big (1,3,3) convolutions (73 728-552 960 positions x 256-512 channels, K = 2304-4608),
// which carry two thirds of the tokenizer's conv time at the headline clip.  Those are plain MFMA-bound GEMMs whose A rows are
// gathered: this kernel runs them through gemm256s_core.h - same continuous MFMA stream, same LDS layout, same wait counts -
// with a DMA macro that gathers a half-tile's 128 positions of one tap's 64-channel slice.
//
//   * A rows: position p of the tile -> (to, ho, wo) once per tile; per lane a 32-bit byte offset of the position's (kt = 0,
//     kh = 0, kw = 0) corner inside input frame max(to*sT - t_off + kt, 0) (the causal clamp; stepped by one frame when the
//     walk reaches the next kt).  A K step adds a SCALAR tap offset ((kh*Wp + kw)*C + c0) to the base pointer:
//     `global_load_lds saddr + voffset`, no vector arithmetic per K step.  (Launcher: the input tensor is < 4 GiB.)
//   * the K index handed to the DMA macro is ignored: the tap walk (c0 -> kw -> kh -> kt) is a scalar cursor advanced once per
//     K step, in the order the main loop requests K steps (0, 1, 2, ...; past the end it stays on the last step, as the GEMM's
//     re-requests do).
//   * epilogue: + bias, round, + residual (loaded a half-tile at a time, before that half's stores - it may alias the
//     output), 16-byte stores through the permlane16 exchange, halo-padded channels-last output rows.
// Accumulation order per output element = conv_igemm.hip's (K ascending, 32 per MFMA): results are bit-identical.
#include <stdlib.h>
#include "drn_common.h"
#include "conv_geom.h"
#define BETWEEN_PROLOGUE_STEPS() ADVANCE()
#include "gemm256s_core.h"

__global__ __launch_bounds__(512, 2) void conv256s_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wt,
                                                          const bf16_t* __restrict__ bias, bf16_t* Y, const bf16_t* R,
                                                          ConvGeom g, int C, int N, int64_t ldw, int64_t ldc, int64_t ldr,
                                                          int GROUP) {
    const int64_t M = (int64_t)g.To * g.Ho * g.Wo;
    const int64_t K = (int64_t)g.kT * g.kH * g.kW * C;
    THREAD_SETUP();
    (void)gsrc;
    int64_t m0, n0;
    tile_of(blockIdx.x, nwg, tiles_m, tiles_n, GROUP, m0, n0);

    // ---- per-lane gather state of this wave's 2 pieces of each A half
    uint32_t a_off[2][2];      // bytes: corner of the position's receptive field in input frame max(a_ts + cur_kt, 0) + this
                               // lane's 16-byte chunk; stepped by one frame when the cursor's kt advances (ADVANCE)
    int a_ts[2][2];            // first input frame: to*sT - t_off (negative = the causal padding: clamped to frame 0)
    uint32_t voffw[2];
    const uint32_t frame_bytes = (uint32_t)g.Hp * (uint32_t)g.Wp * (uint32_t)C * 2u;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int rl = p * 8 + (lane >> 3);                              // row inside this wave's 16 rows of a half-tile
        const int c = (lane & 7) ^ ((rl >> 1) & 7);                      // (wave * 16 is a multiple of 16: no part in the swizzle)
        voffw[p] = (uint32_t)((rl * ldw + c * 8) * 2);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int64_t pos = m0 + h * 128 + wave * 16 + rl;
            if (pos > M - 1) pos = M - 1;
            const uint32_t pu = (uint32_t)pos;                            // launcher: M < 2^31
            const uint32_t q = pu / (uint32_t)g.Wo;
            const int wo = (int)(pu - q * (uint32_t)g.Wo);
            const uint32_t to = q / (uint32_t)g.Ho;
            const int ho = (int)(q - to * (uint32_t)g.Ho);
            const int hp = ho * g.sH - g.pad + g.ih0;
            const int wp = wo * g.sW - g.pad + g.iw0;
            a_ts[h][p] = (int)to * g.sT - g.t_off;
            a_off[h][p] = ((uint32_t)(hp * g.Wp + wp) * (uint32_t)C + (uint32_t)(c * 8)) * 2u +
                          (uint32_t)max(a_ts[h][p], 0) * frame_bytes;
        }
    }
    const char* sw_tile = reinterpret_cast<const char*>(Wt + (n0 + wave * 16) * ldw);

    // ---- the tap cursor (scalar): K step cur_k = channels [cur_c0, cur_c0 + 64) of tap (cur_kt, cur_kh, cur_kw)
    int cur_k = 0, cur_c0 = 0, cur_kw = 0, cur_kh = 0, cur_kt = 0;
    uint32_t tap_bytes = 0;
#define ADVANCE()                                                                                                      \
    do {                                                                                                               \
        if (cur_k + 1 < nk) {                                                                                          \
            ++cur_k;                                                                                                   \
            cur_c0 += BK;                                                                                              \
            if (cur_c0 == C) {                                                                                         \
                cur_c0 = 0;                                                                                            \
                if (++cur_kw == g.kW) {                                                                                \
                    cur_kw = 0;                                                                                        \
                    if (++cur_kh == g.kH) {                                                                            \
                        cur_kh = 0;                                                                                    \
                        ++cur_kt;           /* next input frame, unless still inside the causal padding */             \
                        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                               \
                            _Pragma("unroll") for (int p_ = 0; p_ < 2; ++p_)                                           \
                                a_off[h_][p_] += (a_ts[h_][p_] + cur_kt > 0) ? frame_bytes : 0u;                       \
                    }                                                                                                  \
                }                                                                                                      \
            }                                                                                                          \
            tap_bytes = ((uint32_t)(cur_kh * g.Wp + cur_kw) * (uint32_t)C + (uint32_t)cur_c0) * 2u;                      \
        }                                                                                                              \
    } while (0)
#undef DMA
#define DMA(H, KD, S)                                                                                                  \
    do {                                                                                                               \
        char* dst_ = smem + (S) * STAGE_BYTES + (H) * HALF_BYTES + dma_off;                                            \
        if ((H) < 2) {                                                                                                 \
            const char* xa_ = reinterpret_cast<const char*>(X) + tap_bytes;                                            \
            __builtin_amdgcn_global_load_lds((gptr_t)(xa_ + a_off[(H) & 1][0]), (lptr_t)dst_, 16, 0, 0);               \
            __builtin_amdgcn_global_load_lds((gptr_t)(xa_ + a_off[(H) & 1][1]), (lptr_t)(dst_ + 1024), 16, 0, 0);      \
        } else {                                                                                                       \
            const char* sb_ = sw_tile + (((int64_t)((((H) - 2) & 1) * 128) * ldw + (int64_t)cur_k * BK) << 1);         \
            __builtin_amdgcn_global_load_lds((gptr_t)(sb_ + voffw[0]), (lptr_t)dst_, 16, 0, 0);                        \
            __builtin_amdgcn_global_load_lds((gptr_t)(sb_ + voffw[1]), (lptr_t)(dst_ + 1024), 16, 0, 0);               \
        }                                                                                                              \
    } while (0)

    ZERO_ACC();
    const int k_second = 0;            // (unused by this DMA macro: the cursor decides)
    PROLOGUE();                        // K step 0, ADVANCE, K step 1
    ADVANCE();
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        KSTEP(0, wx, wy, k_second, 10, 10, 10, 10);
        ADVANCE();
        KSTEP(1, wy, wx, k_second, 10, 10, 10, 10);
        ADVANCE();
    }
    if (kt < nk) KSTEP(0, wx, wy, k_second, 10, 10, 10, 10);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the re-requests past the last K step

    // ---- epilogue
    int frl = fr, fql = fq;
    asm volatile("" : "+v"(frl), "+v"(fql));                             // (address arithmetic stays below the K loop)
    uint2 b2[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int64_t n = n0 + j * 128 + wc * 32 + nt * 16 + fql * 4;
            b2[j][nt] = bias ? *reinterpret_cast<const uint2*>(bias + n) : make_uint2(0u, 0u);
        }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int64_t orow[4];
        bool ok[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int64_t p_raw = m0 + i * 128 + wr * 64 + mt * 16 + frl;
            ok[mt] = p_raw < M;
            const uint32_t pu = (uint32_t)(ok[mt] ? p_raw : M - 1);
            const uint32_t q = pu / (uint32_t)g.Wo;
            const int wo = (int)(pu - q * (uint32_t)g.Wo);
            const uint32_t to = q / (uint32_t)g.Ho;
            const int ho = (int)(q - to * (uint32_t)g.Ho);
            orow[mt] = ((int64_t)to * g.oHp + ho + g.oh0) * g.oWp + wo + g.ow0;
        }
        uint2 r2[4][2][2];
        if (R) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const int64_t n = n0 + j * 128 + wc * 32 + nt * 16 + fql * 4;
                        r2[mt][j][nt] = *reinterpret_cast<const uint2*>(R + orow[mt] * ldr + n);
                    }
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint2 o[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = acc[i][mt][j][nt][r];
                    if (bias) {
                        const uint2 bb = b2[j][nt];
                        v[0] += bflo(bb.x); v[1] += bfhi(bb.x); v[2] += bflo(bb.y); v[3] += bfhi(bb.y);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = rbf(v[r]);
                    if (R) {
                        const uint2 rr = r2[mt][j][nt];
                        v[0] += bflo(rr.x); v[1] += bfhi(rr.x); v[2] += bflo(rr.y); v[3] += bfhi(rr.y);
                    }
                    o[nt].x = pack_bf2(v[0], v[1]);
                    o[nt].y = pack_bf2(v[2], v[3]);
                }
                const auto sx = __builtin_amdgcn_permlane16_swap(o[0].x, o[1].x, false, false);
                const auto sy = __builtin_amdgcn_permlane16_swap(o[0].y, o[1].y, false, false);
                const int64_t n8 = n0 + j * 128 + wc * 32 + (fql & 1) * 16 + (fql >> 1) * 8;
                if (ok[mt])
                    *reinterpret_cast<uint4*>(Y + orow[mt] * ldc + n8) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
            }
    }
}

// 1 when this launch can take the 256^2 streamed kernel (else conv_igemm.hip's 128^2 kernel runs it)
bool drn_conv256s_ok(const ConvGeom& g, int C, int N, int64_t ldc, int64_t ldr, bool has_residual, bool out_f32,
                     const void* y, const void* residual) {
    const int64_t M = (int64_t)g.To * g.Ho * g.Wo;
    const int64_t in_bytes = (int64_t)g.T * g.Hp * g.Wp * C * 2;
    if (out_f32 || N % TB != 0 || C % BK != 0) return false;
    if (M >= (1ll << 31) || in_bytes >= (1ll << 32)) return false;
    if (ldc % 8 != 0 || ((uintptr_t)y & 15) != 0) return false;
    if (has_residual && (ldr % 4 != 0 || ((uintptr_t)residual & 7) != 0)) return false;
    // a 256-row tile owns a CU for the whole K loop: below ~3/4 of a round of workgroups the 128^2 kernel (2 per CU) wins
    const int64_t tiles = ((M + TB - 1) / TB) * (N / TB);
    return tiles >= 192;
}

int drn_conv256s_launch(const void* x, const void* w, const void* bias, void* y, const void* residual, const ConvGeom& g, int C,
                        int N, int64_t ldw, int64_t ldc, int64_t ldr, void* stream) {
    static bool configured = false;
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv256s_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
        if (e != hipSuccess) return (int)e;
        configured = true;
    }
    const int64_t M = (int64_t)g.To * g.Ho * g.Wo;
    const int64_t tiles = ((M + TB - 1) / TB) * (N / TB);
    if (tiles >= (1ll << 31)) return DRN_EINVAL;
    conv256s_kernel<<<dim3((unsigned)tiles), dim3(512), 2 * STAGE_BYTES, (hipStream_t)stream>>>(
        (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)bias, (bf16_t*)y, (const bf16_t*)residual, g, C, N, ldw, ldc, ldr, 4);
    return drn_launch_status();
}
