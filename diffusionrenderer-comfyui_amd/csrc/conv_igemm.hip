// Causal 3-D convolution of the Cosmos CV8x8x8 tokenizer as an implicit GEMM on gfx950 MFMA
// ("im2col + MFMA": the im2col matrix is never materialised, its rows are gathered straight into LDS).
//
//   Y[p, n] = epi( bias[n] + sum_{tap, c} X[in(p, tap), c] * Wt[n, tap*C + c] )
//
// Activations are channels-last, X[t][h][w][c] with C contiguous, optionally stored with a zero halo of one
// pixel in H and W (pitches Hp = H + 2, Wp = W + 2, origin (1,1)): spatial zero padding then needs no bounds
// checks at all - every tap of a (1,3,3) kernel is a plain in-range row.  The causal temporal padding
// (frame 0 replicated k_t - 1 times in front, CosmosCausalConv3d) is a clamp of the input frame index.
// Weights are repacked once to [Cout, kT*kH*kW*C] (tap-major, channel-minor) so the GEMM K axis walks one
// tap's channel vector at a time: with C % 64 == 0 every 64-wide K step lies inside ONE tap and the A tile is
// 128 rows x 128 contiguous bytes, gathered by global_load_lds_dwordx4 with a per-lane source address.
//
// The same kernel, with a 1x1x1 "conv" over a compact [M, K] input, is the dense GEMM of the tokenizer's
// mid-block attention (fp32 scaled scores out, then P.V).
//
// Tile / wave layout / LDS swizzle are those of gemm.hip (128x128x64, 4 waves, 16x16x32 bf16 MFMA,
// D = Wfrag x Afrag so a lane owns 4 consecutive output channels of one position).
#include <stdlib.h>
#include "drn_common.h"

#define BM 128
#define BN 128
#define BK 64
#define TILE_BYTES (BM * BK * 2)
#define STAGE_BYTES (2 * TILE_BYTES)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#include "conv_geom.h"

template <bool OUT_F32>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Wt,
                                                            const bf16_t* __restrict__ bias, void* __restrict__ Yv,
                                                            const bf16_t* __restrict__ R, ConvGeom g, int C, int N,
                                                            int64_t ldw, int64_t ldc, int64_t ldr, float alpha) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int64_t M = (int64_t)g.To * g.Ho * g.Wo;
    const int tiles_m = (int)((M + BM - 1) / BM);
    const int tiles_n = (N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    int pid;
    {
        const int bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    // N fastest: the tiles sharing an activation band run together (weights are the small operand here)
    const int tm = pid / tiles_n, tn = pid - tm * tiles_n;
    const int64_t m0 = (int64_t)tm * BM;
    const int n0 = tn * BN;

    // ---- per-lane staging rows: 4 pieces of A (positions) and 4 of W (output channels)
    int64_t a_sp[4];     // element offset of tap (kt,0,0) minus the frame term: (hp*Wp + wp) * C + chunk
    int a_ts[4];         // to*sT - t_off
    const bf16_t* gw[4];
    const int64_t frame_elems = (int64_t)g.Hp * g.Wp * C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int64_t p = m0 + r;
        if (p > M - 1) p = M - 1;
        const int wo = (int)(p % g.Wo);
        const int64_t q = p / g.Wo;
        const int ho = (int)(q % g.Ho);
        const int to = (int)(q / g.Ho);
        const int hp = ho * g.sH - g.pad + g.ih0;
        const int wp = wo * g.sW - g.pad + g.iw0;
        a_sp[i] = ((int64_t)hp * g.Wp + wp) * C + c * 8;
        a_ts[i] = to * g.sT - g.t_off;
        int nr = n0 + r;
        if (nr > N - 1) nr = N - 1;
        gw[i] = Wt + (int64_t)nr * ldw + c * 8;
    }
    const int csteps = C / BK;                  // K steps per tap
    const int nk = g.kT * g.kH * g.kW * csteps;
    auto stage = [&](int kk, int buf) {
        const int tap = kk / csteps;
        const int c0 = (kk - tap * csteps) * BK;
        const int kw = tap % g.kW;
        const int kh = (tap / g.kW) % g.kH;
        const int kt = tap / (g.kW * g.kH);
        const int64_t tap_sp = ((int64_t)kh * g.Wp + kw) * C + c0;
        char* sa = smem + buf * STAGE_BYTES + wave * 4096;
        char* sw = sa + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int ti = a_ts[i] + kt;
            ti = ti < 0 ? 0 : ti;
            const bf16_t* src = X + ti * frame_elems + a_sp[i] + tap_sp;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(sa + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(gw[i] + (int64_t)kk * BK), (lptr_t)(sw + i * 1024), 16, 0, 0);
        }
    };

    const int fr = lane & 15, fq = lane >> 4;
    int offa[4][2], offw[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra = wm * 64 + i * 16 + fr;
        const int rw = wn * 64 + i * 16 + fr;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int c = ks * 4 + fq;
            offa[i][ks] = ra * 128 + ((c ^ ((ra >> 1) & 7)) << 4);
            offw[i][ks] = rw * 128 + ((c ^ ((rw >> 1) & 7)) << 4);
        }
    }

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    stage(0, 0);
    for (int kk = 0; kk < nk; ++kk) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kk + 1 < nk) stage(kk + 1, (kk + 1) & 1);
        const char* sa = smem + (kk & 1) * STAGE_BYTES;
        const char* sw = sa + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(sa + offa[i][ks]);
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(sw + offw[j][ks]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t p = m0 + wm * 64 + i * 16 + fr;
        if (p >= M) continue;
        const int wo = (int)(p % g.Wo);
        const int64_t q = p / g.Wo;
        const int ho = (int)(q % g.Ho);
        const int to = (int)(q / g.Ho);
        const int64_t orow = ((int64_t)to * g.oHp + ho + g.oh0) * g.oWp + wo + g.ow0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + fq * 4;
            if (n >= N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
            if (bias) {
                const uint2 b2 = *reinterpret_cast<const uint2*>(bias + n);
                v[0] += bflo(b2.x); v[1] += bfhi(b2.x); v[2] += bflo(b2.y); v[3] += bfhi(b2.y);
            }
            if (OUT_F32) {
                float4 o = make_float4(v[0] * alpha, v[1] * alpha, v[2] * alpha, v[3] * alpha);
                *reinterpret_cast<float4*>((float*)Yv + orow * ldc + n) = o;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = rbf(v[r]);
                if (R) {
                    const uint2 r2 = *reinterpret_cast<const uint2*>(R + orow * ldr + n);
                    v[0] += bflo(r2.x); v[1] += bfhi(r2.x); v[2] += bflo(r2.y); v[3] += bfhi(r2.y);
                }
                uint2 o;
                o.x = pack_bf2(v[0], v[1]);
                o.y = pack_bf2(v[2], v[3]);
                *reinterpret_cast<uint2*>((bf16_t*)Yv + orow * ldc + n) = o;
            }
        }
    }
}

// conv256s.hip: the streamed 256x256 kernel for the big convolutions
bool drn_conv256s_ok(const ConvGeom& g, int C, int N, int64_t ldc, int64_t ldr, bool has_residual, bool out_f32, const void* y,
                     const void* residual);
int drn_conv256s_launch(const void* x, const void* w, const void* bias, void* y, const void* residual, const ConvGeom& g, int C,
                        int N, int64_t ldw, int64_t ldc, int64_t ldr, void* stream);

// drn_conv_force_tile: -1 automatic, 0 the 128x128 kernel always, 1 the 256x256 streamed kernel wherever it can run (also
// below its size threshold).  drn_conv_last_tile: which kernel the last drn_conv3d_igemm call launched (tests / tools).
static int g_conv_force = -1, g_conv_last = -1;
extern "C" void drn_conv_force_tile(int tile) { g_conv_force = tile; }
extern "C" int drn_conv_last_tile(void) { return g_conv_last; }

extern "C" int drn_conv3d_igemm(const void* x, const void* w, const void* bias, void* y, const void* residual,
                                int T, int H, int W, int C, int in_halo, int N, int kT, int kH, int kW, int sT, int sH,
                                int sW, int pad, int t_off, int To, int Ho, int Wo, int out_halo, int64_t ldc,
                                int64_t ldr, int out_f32, float alpha, void* stream) {
    DRN_CHECK_ARG(x && w && y && T > 0 && H > 0 && W > 0 && C > 0 && C % BK == 0 && N > 0 && N % 4 == 0);
    DRN_CHECK_ARG(kT >= 1 && kH >= 1 && kW >= 1 && sT >= 1 && sH >= 1 && sW >= 1 && (pad == 0 || pad == 1));
    DRN_CHECK_ARG(To > 0 && Ho > 0 && Wo > 0 && ldc >= N && ldc % 4 == 0 && (!residual || (ldr >= N && ldr % 4 == 0)));
    DRN_CHECK_ARG(!(out_f32 && residual));
    DRN_CHECK_ARG(in_halo == 0 || in_halo == 1);
    // every tap of every output position must stay inside the stored (halo-padded) input image
    const int hmax = (Ho - 1) * sH - pad + (kH - 1) + in_halo, wmax = (Wo - 1) * sW - pad + (kW - 1) + in_halo;
    DRN_CHECK_ARG(hmax < H + 2 * in_halo && wmax < W + 2 * in_halo && in_halo - pad >= 0);
    DRN_CHECK_ARG((To - 1) * sT + (kT - 1) - t_off < T);
    DRN_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)y & 7) == 0);
    ConvGeom g;
    g.To = To; g.Ho = Ho; g.Wo = Wo;
    g.T = T; g.Hp = H + 2 * in_halo; g.Wp = W + 2 * in_halo; g.ih0 = in_halo; g.iw0 = in_halo;
    g.kT = kT; g.kH = kH; g.kW = kW; g.sT = sT; g.sH = sH; g.sW = sW; g.t_off = t_off; g.pad = pad;
    g.oHp = Ho + 2 * out_halo; g.oWp = Wo + 2 * out_halo; g.oh0 = out_halo; g.ow0 = out_halo;
    const int64_t M = (int64_t)To * Ho * Wo;
    const int64_t tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    DRN_CHECK_ARG(tiles < (1ll << 31));
    const int64_t ldw = (int64_t)kT * kH * kW * C;
    if (g_conv_force != 0) {
        static int mode = -1;
        if (mode < 0) {
            const char* e = getenv("DRN_CONV256");             // 0: off (A/B runs)
            mode = (e && e[0] == '0') ? 0 : 1;
        }
        const bool can = drn_conv256s_ok(g, C, N, ldc, ldr, residual != nullptr, out_f32 != 0, y, residual);
        const bool small_ok = g_conv_force == 1 && !out_f32 && N % 256 == 0 && C % 64 == 0 && ldc % 8 == 0 &&
                              ((uintptr_t)y & 15) == 0 && (int64_t)T * g.Hp * g.Wp * C * 2 < (1ll << 32);
        if ((mode == 1 && can) || small_ok) {
            g_conv_last = 1;
            return drn_conv256s_launch(x, w, bias, y, residual, g, C, N, ldw, ldc, ldr, stream);
        }
    }
    g_conv_last = 0;
    dim3 grid((unsigned)tiles), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (out_f32)
        conv_igemm_kernel<true><<<grid, block, 0, st>>>((const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)bias, y,
                                                        nullptr, g, C, N, ldw, ldc, ldr, alpha);
    else
        conv_igemm_kernel<false><<<grid, block, 0, st>>>((const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)bias, y,
                                                         (const bf16_t*)residual, g, C, N, ldw, ldc, ldr, alpha);
    return drn_launch_status();
}
