// HBM-bound kernels of the Cosmos CV8x8x8 tokenizer (channels-last activations with a 1-pixel zero halo):
// per-frame GroupNorm(1 group)+SiLU, 2-level 3-D Haar patch / unpatch, hybrid down/up-sampling helpers,
// the mid block's causal temporal attention, row softmax and small layout moves.
// Restates (unpinned, see oracle/vae_oracle.py) diffusers' autoencoder_kl_cosmos.py as reached from CleanVAE.py:50-60.
#pragma clang fp contract(off)
#include "drn_common.h"

// ------------------------------------------------------------------------------------------------ GroupNorm
// stats: per frame sum / sum of squares over the whole stored frame (halo is zero, so it does not contribute).
// Accumulated in fp64: the sums are then independent (to ~1e-16) of how the elements are split over threads, blocks and -
// when a frame is cut into row bands over several GPUs - ranks, so mean / rstd come out the same fp32 numbers either way.
__global__ __launch_bounds__(256) void gn_stats_kernel(const bf16_t* __restrict__ x, double* __restrict__ part,
                                                       int64_t frame_elems, int nblk) {
    const int f = blockIdx.y;
    const bf16_t* xf = x + (int64_t)f * frame_elems;
    const int64_t nch = frame_elems / 8;
    double s = 0.0, q = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nch; i += (int64_t)nblk * 256) {
        float v[8];
        unpack8(*reinterpret_cast<const uint4*>(xf + i * 8), v);
        float s8 = 0.f, q8 = 0.f;                 // 8 bf16 values: their fp32 sum of squares is exact enough to be order-free
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s8 += v[j];
            q8 += v[j] * v[j];
        }
        s += (double)s8;
        q += (double)q8;
    }
    __shared__ double sh[2][256];
    sh[0][threadIdx.x] = s;
    sh[1][threadIdx.x] = q;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + st];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[((int64_t)f * nblk + blockIdx.x) * 2 + 0] = sh[0][0];
        part[((int64_t)f * nblk + blockIdx.x) * 2 + 1] = sh[1][0];
    }
}

// apply: y = [silu](bf16((x - mean) * rstd * gamma + beta)) on the interior; the output halo is left untouched (zero).
// Round 3: a thread keeps ONE 8-channel chunk (gamma / beta in registers) and walks rows h and pixels w with plain adds - the
// first version spent two 64-bit divisions per 16 bytes on (pixel, chunk) and an IEEE division + expf per element on the SiLU,
// i.e. it was VALU-bound at 3.3 TB/s.  sigmoid = v_rcp(1 + v_exp(-o log2 e)): 1 ulp of fp32 in front of the bf16 rounding.
template <bool ROWS>
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x, const double* __restrict__ part,
                                                       const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta,
                                                       bf16_t* __restrict__ y, int H, int W, int C, int halo, int nblk,
                                                       float eps, int silu, float cnt) {
    const int f = blockIdx.y;
    double sd = 0.0, qd = 0.0;
    for (int i = 0; i < nblk; ++i) {
        sd += part[((int64_t)f * nblk + i) * 2 + 0];
        qd += part[((int64_t)f * nblk + i) * 2 + 1];
    }
    const float s = (float)sd, q = (float)qd;
    const float mean = s / cnt;
    const float var = fmaxf(q / cnt - mean * mean, 0.f);
    const float rstd = 1.0f / sqrtf(var + eps);
    const int Hp = H + 2 * halo, Wp = W + 2 * halo;
    const int cch = C / 8;
    auto one = [&](int64_t off, const float* gm, const float* bt) {
        float v[8];
        unpack8(*reinterpret_cast<const uint4*>(x + off), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float o = rbf(((v[j] - mean) * rstd) * gm[j] + bt[j]);
            if (silu) o = o * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * o));
            v[j] = o;
        }
        *reinterpret_cast<uint4*>(y + off) = pack8(v);
    };
    if (ROWS) {
        // 256 % cch == 0 (launcher): thread -> (pixel lane tid / cch, chunk tid % cch); a block walks rows h = blockIdx.x, + gridDim.x ..
        const int c = (int)(threadIdx.x % (unsigned)cch) * 8, wl = (int)(threadIdx.x / (unsigned)cch), wstep = 256 / cch;
        float gm[8], bt[8];
        unpack8(*reinterpret_cast<const uint4*>(gamma + c), gm);
        unpack8(*reinterpret_cast<const uint4*>(beta + c), bt);
        for (int h = blockIdx.x; h < H; h += gridDim.x) {
            const int64_t row = (((int64_t)f * Hp + h + halo) * Wp + halo) * C + c;
            for (int w = wl; w < W; w += wstep) one(row + (int64_t)w * C, gm, bt);
        }
        return;
    }
    const int64_t total = (int64_t)H * W * cch;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % cch) * 8;
        const int64_t pw = i / cch;
        const int w = (int)(pw % W), h = (int)(pw / W);
        const int64_t off = (((int64_t)f * Hp + h + halo) * Wp + w + halo) * C + c;
        float gm[8], bt[8];
        unpack8(*reinterpret_cast<const uint4*>(gamma + c), gm);
        unpack8(*reinterpret_cast<const uint4*>(beta + c), bt);
        one(off, gm, bt);
    }
}

// rows-per-block form when the 8-channel chunks of a pixel divide a workgroup (C = 128, 256, 512, ...) and a frame has rows enough
static void gn_apply_launch(const bf16_t* x, const double* part, const bf16_t* gamma, const bf16_t* beta, bf16_t* y, int frames, int H,
                            int W, int C, int halo, int nparts, float eps, int silu, float count, hipStream_t st) {
    const int cch = C / 8;
    if (cch <= 256 && 256 % cch == 0) {
        gn_apply_kernel<true><<<dim3((unsigned)H, frames), dim3(256), 0, st>>>(x, part, gamma, beta, y, H, W, C, halo, nparts, eps, silu,
                                                                             count);
        return;
    }
    int64_t blocks = ((int64_t)H * W * cch + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    gn_apply_kernel<false><<<dim3((unsigned)blocks, frames), dim3(256), 0, st>>>(x, part, gamma, beta, y, H, W, C, halo, nparts, eps,
                                                                                silu, count);
}

#define GN_NBLK 64
extern "C" int drn_groupnorm_silu(const void* x, const void* gamma, const void* beta, void* y, void* workspace,
                                  int frames, int H, int W, int C, int halo, float eps, int silu, void* stream) {
    DRN_CHECK_ARG(x && gamma && beta && y && workspace && frames > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0);
    DRN_CHECK_ARG(halo == 0 || halo == 1);
    DRN_CHECK_ARG(frames <= 65535);
    const int64_t frame_elems = (int64_t)(H + 2 * halo) * (W + 2 * halo) * C;
    hipStream_t st = (hipStream_t)stream;
    gn_stats_kernel<<<dim3(GN_NBLK, frames), dim3(256), 0, st>>>((const bf16_t*)x, (double*)workspace, frame_elems, GN_NBLK);
    gn_apply_launch((const bf16_t*)x, (const double*)workspace, (const bf16_t*)gamma, (const bf16_t*)beta, (bf16_t*)y, frames, H, W, C,
                    halo, GN_NBLK, eps, silu, (float)H * (float)W * (float)C, st);
    return drn_launch_status();
}

// the two halves on their own, for a frame that is spread over several GPUs (row bands): every rank sums its band, the
// partial sums of all ranks are gathered, and every rank normalises its band with the whole frame's statistics
extern "C" int drn_groupnorm_stats(const void* x, void* part, int frames, int H, int W, int C, int halo, void* stream) {
    DRN_CHECK_ARG(x && part && frames > 0 && frames <= 65535 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && (halo == 0 || halo == 1));
    const int64_t frame_elems = (int64_t)(H + 2 * halo) * (W + 2 * halo) * C;
    gn_stats_kernel<<<dim3(GN_NBLK, frames), dim3(256), 0, (hipStream_t)stream>>>((const bf16_t*)x, (double*)part, frame_elems, GN_NBLK);
    return drn_launch_status();
}

extern "C" int drn_groupnorm_apply(const void* x, const void* part, int nparts, float count, const void* gamma, const void* beta,
                                   void* y, int frames, int H, int W, int C, int halo, float eps, int silu, void* stream) {
    DRN_CHECK_ARG(x && part && gamma && beta && y && nparts > 0 && count > 0.f && frames > 0 && frames <= 65535);
    DRN_CHECK_ARG(H > 0 && W > 0 && C > 0 && C % 8 == 0 && (halo == 0 || halo == 1));
    gn_apply_launch((const bf16_t*)x, (const double*)part, (const bf16_t*)gamma, (const bf16_t*)beta, (bf16_t*)y, frames, H, W, C, halo,
                    nparts, eps, silu, count, (hipStream_t)stream);
    return drn_launch_status();
}
extern "C" int64_t drn_groupnorm_workspace_bytes(int frames) { return (int64_t)frames * GN_NBLK * 2 * sizeof(double); }

// ------------------------------------------------------------------------------------------------ Haar
#define HW_ 0.70703125f          // bf16(0.7071067811865476): the wavelet taps are cast to the activation dtype
#define SQRT8 2.8284271247461903f

// one analysis level on a 2x2x2 block a[t][h][w] -> 8 sub-bands [lll llh lhl lhh hll hlh hhl hhh] (t,h,w order),
// each separable stage rounded to bf16 as the three grouped conv3d calls do, then / sqrt(8)
__device__ __forceinline__ void haar_fwd_block(const float a[2][2][2], float o[8]) {
    float t[2][2][2];   // [band_t][h][w]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            t[0][h][w] = rbf(HW_ * a[0][h][w] + HW_ * a[1][h][w]);
            t[1][h][w] = rbf(HW_ * a[0][h][w] + (-HW_) * a[1][h][w]);
        }
    float u[2][2][2];   // [band_t][band_h][w]
#pragma unroll
    for (int bt = 0; bt < 2; ++bt)
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            u[bt][0][w] = rbf(HW_ * t[bt][0][w] + HW_ * t[bt][1][w]);
            u[bt][1][w] = rbf(HW_ * t[bt][0][w] + (-HW_) * t[bt][1][w]);
        }
#pragma unroll
    for (int bt = 0; bt < 2; ++bt)
#pragma unroll
        for (int bh = 0; bh < 2; ++bh) {
            o[4 * bt + 2 * bh + 0] = rbf(rbf(HW_ * u[bt][bh][0] + HW_ * u[bt][bh][1]) / SQRT8);
            o[4 * bt + 2 * bh + 1] = rbf(rbf(HW_ * u[bt][bh][0] + (-HW_) * u[bt][bh][1]) / SQRT8);
        }
}

// one synthesis level: 8 sub-bands -> 2x2x2 block (W, then H, then T transposed convs, each product and each sum rounded)
__device__ __forceinline__ void haar_inv_block(const float s[8], float a[2][2][2]) {
    float u[2][2][2];   // [band_t][band_h][w]
#pragma unroll
    for (int bt = 0; bt < 2; ++bt)
#pragma unroll
        for (int bh = 0; bh < 2; ++bh) {
            const float l = s[4 * bt + 2 * bh + 0], h = s[4 * bt + 2 * bh + 1];
            u[bt][bh][0] = rbf(rbf(HW_ * h) + rbf(HW_ * l));
            u[bt][bh][1] = rbf(rbf((-HW_) * h) + rbf(HW_ * l));
        }
    float t[2][2][2];   // [band_t][h][w]
#pragma unroll
    for (int bt = 0; bt < 2; ++bt)
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            t[bt][0][w] = rbf(rbf(HW_ * u[bt][1][w]) + rbf(HW_ * u[bt][0][w]));
            t[bt][1][w] = rbf(rbf((-HW_) * u[bt][1][w]) + rbf(HW_ * u[bt][0][w]));
        }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            a[0][h][w] = rbf(rbf(rbf(HW_ * t[1][h][w]) + rbf(HW_ * t[0][h][w])) * SQRT8);
            a[1][h][w] = rbf(rbf(rbf((-HW_) * t[1][h][w]) + rbf(HW_ * t[0][h][w])) * SQRT8);
        }
}

// video [Cin][T][H][W] planar bf16 -> patches [Tp][Hq+2h][Wq+2h][64*Cin] channels-last (2 levels, patch 4).
// Frame 0 is repeated 4x in front (CosmosPatchEmbed3d._haar): padded index tau reads frame max(tau - 3, 0).
__global__ __launch_bounds__(256) void haar_patch_kernel(const bf16_t* __restrict__ v, bf16_t* __restrict__ out, int Cin,
                                                         int T, int H, int W, int Tp, int halo) {
    const int Hq = H / 4, Wq = W / 4;
    const int64_t total = (int64_t)Tp * Hq * Wq * Cin;
    const int Cout = 64 * Cin;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cin);
        int64_t r = i / Cin;
        const int wq = (int)(r % Wq); r /= Wq;
        const int hq = (int)(r % Hq);
        const int tq = (int)(r / Hq);
        float l1[8][2][2][2];       // level-1 sub-band sb1 at level-1 position (t,h,w) in the 2x2x2 neighbourhood
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                for (int pw = 0; pw < 2; ++pw) {
                    float a[2][2][2], o[8];
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        int tau = 4 * tq + 2 * pt + dt - 3;
                        tau = tau < 0 ? 0 : tau;
#pragma unroll
                        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                            for (int dw = 0; dw < 2; ++dw)
                                a[dt][dh][dw] = bf2f(v[(((int64_t)c * T + tau) * H + 4 * hq + 2 * ph + dh) * W + 4 * wq + 2 * pw + dw]);
                    }
                    haar_fwd_block(a, o);
#pragma unroll
                    for (int sb = 0; sb < 8; ++sb) l1[sb][pt][ph][pw] = o[sb];
                }
        bf16_t* op = out + (((int64_t)tq * (Hq + 2 * halo) + hq + halo) * (Wq + 2 * halo) + wq + halo) * Cout;
#pragma unroll
        for (int sb1 = 0; sb1 < 8; ++sb1) {
            float o[8];
            haar_fwd_block(l1[sb1], o);
#pragma unroll
            for (int sb2 = 0; sb2 < 8; ++sb2) op[sb2 * (8 * Cin) + sb1 * Cin + c] = f2bf(o[sb2]);
        }
    }
}

// inverse: patches [Tp][Hq+2h][Wq+2h][64*Cout] -> video [Cout][4*Tp-3][4*Hq][4*Wq] planar (first 3 frames dropped)
__global__ __launch_bounds__(256) void haar_unpatch_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ v, int Cimg,
                                                           int Tp, int Hq, int Wq, int halo) {
    const int T = 4 * Tp - 3, H = 4 * Hq, W = 4 * Wq;
    const int Cin = 64 * Cimg;
    const int64_t total = (int64_t)Tp * Hq * Wq * Cimg;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cimg);
        int64_t r = i / Cimg;
        const int wq = (int)(r % Wq); r /= Wq;
        const int hq = (int)(r % Hq);
        const int tq = (int)(r / Hq);
        const bf16_t* ip = in + (((int64_t)tq * (Hq + 2 * halo) + hq + halo) * (Wq + 2 * halo) + wq + halo) * Cin;
        float l1[8][2][2][2];
#pragma unroll
        for (int sb1 = 0; sb1 < 8; ++sb1) {
            float s[8];
#pragma unroll
            for (int sb2 = 0; sb2 < 8; ++sb2) s[sb2] = bf2f(ip[sb2 * (8 * Cimg) + sb1 * Cimg + c]);
            haar_inv_block(s, l1[sb1]);
        }
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                for (int pw = 0; pw < 2; ++pw) {
                    float s[8], a[2][2][2];
#pragma unroll
                    for (int sb = 0; sb < 8; ++sb) s[sb] = l1[sb][pt][ph][pw];
                    haar_inv_block(s, a);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const int tau = 4 * tq + 2 * pt + dt - 3;
                        if (tau < 0) continue;
#pragma unroll
                        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                            for (int dw = 0; dw < 2; ++dw)
                                v[(((int64_t)c * T + tau) * H + 4 * hq + 2 * ph + dh) * W + 4 * wq + 2 * pw + dw] = f2bf(a[dt][dh][dw]);
                    }
                }
    }
}

static inline dim3 grid1d(int64_t n, int cap = 4096) {
    int64_t b = (n + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return dim3((unsigned)b);
}

extern "C" int drn_haar_patch(const void* video, void* out, int Cin, int T, int H, int W, int halo, void* stream) {
    DRN_CHECK_ARG(video && out && Cin > 0 && T > 0 && H > 0 && W > 0 && H % 4 == 0 && W % 4 == 0 && (T + 3) % 4 == 0);
    const int Tp = (T + 3) / 4;
    haar_patch_kernel<<<grid1d((int64_t)Tp * (H / 4) * (W / 4) * Cin), dim3(256), 0, (hipStream_t)stream>>>(
        (const bf16_t*)video, (bf16_t*)out, Cin, T, H, W, Tp, halo);
    return drn_launch_status();
}
extern "C" int drn_haar_unpatch(const void* patches, void* video, int Cimg, int Tp, int Hq, int Wq, int halo, void* stream) {
    DRN_CHECK_ARG(patches && video && Cimg > 0 && Tp > 0 && Hq > 0 && Wq > 0);
    haar_unpatch_kernel<<<grid1d((int64_t)Tp * Hq * Wq * Cimg), dim3(256), 0, (hipStream_t)stream>>>(
        (const bf16_t*)patches, (bf16_t*)video, Cimg, Tp, Hq, Wq, halo);
    return drn_launch_status();
}

// ------------------------------------------------------------------------------------------------ resampling helpers
// mode 0: spatial 2x2 average of the (0,1,0,1)-zero-padded image   (CosmosDownsample3d, spatial branch)
// mode 1: temporal 2-frame average of [x0, x0, x1, ...]            (temporal branch: out[t] = (x[max(2t-1,0)] + x[2t]) / 2)
// mode 2: temporal nearest x2 minus the first frame: out[t] = x[(t+1)/2]   (CosmosUpsample3d; identity if T == 1)
// mode 3: spatial nearest x2: out[h][w] = x[h/2][w/2]
__global__ __launch_bounds__(256) void resample_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int mode,
                                                       int T, int H, int W, int C, int To, int Ho, int Wo, int halo) {
    const int cch = C / 8;
    const int Hp = H + 2 * halo, Wp = W + 2 * halo, oHp = Ho + 2 * halo, oWp = Wo + 2 * halo;
    const int64_t total = (int64_t)To * Ho * Wo * cch;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % cch) * 8;
        int64_t r = i / cch;
        const int wo = (int)(r % Wo); r /= Wo;
        const int ho = (int)(r % Ho);
        const int to = (int)(r / Ho);
        float o[8];
        auto at = [&](int t, int h, int w, float* f) {     // halo reads return the stored zeros
            unpack8(*reinterpret_cast<const uint4*>(x + (((int64_t)t * Hp + h + halo) * Wp + w + halo) * C + c), f);
        };
        if (mode == 0) {
            float a[8], b[8], d[8], e[8];
            at(to, 2 * ho, 2 * wo, a); at(to, 2 * ho, 2 * wo + 1, b); at(to, 2 * ho + 1, 2 * wo, d); at(to, 2 * ho + 1, 2 * wo + 1, e);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (((a[j] + b[j]) + d[j]) + e[j]) * 0.25f;
        } else if (mode == 1) {
            float a[8], b[8];
            const int t0 = 2 * to - 1 < 0 ? 0 : 2 * to - 1;
            at(t0, ho, wo, a); at(2 * to < T ? 2 * to : T - 1, ho, wo, b);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (a[j] + b[j]) * 0.5f;
        } else if (mode == 2) {
            at(T > 1 ? (to + 1) / 2 : to, ho, wo, o);
        } else {
            at(to, ho / 2, wo / 2, o);
        }
        *reinterpret_cast<uint4*>(y + (((int64_t)to * oHp + ho + halo) * oWp + wo + halo) * C + c) = pack8(o);
    }
}

extern "C" int drn_resample(const void* x, void* y, int mode, int T, int H, int W, int C, int To, int Ho, int Wo, int halo,
                            void* stream) {
    DRN_CHECK_ARG(x && y && mode >= 0 && mode <= 3 && T > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && To > 0 && Ho > 0 && Wo > 0);
    DRN_CHECK_ARG(halo == 0 || halo == 1);
    if (mode == 0) DRN_CHECK_ARG(halo == 1 && To == T && Ho == H / 2 && Wo == W / 2 && H % 2 == 0 && W % 2 == 0);
    if (mode == 1) DRN_CHECK_ARG(To == (T + 1) / 2 && Ho == H && Wo == W);
    if (mode == 2) DRN_CHECK_ARG(To == (T > 1 ? 2 * T - 1 : 1) && Ho == H && Wo == W);
    if (mode == 3) DRN_CHECK_ARG(To == T && Ho == 2 * H && Wo == 2 * W);
    resample_kernel<<<grid1d((int64_t)To * Ho * Wo * (C / 8)), dim3(256), 0, (hipStream_t)stream>>>(
        (const bf16_t*)x, (bf16_t*)y, mode, T, H, W, C, To, Ho, Wo, halo);
    return drn_launch_status();
}

// ------------------------------------------------------------------------------------------------ attention helpers
// row softmax: fp32 scores [rows, ld] -> bf16 probabilities [rows, ldp] (columns >= n are written as zero up to ldp)
// (scale: the scores are multiplied by it first - one fp32 product per element, the same one a GEMM epilogue with alpha = scale forms)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, bf16_t* __restrict__ p, int n,
                                                           int64_t ld, int64_t ldp, float scale) {
    const int64_t row = blockIdx.x;
    const float* sr = s + row * ld;
    __shared__ float red[4];
    float m = -INFINITY;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, sr[i] * scale);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) sum += expf(sr[i] * scale - m);
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    sum = (red[0] + red[1]) + (red[2] + red[3]);
    const float inv = 1.0f / sum;
    bf16_t* pr = p + row * ldp;
    for (int i = threadIdx.x; i < ldp; i += 256) pr[i] = i < n ? f2bf(expf(sr[i] * scale - m) * inv) : (bf16_t)0;
}

// the same softmax with the row held in registers: ONE read of the fp32 scores (the kernel above reads them three times - 1.0 GB per
// 9216 x 9216 frame of the mid-block attention against 0.34 GB here).  One workgroup per row, thread t owns the 16-byte vectors
// t, t + 256, ...: n <= 1024 * VPT, n % 4 == 0, 16-byte aligned rows.  Same max / exp / normalise arithmetic as above (the fp32 row
// sum is taken over a different partition of the row).
template <int VPT>
__global__ __launch_bounds__(256) void softmax_rows_reg_kernel(const float* __restrict__ s, bf16_t* __restrict__ p, int n,
                                                               int64_t ld, int64_t ldp, float scale) {
    const int64_t row = blockIdx.x;
    const float4* sr = reinterpret_cast<const float4*>(s + row * ld);
    const int nv = n >> 2;
    __shared__ float red[4];
    float4 v[VPT];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int i = threadIdx.x + 256 * j;
        if (i < nv) {
            v[j] = sr[i];
            v[j].x *= scale; v[j].y *= scale; v[j].z *= scale; v[j].w *= scale;
            m = fmaxf(fmaxf(m, fmaxf(v[j].x, v[j].y)), fmaxf(v[j].z, v[j].w));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int i = threadIdx.x + 256 * j;
        if (i < nv) {
            v[j].x = expf(v[j].x - m); v[j].y = expf(v[j].y - m); v[j].z = expf(v[j].z - m); v[j].w = expf(v[j].w - m);
            sum += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
    }
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    sum = (red[0] + red[1]) + (red[2] + red[3]);
    const float inv = 1.0f / sum;
    uint2* pr = reinterpret_cast<uint2*>(p + row * ldp);
    const int nvp = (int)(ldp >> 2);
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        const int i = threadIdx.x + 256 * j;
        if (i < nv) pr[i] = make_uint2(pack_bf2(v[j].x * inv, v[j].y * inv), pack_bf2(v[j].z * inv, v[j].w * inv));
        else if (i < nvp) pr[i] = make_uint2(0u, 0u);
    }
    for (int i = threadIdx.x + 256 * VPT; i < nvp; i += 256) pr[i] = make_uint2(0u, 0u);
}

extern "C" int drn_softmax_rows_scaled(const void* scores, void* probs, int64_t rows, int n, int64_t ld, int64_t ldp, float scale,
                                       void* stream) {
    DRN_CHECK_ARG(scores && probs && rows >= 0 && n > 0 && ld >= n && ldp >= n && rows < (1ll << 31));
    if (rows == 0) return DRN_OK;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = n % 4 == 0 && ld % 4 == 0 && ldp % 4 == 0 && ((uintptr_t)scores & 15) == 0 && ((uintptr_t)probs & 7) == 0;
#define SM_LAUNCH(V) softmax_rows_reg_kernel<V><<<dim3((unsigned)rows), dim3(256), 0, st>>>((const float*)scores, (bf16_t*)probs, n, ld, ldp, scale)
    if (vec && n <= 1024 * 4) SM_LAUNCH(4);
    else if (vec && n <= 1024 * 9) SM_LAUNCH(9);          // 9216 keys: a 72 x 128 latent frame (the headline clip)
    else if (vec && n <= 1024 * 16) SM_LAUNCH(16);
    else softmax_rows_kernel<<<dim3((unsigned)rows), dim3(256), 0, st>>>((const float*)scores, (bf16_t*)probs, n, ld, ldp, scale);
#undef SM_LAUNCH
    return drn_launch_status();
}
extern "C" int drn_softmax_rows(const void* scores, void* probs, int64_t rows, int n, int64_t ld, int64_t ldp, void* stream) {
    return drn_softmax_rows_scaled(scores, probs, rows, n, ld, ldp, 1.0f, stream);
}

// transpose bf16 [rows, cols] -> [cols, ldo] (zero-filled beyond rows): V -> V^T for the P.V product
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int rows,
                                                        int cols, int64_t ldx, int64_t ldo) {
    __shared__ bf16_t tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int r = by + j, c = bx + tx;
        tile[j][tx] = (r < rows && c < cols) ? x[(int64_t)r * ldx + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = bx + j, r = by + tx;
        if (c < cols && r < ldo) y[(int64_t)c * ldo + r] = tile[tx][j];
    }
}

extern "C" int drn_transpose_bf16(const void* x, void* y, int rows, int cols, int64_t ldx, int64_t ldo, void* stream) {
    DRN_CHECK_ARG(x && y && rows > 0 && cols > 0 && ldx >= cols && ldo >= rows);
    dim3 grid((cols + 31) / 32, (int)((ldo + 31) / 32));
    transpose_kernel<<<grid, dim3(256), 0, (hipStream_t)stream>>>((const bf16_t*)x, (bf16_t*)y, rows, cols, ldx, ldo);
    return drn_launch_status();
}

// causal temporal attention of the mid block: q,k,v,o compact [T][P][C] (P pixels), one wave per pixel, 1 head of dim C.
// scores fp32, P rounded to bf16, output accumulated in fp32 (CosmosTemporalAttentionProcessor2_0 with a tril mask).
template <int TMAX>
__global__ __launch_bounds__(256) void temporal_attn_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                            const bf16_t* __restrict__ v, bf16_t* __restrict__ o, int T,
                                                            int64_t P, int C, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= P) return;
    float sc[TMAX][TMAX];
#pragma unroll
    for (int i = 0; i < TMAX; ++i)
#pragma unroll
        for (int j = 0; j < TMAX; ++j) sc[i][j] = 0.f;
    for (int c = lane * 8; c < C; c += 512) {
        float qv[TMAX][8], kv[TMAX][8];
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < T) {
                unpack8(*reinterpret_cast<const uint4*>(q + ((int64_t)t * P + pix) * C + c), qv[t]);
                unpack8(*reinterpret_cast<const uint4*>(k + ((int64_t)t * P + pix) * C + c), kv[t]);
            }
#pragma unroll
        for (int i = 0; i < TMAX; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j)
                if (i < T) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) sc[i][j] += qv[i][e] * kv[j][e];
                }
    }
    float pr[TMAX][TMAX];
#pragma unroll
    for (int i = 0; i < TMAX; ++i) {
        if (i >= T) continue;
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            sc[i][j] = wave_sum(sc[i][j]) * scale;
            m = fmaxf(m, sc[i][j]);
        }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            pr[i][j] = expf(sc[i][j] - m);
            sum += pr[i][j];
        }
#pragma unroll
        for (int j = 0; j <= i; ++j) pr[i][j] = rbf(pr[i][j] / sum);
    }
    for (int c = lane * 8; c < C; c += 512) {
        float vv[TMAX][8];
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < T) unpack8(*reinterpret_cast<const uint4*>(v + ((int64_t)t * P + pix) * C + c), vv[t]);
#pragma unroll
        for (int i = 0; i < TMAX; ++i) {
            if (i >= T) continue;
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j <= i; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += pr[i][j] * vv[j][e];
            *reinterpret_cast<uint4*>(o + ((int64_t)i * P + pix) * C + c) = pack8(acc);
        }
    }
}

extern "C" int drn_temporal_attention(const void* q, const void* k, const void* v, void* o, int T, int64_t P, int C,
                                      float scale, void* stream) {
    DRN_CHECK_ARG(q && k && v && o && T > 0 && T <= 16 && P > 0 && C > 0 && C % 8 == 0);
    dim3 grid((unsigned)((P + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
#define TA(N) temporal_attn_kernel<N><<<grid, block, 0, st>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, T, P, C, scale)
    if (T <= 1) TA(1);
    else if (T <= 2) TA(2);
    else if (T <= 4) TA(4);
    else if (T <= 8) TA(8);
    else TA(16);
#undef TA
    return drn_launch_status();
}

// ------------------------------------------------------------------------------------------------ latent layout moves
// planar [C][T][H][W] -> channels-last [T][H+2h][W+2h][Cs] (channels >= C are left untouched: zero from allocation)
__global__ __launch_bounds__(256) void planar_to_cl_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int C, int T,
                                                           int H, int W, int Cs, int halo) {
    const int64_t total = (int64_t)T * H * W * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        int64_t r = i / C;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H);
        const int t = (int)(r / H);
        y[(((int64_t)t * (H + 2 * halo) + h + halo) * (W + 2 * halo) + w + halo) * Cs + c] = x[(((int64_t)c * T + t) * H + h) * W + w];
    }
}
__global__ __launch_bounds__(256) void cl_to_planar_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int C, int T,
                                                           int H, int W, int Cs, int halo) {
    const int64_t total = (int64_t)C * T * H * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t r = i;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H); r /= H;
        const int t = (int)(r % T);
        const int c = (int)(r / T);
        y[i] = x[(((int64_t)t * (H + 2 * halo) + h + halo) * (W + 2 * halo) + w + halo) * Cs + c];
    }
}
extern "C" int drn_planar_to_cl(const void* x, void* y, int C, int T, int H, int W, int Cs, int halo, void* stream) {
    DRN_CHECK_ARG(x && y && C > 0 && T > 0 && H > 0 && W > 0 && Cs >= C && (halo == 0 || halo == 1));
    planar_to_cl_kernel<<<grid1d((int64_t)T * H * W * C), dim3(256), 0, (hipStream_t)stream>>>((const bf16_t*)x, (bf16_t*)y, C, T, H, W, Cs, halo);
    return drn_launch_status();
}
extern "C" int drn_cl_to_planar(const void* x, void* y, int C, int T, int H, int W, int Cs, int halo, void* stream) {
    DRN_CHECK_ARG(x && y && C > 0 && T > 0 && H > 0 && W > 0 && Cs >= C && (halo == 0 || halo == 1));
    cl_to_planar_kernel<<<grid1d((int64_t)T * H * W * C), dim3(256), 0, (hipStream_t)stream>>>((const bf16_t*)x, (bf16_t*)y, C, T, H, W, Cs, halo);
    return drn_launch_status();
}
