// bf16 GEMM for the DiT linears on gfx950 MFMA:  C[M,N] = epi(A[M,K] . W[N,K]^T)
//
// Both operands are K-contiguous (activations [tokens, K], nn.Linear weights [out, K]), so both tiles are
// staged by direct-to-LDS 16-byte loads (global_load_lds_dwordx4) and read back as 8-element K fragments.
//
// Tile 128x128x64, 256 threads = 4 waves (2 x 2), 64x64 per wave as 4x4 v_mfma_f32_16x16x32_bf16 tiles.
// The MFMA is issued as D = Wfrag x Afrag so a lane ends up with 4 CONSECUTIVE output features of one token
// (8-byte bf16 stores, and the gate / residual of the fused epilogue are read the same way).
//
// LDS: 2 stages x (A 16 KiB + W 16 KiB) = 64 KiB -> 2 workgroups per CU.  Rows are 128 B; the 16-byte chunk
// index is XOR-swizzled with (row>>1)&7, which makes every ds_read_b128 lane group hit 16 distinct 16-B slots
// of the 256-B bank row (conflict-free).  global_load_lds writes LDS lane-linearly, so the swizzle is applied
// to the per-lane SOURCE address and again on the read (cdna guide rule 21).
//
// Workgroup ids are remapped so that each XCD (own L2) walks a contiguous band of tiles, 8 tile-rows deep.
#include <stdlib.h>
#include "drn_common.h"

#define BM 128
#define BN 128
#define BK 64
#define TILE_BYTES (BM * BK * 2)          // 16 KiB per operand per stage
#define STAGE_BYTES (2 * TILE_BYTES)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                                           bf16_t* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                                                           int64_t ldw, int64_t ldc, const bf16_t* __restrict__ gate,
                                                           const bf16_t* R, int64_t ldr, int64_t rpb,
                                                           float* __restrict__ part) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * STAGE_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- XCD-aware, grouped tile order
    const int tiles_m = (int)((M + BM - 1) / BM);
    const int tiles_n = (int)(N / BN);
    const int nwg = tiles_m * tiles_n;
    int pid;
    {
        const int bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int GROUP = 8;
    const int width = GROUP * tiles_n;
    const int group_id = pid / width;
    const int first_m = group_id * GROUP;
    const int gsz = min(tiles_m - first_m, GROUP);
    const int tm = first_m + (pid % width) % gsz;
    const int tn = (pid % width) / gsz;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

    // ---- staging: each wave copies 4 x 1 KiB pieces (8 rows x 128 B) of A and of W per K step
    const bf16_t* ga[4];
    const bf16_t* gw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave * 4 + i) * 8 + (lane >> 3);          // tile row 0..127
        const int c = (lane & 7) ^ ((r >> 1) & 7);               // source chunk for LDS slot (lane&7)
        int64_t ra = m0 + r;
        if (ra > M - 1) ra = M - 1;                              // clamp: duplicates are never stored
        ga[i] = A + ra * lda + c * 8;
        gw[i] = W + (n0 + r) * ldw + c * 8;
    }
    auto stage = [&](int kt, int buf) {
        char* sa = smem + buf * STAGE_BYTES + wave * 4096;
        char* sw = sa + TILE_BYTES;
        const int64_t ko = (int64_t)kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_global_load_lds((gptr_t)(ga[i] + ko), (lptr_t)(sa + i * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(gw[i] + ko), (lptr_t)(sw + i * 1024), 16, 0, 0);
        }
    };

    // ---- fragment read offsets (bytes within an operand tile), per k-substep ks: chunk = ks*4 + (lane>>4)
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

    // split-K (gridDim.y > 1, small M: too few tiles to keep the CUs streaming weights): this workgroup multiplies K steps
    // [kbeg, kbeg + nk) and stores its fp32 partial tile; gemm_splitk_epilogue_kernel sums the partials and applies the epilogue
    const int nk = (int)(K / BK) / (int)gridDim.y;
    const int kbeg = (int)blockIdx.y * nk;
    stage(kbeg, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage(kbeg + kt + 1, (kt + 1) & 1);
        const char* sa = smem + (kt & 1) * STAGE_BYTES;
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

    if (part) {                                                   // split-K partial: fp32, [split][M][N]
        float* pp = part + (int64_t)blockIdx.y * M * N;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t m = m0 + wm * 64 + i * 16 + fr;
            if (m >= M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<f32x4_t*>(pp + m * N + n0 + wn * 64 + j * 16 + fq * 4) = acc[i][j];
        }
        return;
    }
    // ---- epilogue: lane holds C[m][n..n+3], m = tile row (lane&15), n = 16 j + 4*(lane>>4) + r; column tiles j = 2p / 2p+1 are
    //      exchanged between lane rows fq = 2k / 2k+1 (v_permlane16_swap) so that a lane stores 16 B (see gemm256s.hip)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m_raw = m0 + wm * 64 + i * 16 + fr;
        const bool m_ok = m_raw < M;
        const int64_t m = m_ok ? m_raw : M - 1;
        const int64_t b = (EPI == DRN_EPI_GATE_RES) ? m / rpb : 0;
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
            uint2 o[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int j = 2 * jp + h;
                const int64_t n = n0 + wn * 64 + j * 16 + fq * 4;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = rbf(acc[i][j][r]);
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
                o[h].x = pack_bf2(v[0], v[1]);
                o[h].y = pack_bf2(v[2], v[3]);
            }
            const auto sx = __builtin_amdgcn_permlane16_swap(o[0].x, o[1].x, false, false);
            const auto sy = __builtin_amdgcn_permlane16_swap(o[0].y, o[1].y, false, false);
            const int64_t n8 = n0 + wn * 64 + jp * 32 + (fq & 1) * 16 + (fq >> 1) * 8;
            if (m_ok) *reinterpret_cast<uint4*>(C + m * ldc + n8) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        }
    }
}

// sum of the split-K partials + the fused epilogue (same arithmetic as the in-kernel epilogue on the fp32 total)
template <int EPI>
__global__ __launch_bounds__(256) void gemm_splitk_epilogue_kernel(const float* __restrict__ part, int splits, bf16_t* C,
                                                                   int64_t M, int64_t N, int64_t ldc,
                                                                   const bf16_t* __restrict__ gate, const bf16_t* R,
                                                                   int64_t ldr, int64_t rpb) {
    const int64_t nq = N / 4, total = M * nq;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / nq, n = (i - m * nq) * 4;
        f32x4_t a = *reinterpret_cast<const f32x4_t*>(part + m * N + n);
        for (int s = 1; s < splits; ++s) a += *reinterpret_cast<const f32x4_t*>(part + ((int64_t)s * M + m) * N + n);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = rbf(a[r]);
        if (EPI == DRN_EPI_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gelu_erf_fast(v[r]);
        } else if (EPI == DRN_EPI_GATE_RES) {
            const uint2 g2 = *reinterpret_cast<const uint2*>(gate + (m / rpb) * N + n);
            const uint2 r2 = *reinterpret_cast<const uint2*>(R + m * ldr + n);
            const float g[4] = {bflo(g2.x), bfhi(g2.x), bflo(g2.y), bfhi(g2.y)};
            const float x[4] = {bflo(r2.x), bfhi(r2.x), bflo(r2.y), bfhi(r2.y)};
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = x[r] + rbf(g[r] * v[r]);
        }
        uint2 o;
        o.x = pack_bf2(v[0], v[1]);
        o.y = pack_bf2(v[2], v[3]);
        *reinterpret_cast<uint2*>(C + m * ldc + n) = o;
    }
}

// gemm256s.hip: 256x256x64 tile, streamed schedule (tile kernel 1; 3 and 4 are accepted as aliases by drn_gemm_force_tile)
int drn_gemm256s_dispatch(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                          int64_t ldc, int epilogue, const void* gate, const void* residual, int64_t ldr, int64_t rpb,
                          void* stream, const int64_t* blk);

// gemm144.hip: 144x256x64 kernel (token bands of sequence parallelism: M = 2304 k)
int drn_gemm144_dispatch(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                         int64_t ldc, int epilogue, const void* gate, const void* residual, int64_t ldr, int64_t rpb,
                         void* stream, const int64_t* blk);

// Tile choice by a wave-quantisation model (measured on MI355X).  Time unit = one 256^2 workgroup owning a CU for the whole
// K loop.  128^2 workgroups run two per CU at ~1000 vs ~1300 TF/s: a full round of 512 takes ~0.65 units.  A 144x256
// workgroup does 56 % of the work of a 256^2 one at 0.85-1.0x its rate: DRN_GEMM144_COST = 0.64 units (measured: M = 2304
// -> 0.79 vs 0.97 ms per DiT block; M = 18432 stays on 256^2).
// Returns 0: 128^2, 1: 256^2, 2: 144x256.   DRN_GEMM256=0 / DRN_GEMM144=0 switch a kernel off, DRN_GEMM144=2 forces it (A/B runs).
// drn_gemm_force_tile: -1 automatic, 0 / 1 / 2 as above; 3 = 1 (kept for callers of round 2); 4 = 1 with the split-K slices of
// the small-M path on the 2 + 2 stage kernel instead of the weight-ring kernel (tests, A/B).
static int g_force_tile = -1;
extern "C" void drn_gemm_force_tile(int tile) { g_force_tile = tile; }
static int tall_choice(int64_t M, int64_t N, int64_t K);          // few-token kernel (gemm_tall.hip), defined with the split-K rules
int drn_gemm_tall_dispatch(const void* A, const void* W, void* C, float* partial, int64_t M, int64_t N, int64_t K, int64_t lda,
                           int64_t ldw, int64_t ldc, int epilogue, const void* gate, const void* residual, int64_t ldr,
                           int64_t rpb, int splits, void* stream);

static int pick_gemm_tile(int64_t M, int64_t N, double* cost_out = nullptr) {
    static int mode256 = -1, mode144 = -1;
    static double cost144 = 0.0;
    if (mode256 < 0) {
        const char* e = getenv("DRN_GEMM256");
        mode256 = (e && e[0] == '0') ? 0 : 1;
        e = getenv("DRN_GEMM144");
        mode144 = e ? atoi(e) : 1;
        e = getenv("DRN_GEMM144_COST");
        cost144 = e ? atof(e) : 0.64;
    }
    if (cost_out) *cost_out = 1e30;
    if (N < 256 || N % 256 != 0) return 0;
    if (g_force_tile >= 0) return g_force_tile;
    const int64_t t256 = ((M + 255) / 256) * (N / 256), t128 = ((M + 127) / 128) * (N / 128), t144 = ((M + 143) / 144) * (N / 256);
    const double c128 = 0.65 * (double)((t128 + 511) / 512);
    const double c256 = (mode256 == 1 && M >= 256) ? (double)((t256 + 255) / 256) : 1e30;
    // (well below one round of workgroups the model says nothing - measured at M = 1024, N = 4096: 128 workgroups of 144x256
    //  550 TF/s vs 256 of 128^2 655 TF/s - so the 144-row tile needs at least 3/4 of a round)
    const double c144 = (mode144 >= 1 && M >= 144 && t144 >= 192) ? cost144 * (double)((t144 + 255) / 256) : 1e30;
    if (mode144 == 2 && M >= 144) return 2;
    const int best = (c144 < c256 && c144 < c128) ? 2 : (c256 <= c128 ? 1 : 0);
    if (cost_out) *cost_out = best == 2 ? c144 : (best == 1 ? c256 : c128);
    return best;
}

extern "C" int drn_gemm_tile_choice(int64_t M, int64_t N) { return pick_gemm_tile(M, N); }

// blk = {a_shift, a_block_stride, c_shift, c_block_stride}; shift 62 = plain layout
static const int64_t kPlain[4] = {62, 0, 62, 0};

static int gemm_launch_tile(int tile, const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                            int64_t ldw, int64_t ldc, int epilogue, const void* gate, const void* residual, int64_t ldr,
                            int64_t rows_per_batch, void* stream, const int64_t* blk) {
    if (tile == 1 || tile == 3 || tile == 4)
        return drn_gemm256s_dispatch(A, W, C, M, N, K, lda, ldw, ldc, epilogue, gate, residual, ldr, rows_per_batch, stream, blk);
    if (tile == 2)
        return drn_gemm144_dispatch(A, W, C, M, N, K, lda, ldw, ldc, epilogue, gate, residual, ldr, rows_per_batch, stream, blk);
    const int64_t tiles = ((M + BM - 1) / BM) * (N / BN);
    DRN_CHECK_ARG(tiles < (1ll << 31));
    dim3 grid((unsigned)tiles), block(256);
    hipStream_t st = (hipStream_t)stream;
#define ARGS (const bf16_t*)A, (const bf16_t*)W, (bf16_t*)C, M, N, K, lda, ldw, ldc, (const bf16_t*)gate, \
             (const bf16_t*)residual, ldr, rows_per_batch, (float*)nullptr
    switch (epilogue) {
        case DRN_EPI_NONE: gemm_bf16_kernel<DRN_EPI_NONE><<<grid, block, 0, st>>>(ARGS); break;
        case DRN_EPI_GELU: gemm_bf16_kernel<DRN_EPI_GELU><<<grid, block, 0, st>>>(ARGS); break;
        case DRN_EPI_GATE_RES: gemm_bf16_kernel<DRN_EPI_GATE_RES><<<grid, block, 0, st>>>(ARGS); break;
        default: return DRN_EINVAL;
    }
#undef ARGS
    return drn_launch_status();
}

// How the rows of ONE clip (M rows, one gate row) are covered: the tile kernel for the rows that fill whole rounds of 256^2
// workgroups, and - when the last round would be fractional (72 x 16 tiles = 4.5 rounds cost 5) - a second launch for the
// remaining rows with the tile that suits them (DRN_GEMM_TAIL=0 switches the split off).  Returns the row count of the first
// launch (M = no split) and its tile.
static int64_t gemm_plan_rows(int64_t M, int64_t N, bool blocked, int* tile_out) {
    double cost_all = 0.0;
    const int tile = pick_gemm_tile(M, N, &cost_all);
    *tile_out = tile;
    static int tail_mode = -1;
    if (tail_mode < 0) {
        const char* e = getenv("DRN_GEMM_TAIL");
        tail_mode = (e && e[0] == '0') ? 0 : 1;
    }
    if (tile != 1 || tail_mode != 1 || g_force_tile >= 0) return M;
    const int64_t tm = (M + 255) / 256, tn = N / 256;
    const int64_t rounds = tm * tn / 256, rem = tm * tn % 256;
    const int64_t tm_main = rounds * 256 / tn;
    if (rounds < 1 || rem == 0 || tm_main < 1 || tm_main >= tm) return M;
    const int64_t M_main = tm_main * 256;
    double cost_tail = 0.0;
    const int tail_tile = pick_gemm_tile(M - M_main, N, &cost_tail);
    if (blocked && tail_tile == 0) cost_tail = 1e30;
    const double cost_split = (double)((tm_main * tn + 255) / 256) + cost_tail + 0.02;
    return cost_split < cost_all - 0.05 ? M_main : M;
}

// one clip (or a plain problem): rows_per_batch >= M here
static int gemm_clip(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                     int64_t ldc, int epilogue, const void* gate, const void* residual, int64_t ldr, void* stream,
                     const int64_t* blk) {
    const bool blocked = blk[0] != 62 || blk[2] != 62;
    int tile = 0;
    const int64_t M_main = gemm_plan_rows(M, N, blocked, &tile);
    if (blocked && tile == 0) return DRN_EINVAL;           // the 128x128 kernel has no blocked layouts (callers regroup instead)
    if (M_main == M)
        return gemm_launch_tile(tile, A, W, C, M, N, K, lda, ldw, ldc, epilogue, gate, residual, ldr, M, stream, blk);
    const int rc = gemm_launch_tile(1, A, W, C, M_main, N, K, lda, ldw, ldc, epilogue, gate, residual, ldr, M_main, stream, blk);
    if (rc != DRN_OK) return rc;
    return gemm_clip((const bf16_t*)A + M_main * lda, W, (bf16_t*)C + M_main * ldc, M - M_main, N, K, lda, ldw, ldc, epilogue, gate,
                     residual ? (const bf16_t*)residual + M_main * ldr : nullptr, ldr, stream, blk);
}

static int gemm_impl(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                     int64_t ldw, int64_t ldc, int epilogue, const void* gate, const void* residual,
                     int64_t ldr, int64_t rows_per_batch, void* stream, const int64_t* blk) {
    DRN_CHECK_ARG(A && W && C && M >= 0 && N > 0 && K > 0);
    DRN_CHECK_ARG(K % BK == 0 && N % BN == 0);
    DRN_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0 && ldw >= K);
    DRN_CHECK_ARG((blk[0] != 62 || lda >= K) && (blk[2] != 62 || ldc >= N));
    DRN_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)C & 15) == 0);
    if (epilogue == DRN_EPI_GATE_RES)
        DRN_CHECK_ARG(gate && residual && ldr % 8 == 0 && ldr >= N && rows_per_batch > 0 && ((uintptr_t)residual & 7) == 0);
    if (M == 0) return DRN_OK;
    if (epilogue < DRN_EPI_NONE || epilogue > DRN_EPI_GATE_RES) return DRN_EINVAL;
    // B clips stacked along the rows (rows_per_batch = rows of one clip): the tile kernel and the tail split decide the
    // accumulation order of an output element, so both are chosen from ONE clip's rows - a clip then gets the same bits
    // whatever it is batched with (the reference steps its G-buffer passes / CFG halves one clip at a time, nodes.py:187-213).
    const bool batched = rows_per_batch > 0 && rows_per_batch < M && M % rows_per_batch == 0 && blk[0] == 62 && blk[2] == 62;
    if (blk[0] == 62 && blk[2] == 62 && M % 256 == 0 && lda >= K && tall_choice(batched ? rows_per_batch : M, N, K) == 1)
        return drn_gemm_tall_dispatch(A, W, C, nullptr, M, N, K, lda, ldw, ldc, epilogue, gate, residual, ldr,
                                      rows_per_batch > 0 ? rows_per_batch : M, 1, stream);
    if (!batched) {
        if (epilogue == DRN_EPI_GATE_RES && rows_per_batch < M) {
            // ragged batches (the last one shorter): one launch with the tile of the whole problem, gates by row / rows_per_batch
            int tile = 0;
            (void)gemm_plan_rows(M, N, blk[0] != 62 || blk[2] != 62, &tile);
            return gemm_launch_tile(tile, A, W, C, M, N, K, lda, ldw, ldc, epilogue, gate, residual, ldr, rows_per_batch, stream, blk);
        }
        return gemm_clip(A, W, C, M, N, K, lda, ldw, ldc, epilogue, gate, residual, ldr, stream, blk);
    }
    const int64_t Mb = rows_per_batch, nb = M / Mb;
    int tile = 0;
    if (gemm_plan_rows(Mb, N, false, &tile) == Mb)          // one kernel covers a clip: one launch covers them all
        return gemm_launch_tile(tile, A, W, C, M, N, K, lda, ldw, ldc, epilogue, gate, residual, ldr, Mb, stream, blk);
    for (int64_t b = 0; b < nb; ++b) {
        const int rc = gemm_clip((const bf16_t*)A + b * Mb * lda, W, (bf16_t*)C + b * Mb * ldc, Mb, N, K, lda, ldw, ldc, epilogue,
                                 epilogue == DRN_EPI_GATE_RES ? (const bf16_t*)gate + b * N : nullptr,
                                 residual ? (const bf16_t*)residual + b * Mb * ldr : nullptr, ldr, stream, blk);
        if (rc != DRN_OK) return rc;
    }
    return DRN_OK;
}

extern "C" int drn_gemm_bf16(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                             int64_t ldw, int64_t ldc, int epilogue, const void* gate, const void* residual,
                             int64_t ldr, int64_t rows_per_batch, void* stream) {
    return gemm_impl(A, W, C, M, N, K, lda, ldw, ldc, epilogue, gate, residual, ldr, rows_per_batch, stream, kPlain);
}

static int log2_exact(int64_t v) {
    int s = 0;
    while ((1ll << s) < v) ++s;
    return (1ll << s) == v ? s : -1;
}

// Same product with operands stored in column blocks ("planes"): logical A[m][k] lives at
// A + (k / a_block_cols) * a_block_stride + m * lda + k % a_block_cols, logical C[m][n] at
// C + (n / c_block_cols) * c_block_stride + m * ldc + n % c_block_cols (block_cols = 0: plain).  Block widths are powers of
// two, >= 64 for A and >= 256 for C.  Lets the sequence-parallel projections write the rank-major send buffer of the
// head <-> token all-to-all directly, and the output projection read the rank-major receive buffer.
extern "C" int drn_gemm_bf16_blocked(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                                     int64_t ldw, int64_t ldc, int epilogue, const void* gate, const void* residual,
                                     int64_t ldr, int64_t rows_per_batch, int64_t a_block_cols, int64_t a_block_stride,
                                     int64_t c_block_cols, int64_t c_block_stride, void* stream) {
    int64_t blk[4] = {62, 0, 62, 0};
    if (a_block_cols) {
        const int sh = log2_exact(a_block_cols);
        DRN_CHECK_ARG(sh >= 6 && K % a_block_cols == 0 && a_block_stride % 8 == 0 && lda >= a_block_cols);
        blk[0] = sh;
        blk[1] = a_block_stride;
    }
    if (c_block_cols) {
        const int sh = log2_exact(c_block_cols);
        DRN_CHECK_ARG(sh >= 8 && N % c_block_cols == 0 && c_block_stride % 8 == 0 && ldc >= c_block_cols);
        blk[2] = sh;
        blk[3] = c_block_stride;
    }
    return gemm_impl(A, W, C, M, N, K, lda, ldw, ldc, epilogue, gate, residual, ldr, rows_per_batch, stream, blk);
}

// ---- split-K for small M (weight-streaming regime): `splits` workgroups per output tile, fp32 partials in `workspace`
extern "C" int64_t drn_gemm_splitk_workspace_bytes(int64_t M, int64_t N, int splits) {
    return splits > 1 ? (int64_t)splits * M * N * (int64_t)sizeof(float) : 0;
}

// how many K splits keep the CUs busy for an [M, N, K] product (1 = none): at most half of the 512 workgroup slots filled by
// 128^2 tiles, at most 512 workgroups after the split and at least 16 K steps left per split.  Callers with B clips stacked
// along the rows pass ONE clip's rows (the split changes the summation order: it must not depend on the batch)
// Since round 2 the slices of a product whose 256 x 256 tiles x splits fill at least 3/4 of the CUs with >= 16 K steps each run
// on the streamed kernel (gemm256s.hip; one workgroup per CU, the A slice read once per 256 output columns instead of once per
// 128): QKV (48 tiles x 4), MLP-up (64 x 4), MLP-down (16 x 16).  DRN_SPLITK256=0 switches that off (A/B runs).
// gemm_tall.hip: one clip of 256 tokens - a workgroup owns all 256 rows x 64 columns over the whole K (no slices where N / 64
// workgroups fill the chip: QKV 192, MLP-up 256), or over a K slice (out-proj, MLP-down: 64 tiles x 4).  Returns the slice
// count (1 = unsplit, fused epilogue), 0 = not this kernel.  DRN_GEMM_TALL=0 switches it off (A/B runs).
int drn_gemm_tall_dispatch(const void* A, const void* W, void* C, float* partial, int64_t M, int64_t N, int64_t K, int64_t lda,
                           int64_t ldw, int64_t ldc, int epilogue, const void* gate, const void* residual, int64_t ldr,
                           int64_t rpb, int splits, void* stream);
static int tall_choice(int64_t M, int64_t N, int64_t K) {
    static int mode = -1;
    if (mode < 0) {
        const char* e = getenv("DRN_GEMM_TALL");
        mode = (e && e[0] == '0') ? 0 : 1;
    }
    if (!mode || g_force_tile >= 0 || M != 256 || N % 64 != 0 || K % BK != 0) return 0;
    const int64_t tiles = N / 64;
    if (tiles >= 192) return 1;
    // K slices only while a slice stays short (out-proj: 64 tiles x 4 slices of 16 K steps, 18 + 5 us against 21 + 9 on the
    // 128^2 path); with 64 K steps per slice every workgroup takes in a quarter of the A panel per slice and the 256^2 slices
    // win (MLP-down: 58 + 5 us here against 43 + 9; round 3, 128 x 128 tiles, 4 slices of 64 K steps: 59 us against 58 in all)
    for (int s = 2; s <= 8; s *= 2)
        if (tiles * s >= 192 && tiles * s <= 256 && (K / BK) % s == 0 && K / s >= 1024 && K / s <= 2048) return s;
    return 0;
}

static int splitk256_choice(int64_t M, int64_t N, int64_t K) {
    static int mode = -1;
    if (mode < 0) {
        const char* e = getenv("DRN_SPLITK256");
        mode = (e && e[0] == '0') ? 0 : 1;
    }
    if (!mode || M % 256 != 0 || N % 256 != 0 || K % BK != 0 || M > 1024) return 0;
    const int64_t tiles = (M / 256) * (N / 256);
    int best = 0;
    for (int s = 2; s <= 16; s *= 2)
        if (tiles * s <= 256 && (K / BK) % s == 0 && K / s >= 1024) best = s;
    return (best && tiles * best >= 192) ? best : 0;
}

int drn_gemm256s_partial(const void* A, const void* W, float* partial, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                         int splits, void* stream, bool wring_ok);

extern "C" int drn_gemm_splitk_choice(int64_t M, int64_t N, int64_t K) {
    if (N % BN != 0 || K % BK != 0 || M <= 0) return 1;
    {
        const int st = tall_choice(M, N, K);
        if (st) return st;
    }
    if (g_force_tile < 0 || g_force_tile == 4) {           // (4: the same rule, slices on the 2 + 2 stage kernel - tests, A/B)
        const int s256 = splitk256_choice(M, N, K);
        if (s256) return s256;
    }
    if (pick_gemm_tile(M, N) != 0) return 1;
    const int64_t tiles = ((M + BM - 1) / BM) * (N / BN);
    int best = 1;
    for (int s = 2; s <= 8; s *= 2)
        if (tiles * s <= 512 && (K / BK) % s == 0 && K / s >= 1024) best = s;
    return tiles <= 256 ? best : 1;
}

// the K slices of a split-K product: fp32 partials [splits][M][N] in `workspace`, nothing else (arguments validated by the callers)
static int splitk_slices(const void* A, const void* W, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                         int64_t rows_per_batch, int splits, void* workspace, void* stream) {
    const int64_t tiles = ((M + BM - 1) / BM) * (N / BN);
    DRN_CHECK_ARG(tiles < 65536);
    hipStream_t st = (hipStream_t)stream;
    // which kernel computes the slices is a function of ONE clip's rows and the split count the caller got from
    // drn_gemm_splitk_choice for those rows (batch-invariant results)
    const int64_t Mb = (rows_per_batch > 0 && rows_per_batch < M && M % rows_per_batch == 0) ? rows_per_batch : M;
    if (M % 256 == 0 && tall_choice(Mb, N, K) == splits)
        return drn_gemm_tall_dispatch(A, W, workspace, (float*)workspace, M, N, K, lda, ldw, N, DRN_EPI_NONE, nullptr, nullptr, 0,
                                      rows_per_batch, splits, stream);
    if ((g_force_tile < 0 || g_force_tile == 4) && M % 256 == 0 && splitk256_choice(Mb, N, K) == splits)
        return drn_gemm256s_partial(A, W, (float*)workspace, M, N, K, lda, ldw, splits, stream, g_force_tile != 4);
    gemm_bf16_kernel<DRN_EPI_NONE><<<dim3((unsigned)tiles, (unsigned)splits), dim3(256), 0, st>>>(
        (const bf16_t*)A, (const bf16_t*)W, (bf16_t*)workspace, M, N, K, lda, ldw, N, nullptr, nullptr, 0, 1, (float*)workspace);
    return drn_launch_status();
}

// C_f32[M, N] = A . W^T as it leaves the accumulators (no rounding, no epilogue): ONE "slice" of the streamed 256 x 256 kernel's
// split-K form.  For products whose fp32 result feeds a softmax (the tokenizer's one-head spatial attention: 9216 x 9216 x 512 per
// frame).  M, N multiples of 256, K a multiple of 64; same K order per output element as every other tile kernel here.
extern "C" int drn_gemm_bf16_f32out(const void* A, const void* W, float* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                                    void* stream) {
    DRN_CHECK_ARG(A && W && C && M > 0 && N > 0 && K > 0 && M % 256 == 0 && N % 256 == 0 && K % BK == 0);
    DRN_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K);
    DRN_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)C & 15) == 0);
    return drn_gemm256s_partial(A, W, C, M, N, K, lda, ldw, 1, stream, false);
}

// slices only: the caller sums them itself (drn_splitk_gate_res_ln_modulate folds the sum, the gated residual and the next
// LayerNorm into one pass).  Same slices, bit for bit, as drn_gemm_bf16_splitk computes.
extern "C" int drn_gemm_bf16_splitk_partials(const void* A, const void* W, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw,
                                             int64_t rows_per_batch, int splits, void* workspace, void* stream) {
    DRN_CHECK_ARG(A && W && workspace && M > 0 && N > 0 && K > 0 && splits > 1 && splits <= 64);
    DRN_CHECK_ARG(K % BK == 0 && N % BN == 0 && (K / BK) % splits == 0);
    DRN_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && lda >= K && ldw >= K);
    DRN_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)workspace & 15) == 0);
    return splitk_slices(A, W, M, N, K, lda, ldw, rows_per_batch, splits, workspace, stream);
}

extern "C" int drn_gemm_bf16_splitk(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K, int64_t lda,
                                    int64_t ldw, int64_t ldc, int epilogue, const void* gate, const void* residual,
                                    int64_t ldr, int64_t rows_per_batch, int splits, void* workspace, void* stream) {
    if (splits <= 1)
        return drn_gemm_bf16(A, W, C, M, N, K, lda, ldw, ldc, epilogue, gate, residual, ldr, rows_per_batch, stream);
    DRN_CHECK_ARG(A && W && C && workspace && M > 0 && N > 0 && K > 0 && splits <= 64);
    DRN_CHECK_ARG(K % BK == 0 && N % BN == 0 && (K / BK) % splits == 0);
    DRN_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0 && lda >= K && ldw >= K && ldc >= N);
    DRN_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)C & 15) == 0 && ((uintptr_t)workspace & 15) == 0);
    if (epilogue == DRN_EPI_GATE_RES)
        DRN_CHECK_ARG(gate && residual && ldr % 8 == 0 && ldr >= N && rows_per_batch > 0 && ((uintptr_t)residual & 7) == 0);
    if (epilogue < DRN_EPI_NONE || epilogue > DRN_EPI_GATE_RES) return DRN_EINVAL;
    {
        const int rc = splitk_slices(A, W, M, N, K, lda, ldw, rows_per_batch, splits, workspace, stream);
        if (rc != DRN_OK) return rc;
    }
    hipStream_t st = (hipStream_t)stream;
    int64_t blocks = (M * (N / 4) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
#define EARGS (const float*)workspace, splits, (bf16_t*)C, M, N, ldc, (const bf16_t*)gate, (const bf16_t*)residual, ldr, rows_per_batch
    switch (epilogue) {
        case DRN_EPI_NONE: gemm_splitk_epilogue_kernel<DRN_EPI_NONE><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(EARGS); break;
        case DRN_EPI_GELU: gemm_splitk_epilogue_kernel<DRN_EPI_GELU><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(EARGS); break;
        default: gemm_splitk_epilogue_kernel<DRN_EPI_GATE_RES><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(EARGS); break;
    }
#undef EARGS
    return drn_launch_status();
}
