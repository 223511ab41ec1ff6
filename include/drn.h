/* drn.h - C ABI of libdrn.so: the MI355X (gfx950) kernels behind the DiffusionRenderer
 * denoising hot path (CleanGeneralDIT forward + EDM sampler steps + Cosmos CV8x8x8 tokenizer).
 *
 * The reference (eggsbenedicto/DiffusionRenderer-ComfyUI) is pure Python/torch and has no FFI;
 * each entry point below names the reference code (file:line under /root/reference) whose
 * arithmetic it replaces.  INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is DEVICE memory owned by the caller (e.g. torch tensor.data_ptr());
 *     the library allocates nothing and keeps no pointer after a call returns;
 *   - bf16 tensors are raw 16-bit words, row-major, innermost dimension contiguous;
 *   - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream); kernels are only
 *     enqueued, never synchronised;
 *   - return value: DRN_OK (0), DRN_EINVAL (-1) for a shape/alignment the kernels do not
 *     support (nothing was launched), or a positive hipError_t from the launch;
 *   - not thread-safe per stream; one host thread per process/GPU (ComfyUI runs nodes serially);
 *   - multi-GPU: one process per GPU.  SURVEY.md 8(b) sketched a `drn_comm_init(ncclUniqueId, rank, world)` entry; it was
 *     NOT built: the library has no communication state at all.  The sequence-parallel exchanges (head <-> token all-to-all,
 *     K|V all-gather; parallel.py) are issued by the host through torch.distributed (backend "nccl" = RCCL over xGMI) on the
 *     tensors the kernels below read and write - drn_gemm_bf16_blocked / drn_permute_021 produce and consume the rank-major
 *     slabs of those exchanges directly.
 *   - drn_attention_bf16: the output base and its row / batch strides must allow 16-byte stores (o % 16 == 0, ldo % 8 == 0);
 *     the K / V tile DMA addresses a row as a 32-bit byte offset from its tile's first row: 64 * ldk * 2 and 64 * ldv * 2 < 2^32.
 */
#ifndef DRN_H
#define DRN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRN_OK 0
#define DRN_EINVAL (-1)

#define DRN_ABI_VERSION 1

/* GEMM epilogues */
#define DRN_EPI_NONE 0      /* C = bf16(acc)                                            nn.Linear */
#define DRN_EPI_GELU 1      /* C = bf16(gelu_erf(bf16(acc)))                            CleanGeneralDIT.py:454-457 */
#define DRN_EPI_GATE_RES 2  /* C = bf16(R + bf16(gate * bf16(acc)))                     CleanGeneralDIT.py:517 */

/* GEMV input activation */
#define DRN_ACT_NONE 0
#define DRN_ACT_SILU 1      /* x <- bf16(silu(x)) before the product                     CleanGeneralDIT.py:342,485 */

int drn_abi_version(void);
const char* drn_error_string(int code);

/* ---- token-parallel GEMM: C[M,N] = epi(A[M,K] . W[N,K]^T), bf16 in, fp32 MFMA accumulate, bf16 out.
 * Replaces every nn.Linear of the DiT blocks (CleanGeneralDIT.py:273-276 q/k/v, :254-257 to_out,
 * :445-447 MLP, :386/:417 patch embed, :555/:590 final linear) plus the GELU (:446) and the gated
 * residual (:517) that follow them.
 * Requirements: K % 64 == 0, N % 128 == 0, lda/ldw/ldc/ldr % 8 == 0, 16-byte aligned bases.
 * gate: [batches, N] bf16 (row r uses batch r / rows_per_batch); residual R: [M, ldr] bf16 (may alias C). */
int drn_gemm_bf16(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K,
                  int64_t lda, int64_t ldw, int64_t ldc, int epilogue,
                  const void* gate, const void* residual, int64_t ldr, int64_t rows_per_batch,
                  void* stream);

/* ---- the same product with operands stored in column blocks: logical A[m][k] at
 * A + (k / a_block_cols) * a_block_stride + m * lda + k % a_block_cols, logical C[m][n] at
 * C + (n / c_block_cols) * c_block_stride + m * ldc + n % c_block_cols (block_cols 0 = plain; powers of two, >= 64 for A,
 * >= 256 for C).  New on this path (the reference has no multi-GPU code): the sequence-parallel K|V / Q projections write
 * the rank-major send buffer of the head <-> token all-to-all directly and the output projection reads the rank-major
 * receive buffer, instead of a regroup pass either side.  DRN_EINVAL when the shape would take the 128x128 kernel. */
int drn_gemm_bf16_blocked(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K,
                          int64_t lda, int64_t ldw, int64_t ldc, int epilogue,
                          const void* gate, const void* residual, int64_t ldr, int64_t rows_per_batch,
                          int64_t a_block_cols, int64_t a_block_stride, int64_t c_block_cols, int64_t c_block_stride,
                          void* stream);

/* ---- the same product for small M (a few hundred tokens: cfg 1, weight-streaming bound): `splits` workgroups share the K
 * range of every output tile so that enough CUs stream the weight matrix; fp32 partials go to `workspace`
 * (drn_gemm_splitk_workspace_bytes), a second kernel sums them and applies the epilogue.  drn_gemm_splitk_choice returns the
 * split count the dispatcher would use for a shape (1 = plain drn_gemm_bf16). */
int drn_gemm_bf16_splitk(const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K,
                         int64_t lda, int64_t ldw, int64_t ldc, int epilogue,
                         const void* gate, const void* residual, int64_t ldr, int64_t rows_per_batch,
                         int splits, void* workspace, void* stream);
int64_t drn_gemm_splitk_workspace_bytes(int64_t M, int64_t N, int splits);
/* C_f32[M, N] = A . W^T in fp32 as it leaves the accumulators (M, N multiples of 256, K of 64; 256 x 256 streamed kernel): products
 * whose result feeds a softmax - the tokenizer's one-head spatial attention (CleanVAE.py:50-60 -> diffusers' mid-block attention). */
int drn_gemm_bf16_f32out(const void* A, const void* W, float* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw, void* stream);

/* the K slices alone: fp32 partials [splits][M][N] in `workspace`, no sum, no epilogue (the same slices, bit for bit) */
int drn_gemm_bf16_splitk_partials(const void* A, const void* W, int64_t M, int64_t N, int64_t K,
                                  int64_t lda, int64_t ldw, int64_t rows_per_batch, int splits, void* workspace, void* stream);
int drn_gemm_splitk_choice(int64_t M, int64_t N, int64_t K);

/* ---- tuning hooks of drn_gemm_bf16 (no reference counterpart): which tile kernel the wave-quantisation model picks for an
 * [M, N] output (0: 128x128, 1: 256x256, 2: 144x256; N % 256 != 0 always takes 128x128), and a process-wide override for
 * A/B runs and tests (-1 = automatic). */
int drn_gemm_tile_choice(int64_t M, int64_t N);
void drn_gemm_force_tile(int tile);
/* tuning hook (tests / A-B runs): the gated-residual epilogue of the 256x256 kernel takes its residual tile through LDS, requested
 * under the last two K steps, when the launch allows it (whole tiles inside one clip, an even number >= 4 of K steps, 16-byte
 * residual rows); 0 makes every launch load it after the loop as the other kernels do (same bits), 1 restores the default,
 * -1 only queries.  Returns the previous setting.  (Environment: DRN_GEMM_RES_PREFETCH=0 disables it for the process.) */
int drn_gemm_force_res_prefetch(int on);
/* tuning hook (tests / A-B runs): tile of the few-token kernel (one clip of 256 rows), 0 = 256 rows x 64 columns, 1 = 128 x 128
 * (same bits), -1 = the default (environment DRN_GEMM_TALL_SHAPE, else built in).  Returns the previous setting. */
int drn_gemm_tall_force_shape(int shape);

/* ---- weight-streaming GEMV family (batch-1 vectors: timestep MLP, AdaLN-LoRA, the 1-key cross-attention).
 * For g in [0,groups), b in [0,batch): y[g,b,:] = epi(W[g] . act(x[g,b,:]))   W[g]: [N,K] bf16
 *   y = bf16(acc); if add: y = bf16(y + add[g,b,n]); if mul: y = bf16(mul[g,b,n] * y)
 * Replaces CleanGeneralDIT.py:358-360 (timestep MLP), :483-488/:500-505 (AdaLN-LoRA), :558-562/:572 (final AdaLN),
 * and to_v/to_out of the cross-attention whose single key makes softmax == 1.
 * Strides are in elements; a stride of 0 shares the operand between groups.  K % 8 == 0. */
int drn_gemv_bf16(const void* x, const void* W, void* y, int64_t N, int64_t K,
                  int groups, int batch,
                  int64_t x_gstride, int64_t x_bstride, int64_t w_gstride,
                  int64_t y_gstride, int64_t y_bstride,
                  const void* add, int64_t add_gstride, int64_t add_bstride,
                  const void* mul, int64_t mul_gstride, int64_t mul_bstride,
                  int act, void* stream);

/* ---- LayerNorm(eps, no affine) + AdaLN modulate, one pass: h = bf16(bf16(LN(x) * bf16(1+scale)) + shift)
 * with the reference's rounding points (CleanGeneralDIT.py:7-11, :481, :506; final layer :587).
 * If add_vec != NULL the row is first updated in place, x <- bf16(x + add_vec[batch]) (the broadcast
 * cross-attention residual, :517 with F8), and LN runs on the updated row.
 * x,h: [rows, D] bf16; shift/scale/add_vec: [batches, D] bf16; batch = row / rows_per_batch. D % 8 == 0, D <= 8192. */
int drn_ln_modulate(void* x, const void* add_vec, const void* shift, const void* scale, void* h,
                    int64_t rows, int64_t D, int64_t rows_per_batch, float eps, void* stream);
/* ---- the sum of split-K partials, the gated residual (CleanGeneralDIT.py:517) and the NEXT sub-block's LayerNorm + modulate
 * (:481, :506) in one pass over [rows, D] (few-token shapes: drn_dit_forward uses it where a linear was split along K):
 *   x <- bf16(x + bf16(gate * bf16(sum_s partials[s]))) [; x <- bf16(x + add_vec)];  h = modulate(LN(x)).
 * Rounds exactly where drn_gemm_bf16_splitk(DRN_EPI_GATE_RES) followed by drn_ln_modulate rounds: the same bits.
 * partials: fp32 [splits][rows][D]; gate / add_vec / shift / scale: [batches, D] bf16.  1024 < D <= 8192. */
int drn_splitk_gate_res_ln_modulate(const void* partials, int splits, void* x, const void* gate, const void* add_vec,
                                    const void* shift, const void* scale, void* h, int64_t rows, int64_t D,
                                    int64_t rows_per_batch, float eps, void* stream);
/* tuning hook (tests / A-B runs; no reference counterpart): the row statistics come from ONE summation tree that a
 * one-wave-per-row and a four-waves-per-row kernel share (same bits); -1 = chosen by row count, 0 / 1 force either. */
void drn_ln_force_kernel(int which);

/* ---- x[rows,D] <- bf16(x + vec[batch,:]) (stand-alone broadcast residual; same arithmetic as above) */
int drn_bcast_add(void* x, const void* vec, int64_t rows, int64_t D, int64_t rows_per_batch, void* stream);

/* ---- y[b][a][:] = x[a][b][:] for x [A, B, C] bf16 (C % 8 == 0): the regroup either side of the head <-> token all-to-all of
 * the sequence-parallel self-attention (token-major [rows, ranks, C] <-> rank-major [ranks, rows, C]); pure data movement.
 * New on this path: the reference has no multi-GPU code. */
int drn_permute_021(const void* x, void* y, int64_t A, int64_t B, int64_t C, void* stream);

/* ---- RMSNorm over the last dim, fp32 internal: y = bf16(x * rsqrt(mean(x^2)+eps) * w)   CleanGeneralDIT.py:14-33 */
int drn_rmsnorm(const void* x, const void* w, void* y, int64_t rows, int64_t D, float eps, void* stream);

/* ---- per-head RMSNorm(q), RMSNorm(k) + 3-D RoPE, in place (CleanGeneralDIT.py:288-295, :45-84).
 * q,k: [tokens, heads, 128] views with row strides ldq / ldk elements (e.g. slices of the fused QKV GEMM output);
 * wq,wk: [128] bf16; cos,sin: [tokens_per_batch, 128] bf16 host-built tables, NULL = no RoPE.
 * rotate_half pairs lane i with i+64.  token t uses table row pos_offset + (t % tokens_per_batch). head_dim must be 128.
 * Either q or k may be NULL (the sequence-parallel path normalises K first so its all-gather can start early). */
int drn_qk_norm_rope(void* q, void* k, const void* wq, const void* wk, const void* cos, const void* sin,
                     int64_t tokens, int heads, int64_t ldq, int64_t ldk, int64_t tokens_per_batch,
                     int64_t pos_offset, float eps, void* stream);

/* ---- non-causal scaled-dot-product attention, head_dim 128, online softmax in fp32, bf16 P for the PV MFMA.
 * Replaces F.scaled_dot_product_attention + the sbhd<->bhsd permutes + the F1 head flatten
 * (CleanGeneralDIT.py:181-203, :299-304).  q,o: [batch, Sq, heads, 128]; k,v: [batch, Sk, heads, 128] given by
 * element strides (token stride ld*, batch stride bs*; head h at offset h*128). */
int drn_attention_bf16(const void* q, const void* k, const void* v, void* o,
                       int batch, int heads, int64_t Sq, int64_t Sk,
                       int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                       int64_t bsq, int64_t bsk, int64_t bsv, int64_t bso,
                       float scale, void* stream);

/* tuning hook (tests / A-B runs; no reference counterpart): the kernel body on v_mfma_f32_16x16x32_bf16 (csrc/attention16.hip)
 * instead of 32x32x16; -1 = DRN_ATT16 from the environment / the build default, 0 / 1 forced.  Same arithmetic contract, a
 * different (equally valid) fp32 summation order. */
void drn_attention_force_shape16(int on);

/* ---- same attention with the keys cut into `nsplit` chunks (flash-decoding style): every (q-block, head, chunk) is a
 * workgroup writing an un-normalised fp32 partial into `workspace`, a second kernel merges them.  Arithmetic per chunk is
 * that of drn_attention_bf16; used when (q-blocks x heads) under-fills the 256 CUs, e.g. the 2304-query token bands of
 * 8-way sequence parallelism (288 workgroups -> 2304).  workspace: drn_attention_splitkv_workspace_bytes(...) bytes. */
int drn_attention_splitkv_bf16(const void* q, const void* k, const void* v, void* o,
                               int batch, int heads, int64_t Sq, int64_t Sk,
                               int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                               int64_t bsq, int64_t bsk, int64_t bsv, int64_t bso,
                               float scale, int nsplit, void* workspace, void* stream);
int64_t drn_attention_splitkv_workspace_bytes(int batch, int heads, int64_t Sq, int nsplit);

/* ---- how drn_dit_forward / the host wrapper cover the (q-block, head) grid of ONE clip with whole rounds of the 256 CUs:
 * plan[3 i + {0,1,2}] = (q_begin, q_end, kv_splits) for launch i; returns the number of launches (1 or 2).  The q-blocks that
 * fill whole rounds run unsplit, a fractional last round runs as a second launch with its keys split (split-KV + combine).
 * Host-only (no GPU needed).  No reference counterpart (the reference calls F.scaled_dot_product_attention). */
int drn_attention_plan(int heads, int64_t Sq, int64_t Sk, int64_t* plan);

/* ---- one whole CleanGeneralDIT forward on one GPU, enqueued by ONE call: patch-embed GEMM, n_sub sub-blocks
 * ({FA, CA, MLP} x 28), final LayerNorm + Linear.  Replaces the module loop of CleanGeneralDIT.py:686-706 (x_embedder,
 * `for block in self.blocks.values()`, final_layer) as a SEQUENCER: every launch is one of the kernels above with the arguments
 * the per-launch host path passes (dit_engine.HipDiT._run), so the results are bit-identical to it; ~570 launches cost the host
 * ~2 ms from here instead of 4-7 ms through ctypes.  Patchify before and unpatchify after stay separate calls.
 * B clips of S tokens are stacked along the rows (row = b S + s; attention runs per clip).  All pointers device memory. */
#define DRN_SUB_FA 0     /* self-attention sub-block      CleanGeneralDIT.py:465-517 with block_type "FA" */
#define DRN_SUB_CA 1     /* cross-attention sub-block: one key -> softmax == 1 -> x += bf16(gate * to_out(to_v(ctx))) */
#define DRN_SUB_MLP 2    /* GPT-2 feed-forward sub-block  CleanGeneralDIT.py:442-462 */
typedef struct drn_dit_sub {
    int32_t kind;        /* DRN_SUB_* */
    int32_t site;        /* AdaLN site: row of the shift / scale / gate tables */
    int32_t ca_index;    /* CA: row of `addvec`; else -1 */
    int32_t reserved;
    const void* w_a;     /* FA: fused q|k|v weights [3D, D];  MLP: layer1 [hidden, D] */
    const void* w_b;     /* FA: to_out [D, D];               MLP: layer2 [D, hidden] */
    const void* qn;      /* FA: RMSNorm weight of q [128] */
    const void* kn;      /* FA: RMSNorm weight of k [128] */
} drn_dit_sub;
typedef struct drn_dit_forward_args {
    int64_t struct_bytes;                 /* sizeof(drn_dit_forward_args): ABI check */
    int64_t S, B;                         /* tokens per clip, clips */
    int64_t D, hidden;                    /* model width (heads x 128), MLP width */
    int32_t heads, n_sub;
    const drn_dit_sub* subs;              /* HOST array of n_sub entries */
    const void* shift; const void* scale; const void* gate;     /* AdaLN vectors: site s at base + s * site_stride, [B or 1][D] rows */
    int64_t shift_site_stride, scale_site_stride, gate_site_stride;       /* in elements */
    const void* addvec; int64_t addvec_stride;                  /* CA: bf16(gate * c_i) rows, entry i at addvec + i * stride, [B or 1][D] */
    const void* cos; const void* sin;     /* RoPE tables [S, 128] */
    const void* P; int64_t kpad; const void* w_patch;           /* patchified input [B S, kpad], patch-embed weight [D, kpad] */
    const void* final_shift; const void* final_scale; const void* w_final; int64_t n_final;   /* final layer; w_final [n_final, D] */
    void* X; void* H; void* QKV; void* O; void* U; void* Y;     /* activations: [B S, D] x2, [B S, 3D], [B S, D], [B S, hidden], [B S, n_final] */
    void* gemm_ws; int64_t gemm_ws_bytes;                       /* split-K partials (drn_dit_forward_gemm_workspace_bytes) */
    void* attn_ws; int64_t attn_ws_bytes;                       /* split-KV partials (drn_dit_forward_attn_workspace_bytes) */
    void* timer;                          /* drn_timer_create handle or NULL */
    float eps; int32_t reserved;
} drn_dit_forward_args;
int drn_dit_forward(const drn_dit_forward_args* args, void* stream);
int64_t drn_dit_forward_args_bytes(void);     /* sizeof the two structs as this library was compiled (binding self-check) */
int64_t drn_dit_sub_bytes(void);
int64_t drn_dit_forward_gemm_workspace_bytes(int64_t B, int64_t S, int64_t D, int64_t hidden, int64_t n_final, int64_t kpad);
int64_t drn_dit_forward_attn_workspace_bytes(int64_t B, int heads, int64_t S);

/* ---- per-launch timing inside drn_dit_forward (the roofline leg of bench.py; no reference counterpart): a pool of HIP event
 * pairs; every `sample_every`-th GEMM (kind 0) and attention (kind 1) call of a forward is bracketed on the launch stream.
 * Read after the stream is synchronised.  The one object this library allocates, owned by the caller via create / destroy. */
void* drn_timer_create(int capacity, int sample_every);
void drn_timer_destroy(void* timer);
int drn_timer_count(void* timer);
int drn_timer_seen(void* timer, int kind);
int drn_timer_read(void* timer, int index, int* kind, float* ms, double* flops, double* bytes);

/* ---- patchify + channel concat: out[b*T*H*W + (t,h,w), (c r m n)] gathered from x | cond | ones-mask
 * (CleanGeneralDIT.py:669-675, :409-414).  x: [B,Cx,Tl,Hl,Wl], cond: [B,Cc,Tl,Hl,Wl] bf16; with_mask appends the
 * all-ones channel; columns [C*pt*ps*ps, ldo) are zero-filled (K padding for the GEMM).  Bit-exact index op. */
int drn_patchify_concat(const void* x, const void* cond, void* out, int B, int Cx, int Cc, int with_mask,
                        int Tl, int Hl, int Wl, int pt, int ps, int64_t ldo, void* stream);

/* ---- unpatchify: y[(B T)(H W), (ph pw pt C)] -> out[B, C, T*pt, H*ps, W*ps]  (CleanGeneralDIT.py:709-716). Bit-exact. */
int drn_unpatchify(const void* y, int64_t ldy, void* out, int B, int C, int Tp, int Hp, int Wp, int pt, int ps,
                   void* stream);

/* ---- EDM Euler sampler, fp32 math on bf16 latents (model_diffusion_renderer.py:30-82, :232) */
int drn_edm_scale_input(const void* x, void* out, int64_t n, float c_in, void* stream);
int drn_edm_step(const void* model_out, const void* sample, void* out, int64_t n,
                 float c_skip, float c_out, float sigma, float dt, void* stream);
int drn_cfg_combine(const void* cond, const void* uncond, void* out, int64_t n, float guidance, void* stream);

/* ---- pipeline post-process (diffusion_renderer_pipeline.py:299-318): optional normal re-normalisation blend,
 * (1+v).clamp(0,2)/2, permute to (B,T,H,W,C), *255, truncating uint8 cast.  video: [B,3,T,H,W] bf16. Bit-exact. */
int drn_postprocess_u8(const void* video, void* out_u8, int B, int T, int H, int W, int normalize_normal, void* stream);

/* =====================================================================================================
 * Cosmos-1.0 CV8x8x8 tokenizer (CleanVAE.py:45-60 -> diffusers.AutoencoderKLCosmos, un-vendored: parity unpinned).
 * Activations are channels-last bf16, X[t][h][w][c], stored with an optional 1-pixel zero halo in H and W
 * (`halo` = 0 compact / 1 padded); every kernel writes interior pixels only, so a zero-initialised halo stays zero.
 * ===================================================================================================== */

/* ---- causal conv3d as an implicit GEMM on MFMA (CosmosCausalConv3d / ConvProjection3d / 1x1x1 convs, and - with a
 * 1x1x1 kernel over a compact [M,K] matrix - the dense GEMMs of the mid-block attention).
 *   y[p, n] = bf16(bias[n] + sum x[in(p,tap), c] * w[n, tap*C + c]) (+ residual[p, n], rounded again)
 *   or, out_f32 != 0:  y_f32[p, n] = alpha * (bias[n] + sum ...)
 * x: [T][H+2*in_halo][W+2*in_halo][C]; w: [N][kT*kH*kW*C] (tap-major repack of the conv weight); bias: [N] or NULL;
 * y / residual: [To][Ho+2*out_halo][Wo+2*out_halo][ldc / ldr].  Input pixel of tap (kt,kh,kw) for output (to,ho,wo):
 *   t = max(to*sT + kt - t_off, 0) (causal replicate padding), h = ho*sH + kh - pad, w = wo*sW + kw - pad
 * (pad = 1 reads the zero halo).  C % 64 == 0, N % 4 == 0. */
int drn_conv3d_igemm(const void* x, const void* w, const void* bias, void* y, const void* residual,
                     int T, int H, int W, int C, int in_halo, int N,
                     int kT, int kH, int kW, int sT, int sH, int sW, int pad, int t_off,
                     int To, int Ho, int Wo, int out_halo, int64_t ldc, int64_t ldr,
                     int out_f32, float alpha, void* stream);
/* tuning hooks (no reference counterpart): the big convolutions (N % 256 == 0, bf16 output, >= 192 tiles of 256 positions x
 * 256 channels, input < 4 GiB) run on the streamed 256x256 kernel (csrc/conv256s.hip), everything else on the 128x128 one;
 * results are bit-identical.  drn_conv_force_tile: -1 automatic, 0 always 128x128, 1 256x256 wherever it can run;
 * drn_conv_last_tile: the kernel the last drn_conv3d_igemm call launched (0 / 1). */
void drn_conv_force_tile(int tile);
int drn_conv_last_tile(void);

/* ---- CosmosCausalGroupNorm(1 group, per frame) [+ SiLU]: y = [silu](bf16((x-mean)*rstd*gamma + beta)).
 * workspace: drn_groupnorm_workspace_bytes(frames) bytes of device scratch. */
int drn_groupnorm_silu(const void* x, const void* gamma, const void* beta, void* y, void* workspace,
                       int frames, int H, int W, int C, int halo, float eps, int silu, void* stream);
int64_t drn_groupnorm_workspace_bytes(int frames);
/* the same normalisation in two steps, for frames whose rows are spread over several GPUs: per-rank
 * partial sums part[frames][64][2] as fp64 (sum, sum of squares over the stored band; the halo rows must still be zero; fp64 makes
 * the totals independent of the split), then - after the ranks' partials were gathered to part[frames][nparts][2] - the normalisation of this rank's band with
 * count = H_total * W * C elements per frame. */
int drn_groupnorm_stats(const void* x, void* part, int frames, int H, int W, int C, int halo, void* stream);
int drn_groupnorm_apply(const void* x, const void* part, int nparts, float count, const void* gamma, const void* beta,
                        void* y, int frames, int H, int W, int C, int halo, float eps, int silu, void* stream);

/* ---- 2-level 3-D Haar patching (CosmosPatchEmbed3d, patch 4): video [Cin][T][H][W] planar ->
 * [ (T+3)/4 ][H/4+2*halo][W/4+2*halo][64*Cin]; and its inverse (CosmosUnpatcher3d) -> [C][4*Tp-3][4*Hq][4*Wq]. */
int drn_haar_patch(const void* video, void* out, int Cin, int T, int H, int W, int halo, void* stream);
int drn_haar_unpatch(const void* patches, void* video, int Cimg, int Tp, int Hq, int Wq, int halo, void* stream);

/* ---- resampling helpers of CosmosDownsample3d / CosmosUpsample3d.  mode 0: spatial 2x2 mean, 1: causal temporal
 * 2-frame mean, 2: temporal nearest x2 minus the first frame, 3: spatial nearest x2. */
int drn_resample(const void* x, void* y, int mode, int T, int H, int W, int C, int To, int Ho, int Wo, int halo,
                 void* stream);

/* ---- mid-block attention pieces (1 head of dim C): fp32 row softmax -> bf16, bf16 transpose, causal temporal attention */
int drn_softmax_rows(const void* scores, void* probs, int64_t rows, int n, int64_t ld, int64_t ldp, void* stream);
/* the same with the scores multiplied by `scale` first (one fp32 product per element: what a GEMM epilogue with alpha = scale forms) */
int drn_softmax_rows_scaled(const void* scores, void* probs, int64_t rows, int n, int64_t ld, int64_t ldp, float scale, void* stream);
int drn_transpose_bf16(const void* x, void* y, int rows, int cols, int64_t ldx, int64_t ldo, void* stream);
int drn_temporal_attention(const void* q, const void* k, const void* v, void* o, int T, int64_t P, int C, float scale,
                           void* stream);

/* ---- latent layout moves: planar [C][T][H][W] <-> channels-last [T][H+2*halo][W+2*halo][Cs] (Cs >= C) */
int drn_planar_to_cl(const void* x, void* y, int C, int T, int H, int W, int Cs, int halo, void* stream);
int drn_cl_to_planar(const void* x, void* y, int C, int T, int H, int W, int Cs, int halo, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DRN_H */
