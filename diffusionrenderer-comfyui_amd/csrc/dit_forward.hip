// drn_dit_forward: the launch sequence of one CleanGeneralDIT forward on one GPU, enqueued by ONE call.
//
// Replaces the Python-level block loop of the reference (CleanGeneralDIT.py:686-706: patch embed, 28 x {FA, CA, MLP}, final
// layer) as a SEQUENCER only: every launch below is one of the kernels behind the other entry points of drn.h, called with the
// arguments the per-launch host path (dit_engine.HipDiT._run) passes, in the same order - results are bit-identical to it.
// Why it exists: a forward is ~570 launches; through ctypes + torch wrappers each costs the host 6-12 us, which at S = 256
// (cfg 1: a 6.8 ms GPU step) made the host the bound of the denoising loop.  From C the same launches cost the host ~2 ms.
// Host code only (no kernel lives in this file).
#include <stdlib.h>
#include "drn_common.h"

// ---- how to cover the (q-block, head) grid with whole rounds of the 256 CUs (was native.attention_plan; measured cost model)
static const int kCUs = 256;

static int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

static int pick_kv_splits(int64_t batch, int64_t heads, int64_t Sq, int64_t Sk, double* cost_out) {
    const int64_t blocks = batch * heads * ((Sq + 255) / 256);
    int best = 1;
    double best_cost = (double)ceil_div(blocks, kCUs);
    const int cand[3] = {2, 4, 8};
    for (int i = 0; i < 3; ++i) {
        const int n = cand[i];
        if (Sk / n < 512) break;
        // ~4 % of an unsplit workgroup per extra workgroup (prologue, fp32 partial store) and ~6 % for the combine pass
        const double cost = (double)ceil_div(blocks * n, kCUs) * (1.0 / n + 0.04) + 0.06;
        const double lim = (double)ceil_div(blocks, kCUs) * 0.93;
        if (cost < (best_cost < lim ? best_cost : lim)) {
            best = n;
            best_cost = cost;
        }
    }
    if (cost_out) *cost_out = best_cost;
    return best;
}

// plan[3 * i + {0, 1, 2}] = (q_begin, q_end, kv_splits); returns the number of launches (1 or 2)
extern "C" int drn_attention_plan(int heads, int64_t Sq, int64_t Sk, int64_t* plan) {
    if (heads <= 0 || Sq <= 0 || Sk <= 0 || !plan) return 0;
    const int64_t per_qb = heads;                       // the plan of ONE clip (batch-invariant summation order)
    const int64_t nqb = (Sq + 255) / 256;
    const int64_t blocks = nqb * per_qb;
    const int n_all = pick_kv_splits(1, heads, Sq, Sk, nullptr);
    const double cost_all = n_all == 1 ? (double)ceil_div(blocks, kCUs)
                                       : (double)ceil_div(blocks * n_all, kCUs) * (1.0 / n_all + 0.04) + 0.06;
    const int64_t rounds = blocks / kCUs;
    const int64_t nqb_main = (rounds * kCUs) / per_qb;
    plan[0] = 0; plan[1] = Sq; plan[2] = n_all;
    if (rounds == 0 || nqb_main == 0 || nqb_main == nqb) return 1;
    const int64_t q_cut = nqb_main * 256;
    const int n_tail = pick_kv_splits(1, heads, Sq - q_cut, Sk, nullptr);
    const int64_t tail_blocks = (nqb - nqb_main) * per_qb;
    const double cost_tail = n_tail == 1 ? (double)ceil_div(tail_blocks, kCUs)
                                         : (double)ceil_div(tail_blocks * n_tail, kCUs) * (1.0 / n_tail + 0.04) + 0.06;
    const double cost_two = (double)ceil_div(nqb_main * per_qb, kCUs) + cost_tail + 0.02;      // + the launch boundary
    if (cost_two < cost_all) {
        plan[0] = 0; plan[1] = q_cut; plan[2] = 1;
        plan[3] = q_cut; plan[4] = Sq; plan[5] = n_tail;
        return 2;
    }
    return 1;
}

// ---- optional per-launch timing (bench.py's roofline leg): HIP event pairs around every `sample_every`-th GEMM / attention
struct drn_timer {
    int capacity, used, sample_every;
    int seen[2];                                        // launches seen per kind (0 = gemm, 1 = attention)
    hipEvent_t* ev;                                     // 2 per record
    int* kind;
    double* flops;
    double* bytes;
};

extern "C" void* drn_timer_create(int capacity, int sample_every) {
    if (capacity <= 0) return nullptr;
    drn_timer* t = new drn_timer();
    t->capacity = capacity;
    t->used = 0;
    t->sample_every = sample_every > 0 ? sample_every : 1;
    t->seen[0] = t->seen[1] = 0;
    t->ev = new hipEvent_t[2 * (size_t)capacity];
    t->kind = new int[capacity];
    t->flops = new double[capacity];
    t->bytes = new double[capacity];
    for (int i = 0; i < 2 * capacity; ++i)
        if (hipEventCreate(&t->ev[i]) != hipSuccess) {
            for (int j = 0; j < i; ++j) (void)hipEventDestroy(t->ev[j]);
            delete[] t->ev; delete[] t->kind; delete[] t->flops; delete[] t->bytes;
            delete t;
            return nullptr;
        }
    return t;
}

extern "C" void drn_timer_destroy(void* h) {
    drn_timer* t = (drn_timer*)h;
    if (!t) return;
    for (int i = 0; i < 2 * t->capacity; ++i) (void)hipEventDestroy(t->ev[i]);
    delete[] t->ev; delete[] t->kind; delete[] t->flops; delete[] t->bytes;
    delete t;
}

extern "C" int drn_timer_count(void* h) { return h ? ((drn_timer*)h)->used : 0; }
extern "C" int drn_timer_seen(void* h, int kind) { return (h && (kind == 0 || kind == 1)) ? ((drn_timer*)h)->seen[kind] : 0; }

// record i after the stream has been synchronised: kind, milliseconds, algorithmic FLOPs and bytes of that launch
extern "C" int drn_timer_read(void* h, int i, int* kind, float* ms, double* flops, double* bytes) {
    drn_timer* t = (drn_timer*)h;
    if (!t || i < 0 || i >= t->used) return DRN_EINVAL;
    const hipError_t e = hipEventElapsedTime(ms, t->ev[2 * i], t->ev[2 * i + 1]);
    if (e != hipSuccess) return (int)e;
    *kind = t->kind[i];
    *flops = t->flops[i];
    *bytes = t->bytes[i];
    return DRN_OK;
}

namespace {
struct Scope {                                           // event pair around one sampled launch
    drn_timer* t;
    int slot;
    hipStream_t st;
    Scope(drn_timer* timer, int kind, double flops, double bytes, hipStream_t stream) : t(timer), slot(-1), st(stream) {
        if (!t) return;
        const int c = t->seen[kind]++;
        if (c % t->sample_every != 0 || t->used >= t->capacity) return;
        slot = t->used++;
        t->kind[slot] = kind;
        t->flops[slot] = flops;
        t->bytes[slot] = bytes;
        (void)hipEventRecord(t->ev[2 * slot], st);
    }
    ~Scope() {
        if (slot >= 0) (void)hipEventRecord(t->ev[2 * slot + 1], st);
    }
};
}  // namespace

#define DRN_TRY(expr)                  \
    do {                               \
        const int rc_ = (expr);        \
        if (rc_ != DRN_OK) return rc_; \
    } while (0)

// out = epi(A . W^T) exactly as native.gemm dispatches it: split-K for few-token products (decided from ONE clip's rows).
// `defer`: a split-K product with the gated-residual epilogue may stop after its slices (returns *defer = split count): the
// caller folds sum + epilogue into the next LayerNorm pass (drn_splitk_gate_res_ln_modulate: the same bits).
static int fwd_gemm(const drn_dit_forward_args* a, const void* A, const void* W, void* C, int64_t M, int64_t N, int64_t K,
                    int64_t lda, int64_t ldc, int epi, const void* gate, const void* residual, int64_t ldr, void* stream,
                    int* defer = nullptr) {
    const int64_t rpb = a->S;
    const int64_t Mb = (rpb > 0 && rpb < M && M % rpb == 0) ? rpb : M;
    const int splits = Mb <= 1024 ? drn_gemm_splitk_choice(Mb, N, K) : 1;
    Scope sc((drn_timer*)a->timer, 0, 2.0 * M * N * K, 2.0 * (M * K + N * K + M * N * (residual ? 2 : 1)), (hipStream_t)stream);
    if (defer) *defer = 0;
    if (splits > 1) {
        if (!a->gemm_ws || drn_gemm_splitk_workspace_bytes(M, N, splits) > a->gemm_ws_bytes) return DRN_EINVAL;
        if (defer && epi == DRN_EPI_GATE_RES && N == a->D && N > 1024 && ldc == N && ldr == N && residual == C) {
            *defer = splits;
            return drn_gemm_bf16_splitk_partials(A, W, M, N, K, lda, K, rpb, splits, a->gemm_ws, stream);
        }
        return drn_gemm_bf16_splitk(A, W, C, M, N, K, lda, K, ldc, epi, gate, residual, ldr, rpb, splits, a->gemm_ws, stream);
    }
    return drn_gemm_bf16(A, W, C, M, N, K, lda, K, ldc, epi, gate, residual, ldr, rpb, stream);
}

extern "C" int64_t drn_dit_forward_attn_workspace_bytes(int64_t B, int heads, int64_t S) {
    int64_t plan[6];
    const int n = drn_attention_plan(heads, S, S, plan);
    int64_t need = 0;
    for (int i = 0; i < n; ++i)
        if (plan[3 * i + 2] > 1) {
            const int64_t b = drn_attention_splitkv_workspace_bytes((int)B, heads, plan[3 * i + 1] - plan[3 * i], (int)plan[3 * i + 2]);
            need = b > need ? b : need;
        }
    return need;
}

extern "C" int64_t drn_dit_forward_gemm_workspace_bytes(int64_t B, int64_t S, int64_t D, int64_t hidden, int64_t n_final, int64_t kpad) {
    const int64_t M = B * S;
    const int64_t shapes[6][2] = {{D, kpad}, {3 * D, D}, {D, D}, {hidden, D}, {D, hidden}, {n_final, D}};
    int64_t need = 0;
    for (int i = 0; i < 6; ++i) {
        const int splits = S <= 1024 ? drn_gemm_splitk_choice(S, shapes[i][0], shapes[i][1]) : 1;
        const int64_t b = drn_gemm_splitk_workspace_bytes(M, shapes[i][0], splits);
        need = b > need ? b : need;
    }
    return need;
}

extern "C" int64_t drn_dit_forward_args_bytes(void) { return (int64_t)sizeof(drn_dit_forward_args); }
extern "C" int64_t drn_dit_sub_bytes(void) { return (int64_t)sizeof(drn_dit_sub); }

extern "C" int drn_dit_forward(const drn_dit_forward_args* a, void* stream) {
    DRN_CHECK_ARG(a && a->struct_bytes == (int64_t)sizeof(drn_dit_forward_args));
    DRN_CHECK_ARG(a->S > 0 && a->B > 0 && a->D > 0 && a->heads > 0 && a->D == (int64_t)a->heads * 128 && a->hidden > 0);
    DRN_CHECK_ARG(a->n_sub >= 0 && (a->n_sub == 0 || a->subs) && a->X && a->H && a->QKV && a->O && a->U && a->Y);
    DRN_CHECK_ARG(a->P && a->w_patch && a->w_final && a->final_shift && a->final_scale && a->shift && a->scale && a->gate);
    const int64_t S = a->S, B = a->B, D = a->D, M = B * S;
    const bf16_t* shift = (const bf16_t*)a->shift;
    const bf16_t* scale = (const bf16_t*)a->scale;
    const bf16_t* gate = (const bf16_t*)a->gate;
    const float sm_scale = (float)(1.0 / sqrt(128.0));      // the double -> float conversion of the host wrapper (native.attention)

    // patch embedding (CleanGeneralDIT.py:386/:417): X = P . w_patch^T
    DRN_TRY(fwd_gemm(a, a->P, a->w_patch, a->X, M, D, a->kpad, a->kpad, D, DRN_EPI_NONE, nullptr, nullptr, 0, stream));

    const bf16_t* pending = nullptr;                     // the broadcast cross-attention residual not yet added to X
    int deferred = 0;                                    // > 0: X still lacks sum(partials) + gate/residual of the last linear
    const bf16_t* deferred_gate = nullptr;
    static int fuse = -1;
    if (fuse < 0) {
        const char* e = getenv("DRN_FUSE_SPLITK_LN");    // 0: the split-K epilogue and the LayerNorm as two launches (A/B runs)
        fuse = (e && e[0] == '0') ? 0 : 1;
    }
    // LayerNorm + modulate of X into H, folding a deferred split-K epilogue in
#define LN_NEXT(SH, SC)                                                                                               \
    do {                                                                                                              \
        if (deferred) {                                                                                               \
            DRN_TRY(drn_splitk_gate_res_ln_modulate(a->gemm_ws, deferred, a->X, deferred_gate, pending, SH, SC, a->H, M, D, S,  \
                                                    a->eps, stream));                                                 \
            deferred = 0;                                                                                             \
        } else {                                                                                                      \
            DRN_TRY(drn_ln_modulate(a->X, pending, SH, SC, a->H, M, D, S, a->eps, stream));                           \
        }                                                                                                             \
        pending = nullptr;                                                                                            \
    } while (0)
    for (int i = 0; i < a->n_sub; ++i) {
        const drn_dit_sub* sb = &a->subs[i];
        const bf16_t* sh = shift + (int64_t)sb->site * a->shift_site_stride;
        const bf16_t* sc = scale + (int64_t)sb->site * a->scale_site_stride;
        const bf16_t* gt = gate + (int64_t)sb->site * a->gate_site_stride;
        if (sb->kind == DRN_SUB_CA) {
            DRN_CHECK_ARG(a->addvec && sb->ca_index >= 0);
            if (pending) {
                // two cross-attention blocks in a row (not in FA-CA-MLP): X must be complete before the stand-alone add
                if (deferred) {
                    DRN_TRY(drn_splitk_gate_res_ln_modulate(a->gemm_ws, deferred, a->X, deferred_gate, nullptr, sh, sc, a->H, M, D, S,
                                                            a->eps, stream));       // (H is scratch here)
                    deferred = 0;
                }
                DRN_TRY(drn_bcast_add(a->X, pending, M, D, S, stream));
            }
            pending = (const bf16_t*)a->addvec + (int64_t)sb->ca_index * a->addvec_stride;
            continue;
        }
        LN_NEXT(sh, sc);
        if (sb->kind == DRN_SUB_FA) {
            DRN_CHECK_ARG(sb->w_a && sb->w_b && sb->qn && sb->kn && a->cos && a->sin);
            bf16_t* q = (bf16_t*)a->QKV;
            bf16_t* k = q + D;
            bf16_t* v = q + 2 * D;
            DRN_TRY(fwd_gemm(a, a->H, sb->w_a, a->QKV, M, 3 * D, D, D, 3 * D, DRN_EPI_NONE, nullptr, nullptr, 0, stream));
            DRN_TRY(drn_qk_norm_rope(q, k, sb->qn, sb->kn, a->cos, a->sin, M, a->heads, 3 * D, 3 * D, S, 0, a->eps, stream));
            {
                Scope tsc((drn_timer*)a->timer, 1, 4.0 * B * a->heads * S * S * 128, 2.0 * B * a->heads * 128 * (4.0 * S), (hipStream_t)stream);
                int64_t plan[6];
                const int n = drn_attention_plan(a->heads, S, S, plan);
                for (int p = 0; p < n; ++p) {
                    const int64_t q0 = plan[3 * p], nq = plan[3 * p + 1] - q0;
                    const int ns = (int)plan[3 * p + 2];
                    const bf16_t* qs = q + q0 * 3 * D;
                    bf16_t* os = (bf16_t*)a->O + q0 * D;
                    if (ns > 1) {
                        if (!a->attn_ws || drn_attention_splitkv_workspace_bytes((int)B, a->heads, nq, ns) > a->attn_ws_bytes) return DRN_EINVAL;
                        DRN_TRY(drn_attention_splitkv_bf16(qs, k, v, os, (int)B, a->heads, nq, S, 3 * D, 3 * D, 3 * D, D, S * 3 * D,
                                                           S * 3 * D, S * 3 * D, S * D, sm_scale, ns, a->attn_ws, stream));
                    } else {
                        DRN_TRY(drn_attention_bf16(qs, k, v, os, (int)B, a->heads, nq, S, 3 * D, 3 * D, 3 * D, D, S * 3 * D, S * 3 * D,
                                                   S * 3 * D, S * D, sm_scale, stream));
                    }
                }
            }
            DRN_TRY(fwd_gemm(a, a->O, sb->w_b, a->X, M, D, D, D, D, DRN_EPI_GATE_RES, gt, a->X, D, stream, fuse ? &deferred : nullptr));
            deferred_gate = gt;
        } else if (sb->kind == DRN_SUB_MLP) {
            DRN_CHECK_ARG(sb->w_a && sb->w_b);
            DRN_TRY(fwd_gemm(a, a->H, sb->w_a, a->U, M, a->hidden, D, D, a->hidden, DRN_EPI_GELU, nullptr, nullptr, 0, stream));
            DRN_TRY(fwd_gemm(a, a->U, sb->w_b, a->X, M, D, a->hidden, a->hidden, D, DRN_EPI_GATE_RES, gt, a->X, D, stream,
                             fuse ? &deferred : nullptr));
            deferred_gate = gt;
        } else {
            return DRN_EINVAL;
        }
    }
    // final layer (CleanGeneralDIT.py:583-590): LN + modulate with the first 2D of the LoRA vector, Linear(D -> n_final)
    LN_NEXT((const bf16_t*)a->final_shift, (const bf16_t*)a->final_scale);
#undef LN_NEXT
    DRN_TRY(fwd_gemm(a, a->H, a->w_final, a->Y, M, a->n_final, D, D, a->n_final, DRN_EPI_NONE, nullptr, nullptr, 0, stream));
    return DRN_OK;
}
