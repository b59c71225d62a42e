// Decoder engine behind the C ABI: owns the paged KV pool, activation workspace and the captured
// hipGraph of one decode step.  Replaces GPTWithKV.forward / GPTBlock.forward / sample_kvcache
// (api_cache.py:51-74, 87-106, 159-184); weight names follow remap_state_dict (api_cache.py:118-134).
//
// Reference quirks reproduced on purpose: no attention mask anywhere (prefill is
// bidirectional), every call adds pos_emb[:T] so decode steps always use position row 0, the
// prefill logits are dropped and the first decode step re-feeds the last prompt token (which
// therefore sits in the cache twice).
#include <stdlib.h>

#include <mutex>
#include <vector>

#include "common.h"

using namespace mgea;

namespace {
enum { T_TOK = 0, T_POS = 1, L_LN1W = 0, L_LN1B, L_INW, L_INB, L_OUTW, L_OUTB, L_LN2W, L_LN2B, L_FC1W, L_FC1B,
       L_FC2W, L_FC2B, L_COUNT };

int arena_layout(const mgea_decoder_config& c, std::vector<int64_t>* offs, int64_t* total) {
    const int64_t V = c.vocab, C = c.d_model, F = c.d_ff, L = c.seq_len;
    std::vector<int64_t> sizes;
    sizes.push_back(V * C);
    sizes.push_back(L * C);
    for (int i = 0; i < c.n_layer; ++i) {
        const int64_t s[L_COUNT] = {C, C, 3 * C * C, 3 * C, C * C, C, C, C, F * C, F, C * F, C};
        for (int j = 0; j < L_COUNT; ++j) sizes.push_back(s[j]);
    }
    sizes.push_back(V * C);
    sizes.push_back(V);
    int64_t o = 0;
    if (offs) offs->clear();
    for (int64_t s : sizes) {
        if (offs) offs->push_back(o);
        o += round_up(s, 64);
    }
    *total = o;
    return (int)sizes.size();
}

int validate(const mgea_decoder_config* c) {
    MGEA_REQUIRE(c, MGEA_EINVAL, "decoder config is NULL");
    MGEA_REQUIRE(c->vocab > 0 && c->seq_len > 0 && c->d_model > 0 && c->n_head > 0 && c->n_layer > 0 && c->d_ff > 0,
                 MGEA_EINVAL, "decoder config: non-positive dimension");
    MGEA_REQUIRE(c->d_model % c->n_head == 0, MGEA_EINVAL, "d_model %d not divisible by n_head %d", c->d_model, c->n_head);
    const int dh = c->d_model / c->n_head;
    // 96 = a 768-wide checkpoint under the reference's hard-coded 8 heads (api_cache.py:112)
    MGEA_REQUIRE(dh == 32 || dh == 64 || dh == 96, MGEA_EINVAL, "head_dim %d not supported (32, 64 or 96)", dh);
    MGEA_REQUIRE(c->d_model % 32 == 0 && c->d_ff % 32 == 0, MGEA_EINVAL, "d_model and d_ff must be multiples of 32");
    MGEA_REQUIRE(c->d_model <= 4096, MGEA_EINVAL, "d_model > 4096 not supported");
    MGEA_REQUIRE(c->max_batch > 0 && c->max_ctx > 0, MGEA_EINVAL, "max_batch / max_ctx must be positive");
    // the sampler keeps a row of logits in registers; refuse at create rather than at the first sampled generate()
    MGEA_REQUIRE(c->vocab <= MGEA_SAMPLER_MAX_VOCAB, MGEA_EINVAL, "vocab %d exceeds the sampler's limit of %d", c->vocab,
                 MGEA_SAMPLER_MAX_VOCAB);
    MGEA_REQUIRE(c->dtype == MGEA_DTYPE_F32 || c->dtype == MGEA_DTYPE_F16, MGEA_EINVAL,
                 "decoder dtype %d not supported (MGEA_DTYPE_F32 or MGEA_DTYPE_F16)", c->dtype);
    MGEA_REQUIRE(c->block_mode == MGEA_BLOCK_PRELN_GELU || c->block_mode == MGEA_BLOCK_POSTLN_RELU, MGEA_EINVAL, "bad block_mode");
    if (c->dtype == MGEA_DTYPE_F16) {   // the fp16 mode lives on the fused decode path (tiled fp16 matrices, fp16 KV pages)
        MGEA_REQUIRE(c->block_mode == MGEA_BLOCK_PRELN_GELU, MGEA_EINVAL, "MGEA_DTYPE_F16 needs the KV-cache block mode");
        MGEA_REQUIRE((c->d_model % 128) == 0 && c->d_model >= 256 && c->d_model <= 1024, MGEA_EINVAL,
                     "MGEA_DTYPE_F16 needs d_model in 256..1024, a multiple of 128 (got %d)", c->d_model);
        MGEA_REQUIRE(dh == 32 || dh == 64, MGEA_EINVAL, "MGEA_DTYPE_F16 needs head_dim 32 or 64 (got %d)", dh);
    }
    return MGEA_OK;
}
}  // namespace

struct mgea_decoder {
    mgea_decoder_config cfg{};
    const float* arena = nullptr;       // what every kernel reads: the caller's arena (f32) or arena_own (f16)
    // MGEA_DTYPE_F16: the model served is "the reference with its five projection-matrix kinds rounded to fp16".  arena_own is
    // a private fp32 copy of the caller's arena (arena_src) with exactly those tensors rounded, so that every path that reads
    // row-major fp32 weights (prefill beyond 512 rows, slab fallback) serves the SAME model as the fp16 tiles of the decode path.
    const float* arena_src = nullptr;
    float* arena_own = nullptr;
    int64_t arena_total = 0;
    bool f16 = false;
    std::vector<int64_t> off;
    int dh = 0;
    std::mutex mu;

    // KV pool
    KvPool kv{};
    int pages_per_row_cap = 0;  // ceil(max_ctx / 64)
    int max_pages = 0;          // page-table row stride (== pages_per_row_cap)
    int32_t* page_table = nullptr;
    // per-row state
    int32_t *ctx_len = nullptr, *cur_ids = nullptr, *done = nullptr, *row_step = nullptr, *n_done = nullptr,
            *sampled = nullptr, *ids_hist = nullptr;
    int ids_hist_stride = 0;
    // host mirror
    int cur_batch = 0, reserved_len = 0, host_max_len = 0, host_min_len = 0;
    // workspace
    int64_t ws_tokens = 0;
    float *x = nullptr, *xn = nullptr, *qkv = nullptr, *att = nullptr, *hbuf = nullptr, *slabs = nullptr,
          *logits = nullptr, *stats = nullptr, *pmax_val = nullptr;
    int32_t* pmax_idx = nullptr;
    bool no_graph = false;       // MGEA_DECODER_NOGRAPH=1: launch every step eagerly (rocprofv3 --pmc runs)
    bool no_gemv = false;        // MGEA_DECODER_NOGEMV=1: keep the MFMA skinny GEMMs for batches of <= 2 rows too (A/B)
    bool force_unfused = false;  // MGEA_DECODER_UNFUSED=1: keep the 9-launch-per-layer path (A/B and fallback)
    int64_t slab_cap = 0;
    // Captured decode-step graphs, one per (batch, greedy | sampled).  Everything a step reads besides its structure
    // lives in device memory (per-row state, page table, and the sampler's scalars in samp_dev), so a request with a
    // new seed / temperature / top-k / top-p / EOS id replays an existing graph: no capture, no instantiate.
    struct GraphEntry { int batch; bool greedy; int steps; hipGraph_t graph; hipGraphExec_t exec; int64_t nodes; uint64_t last_use; };
    std::vector<GraphEntry> graphs;
    uint64_t use_clock = 0;
    SamplerParams* samp_dev = nullptr;
    int32_t* err_flag = nullptr;   // sticky device flags (bit 0: a token id outside the vocabulary was clamped)
    AttnSplit attn_split{};        // scratch of the split-context decode attention (small batches; attn_paged.hip)
    int64_t counters[8] = {0};
    // optional per-kernel-class timing with HIP events on the launch stream (bench.py roofline leg)
    int prof_stride = 0;  // 0 = off; n = time every n-th decode step of generate(), run eagerly
    bool prof_now = false;
    struct ProfRec { hipEvent_t a, b; int cls; };
    std::vector<ProfRec> prof;

    const float* w(int idx) const { return arena + off[idx]; }
    const float* lw(int layer, int j) const { return arena + off[2 + layer * L_COUNT + j]; }
    const float* head_w() const { return arena + off[2 + cfg.n_layer * L_COUNT]; }
    const float* head_b() const { return arena + off[3 + cfg.n_layer * L_COUNT]; }
    // fragment-ordered copies of the five matrix kinds for the fused decode path (common.h: launch_tile_weights);
    // index 4 * layer + {0 in_proj, 1 out_proj, 2 fc1, 3 fc2}, then the head
    void* wt = nullptr;                 // fp32 fragments (f32) or _Float16 fragments (f16); offsets in elements
    std::vector<int64_t> wt_off;
    const float* wt_at(int64_t off) const {
        return reinterpret_cast<const float*>(static_cast<const char*>(wt) + off * (f16 ? 2 : 4));
    }
    // LayerNorm folded into in_proj (ln1) and fc1 (ln2): those two tiled matrices hold gamma * W, and lnv holds per layer
    // [c1 3C][c2 3C][c1 F][c2 F] (common.h: launch_ln_fold)
    float* lnv = nullptr;
    const float* qkv_c1(int l) const { return lnv + (int64_t)l * (6 * cfg.d_model + 2 * cfg.d_ff); }
    const float* qkv_c2(int l) const { return qkv_c1(l) + 3 * cfg.d_model; }
    const float* fc1_c1(int l) const { return qkv_c1(l) + 6 * cfg.d_model; }
    const float* fc1_c2(int l) const { return fc1_c1(l) + cfg.d_ff; }
    const float* tw(int layer, int j) const { return wt_at(wt_off[4 * layer + j]); }
    const float* head_tw() const { return wt_at(wt_off[4 * cfg.n_layer]); }

    // MGEA_DTYPE_F16, big-batch prefill on the f16 matrix cores (run_prefill16 below): fp16 activations, row-major fp16 matrices
    // (in_proj and fc1 with their LayerNorm's gamma folded in), statistics tables.  Allocated at the first such prefill.
    struct Prefill16 {
        int64_t rows = 0;
        void *x0 = nullptr, *x1 = nullptr, *qkv = nullptr, *att = nullptr, *hid = nullptr, *w = nullptr;
        float *rowstat = nullptr, *stats_part = nullptr, *ident = nullptr, *vec = nullptr;
        int32_t* mask = nullptr;
        bool weights_ready = false;
        std::vector<int64_t> w_off;      // elements: per layer in_proj', out_proj, fc1', fc2; then the head
    } p16;
    const char* p16_w(int i) const { return static_cast<const char*>(p16.w) + p16.w_off[i] * 2; }
    float* p16_vec(int l, int which) const {   // 0 c1(in_proj) 1 c2(in_proj) 2 c1(fc1) 3 c2(fc1)
        float* base = p16.vec + (int64_t)l * (6 * cfg.d_model + 2 * cfg.d_ff);
        return which == 0 ? base : which == 1 ? base + 3 * cfg.d_model : which == 2 ? base + 6 * cfg.d_model : base + 6 * cfg.d_model + cfg.d_ff;
    }
};

namespace {

enum { PC_GEMM = 0, PC_ROWOP = 1, PC_ATTN_PAGED = 2, PC_ATTN_DENSE = 3, PC_SAMPLE = 4, PC_COUNT = 5 };

struct ProfScope {
    mgea_decoder* h;
    hipStream_t st;
    bool on;
    hipEvent_t a = nullptr, b = nullptr;
    int cls;
    ProfScope(mgea_decoder* h_, int cls_, hipStream_t st_) : h(h_), st(st_), on(h_->prof_now), cls(cls_) {
        if (on) {
            on = hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess;
            if (on) (void)hipEventRecord(a, st);
        }
    }
    ~ProfScope() {
        if (on) {
            (void)hipEventRecord(b, st);
            h->prof.push_back({a, b, cls});
        }
    }
};
#define PROF(cls, call)                \
    do {                               \
        ProfScope _ps(h, cls, st);     \
        MGEA_TRY(call);                \
    } while (0)

void free_ws(mgea_decoder* h) {
    float** p[] = {&h->x, &h->xn, &h->qkv, &h->att, &h->hbuf, &h->slabs, &h->logits, &h->stats, &h->pmax_val};
    if (h->pmax_idx) (void)hipFree(h->pmax_idx);
    h->pmax_idx = nullptr;
    for (auto q : p) {
        if (*q) (void)hipFree(*q);
        *q = nullptr;
    }
    h->ws_tokens = 0;
    h->slab_cap = 0;
}

void drop_graphs(mgea_decoder* h) {
    for (auto& g : h->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    h->graphs.clear();
    h->counters[4] = 0;
}

int64_t slab_need(const mgea_decoder_config& c, int M) {
    const int C = c.d_model, F = c.d_ff, V = c.vocab;
    const int shapes[5][2] = {{3 * C, C}, {C, C}, {F, C}, {C, F}, {V, C}};
    int64_t need = 0;
    const int Mh = M > 4096 ? 4096 : M;  // the head runs in row chunks of <= 4096
    for (int i = 0; i < 5; ++i) {
        const int m = (i == 4) ? Mh : M;
        const int64_t s = (int64_t)pick_split_k(m, shapes[i][0], shapes[i][1]) * slab_floats(m, shapes[i][0]);
        need = s > need ? s : need;
    }
    return need;
}

int ensure_ws(mgea_decoder* h, int64_t M) {
    M = round_up(M, 64);   // the k-tiled buffers of the fused path are whole 64-row groups
    if (M <= h->ws_tokens) return MGEA_OK;
    MGEA_CHECK_HIP(hipDeviceSynchronize());
    free_ws(h);
    drop_graphs(h);  // captured pointers die with the old workspace
    const int C = h->cfg.d_model, F = h->cfg.d_ff;
    int64_t slab = slab_need(h->cfg, (int)M);
    const int64_t s64 = slab_need(h->cfg, 64);
    slab = slab > s64 ? slab : s64;
#define ALLOC(ptr, n)                                                                                   \
    if (hipMalloc((void**)&(ptr), (size_t)(n) * sizeof(float)) != hipSuccess) {                        \
        set_error("decoder workspace: out of device memory (%lld floats)", (long long)(n));            \
        free_ws(h);                                                                                     \
        return MGEA_ENOMEM;                                                                             \
    }
    ALLOC(h->x, M * C);
    ALLOC(h->xn, M * C);
    ALLOC(h->qkv, M * 3 * C);
    ALLOC(h->att, M * C);
    ALLOC(h->hbuf, M * F);
    ALLOC(h->slabs, slab);
    ALLOC(h->logits, (int64_t)h->cfg.max_batch * h->cfg.vocab);
    const int64_t fr = M < MGEA_FUSED_MAX_ROWS ? M : MGEA_FUSED_MAX_ROWS;   // rows of the fused path
    ALLOC(h->stats, fr * (C / 16 + 1) * 2);
    ALLOC(h->pmax_val, fr * ceil_div(h->cfg.vocab, 16));
    if (hipMalloc((void**)&h->pmax_idx, (size_t)fr * ceil_div(h->cfg.vocab, 16) * sizeof(int32_t)) != hipSuccess) {
        set_error("decoder workspace: out of device memory");
        free_ws(h);
        return MGEA_ENOMEM;
    }
#undef ALLOC
    h->slab_cap = slab;
    h->ws_tokens = M;
    return MGEA_OK;
}

// gemm + slab bookkeeping
int gemm(mgea_decoder* h, const float* A, int lda, const float* W, int M, int N, int K, int* S, hipStream_t st) {
    const int s = pick_split_k(M, N, K, h->slab_cap);
    MGEA_REQUIRE((int64_t)s * slab_floats(M, N) <= h->slab_cap, MGEA_ECAPACITY, "internal: slab workspace too small");
    ProfScope _ps(h, PC_GEMM, st);
    const int rc = launch_gemm_f32(A, lda, W, K, h->slabs, M, N, K, s, st);
    if (rc < 0) return rc;
    *S = rc;
    return MGEA_OK;
}

// The NL blocks over M = B*T rows.  use_cache_attn: attention over the paged cache (decode /
// extend); otherwise dense attention inside the qkv buffer (prefill with empty cache, twin mode).
// kv_only_last: the caller drops the logits (the prompt prefill of sample_kvcache, api_cache.py:163: `_, past = model(idx)`), so nothing
// reads the last block's output: once its K | V are in the cache the pass is over -- no attention, out-projection or MLP for that block.
int run_blocks(mgea_decoder* h, int B, int T, const int32_t* lens, bool use_cache_attn, bool scatter,
               hipStream_t st, bool kv_only_last = false) {
    const auto& c = h->cfg;
    const int C = c.d_model, F = c.d_ff, M = B * T;
    const bool post = c.block_mode == MGEA_BLOCK_POSTLN_RELU;
    for (int l = 0; l < c.n_layer; ++l) {
        int S = 1;
        const float* a_in = post ? h->x : h->xn;
        // big prefill: bias inside the QKV GEMM (qkv written once, no slab) and a scatter that only reads its K | V columns -- 0.5 GB less
        // traffic per block at [64, 1024].  Only where the slab form would not split K either (>= 256 tiles): the same sums.  (Rows past a
        // ragged prompt's length then hold the projection of their padding token instead of zeros; nothing reads them: the attention
        // masks them as keys by `lens` and their own outputs are ignored.)
        if (scatter && M > 64 && (int64_t)ceil_div(M, 128) * ceil_div(3 * C, 128) >= 256) {
            PROF(PC_GEMM, launch_gemm_f32_bias_act(a_in, C, h->lw(l, L_INW), C, h->lw(l, L_INB), h->qkv, 3 * C, M, 3 * C, C, ACT_NONE, st));
            PROF(PC_ROWOP, launch_qkv_scatter(h->qkv, 1, 0, 3 * C, nullptr, nullptr, h->kv, l, h->page_table, h->max_pages, h->ctx_len, lens, B, T, C, st));
        } else {
        MGEA_TRY(gemm(h, a_in, C, h->lw(l, L_INW), M, 3 * C, C, &S, st));
        KvPool kv = h->kv;
        if (!scatter) kv.base = nullptr;
        if (scatter) {
            PROF(PC_ROWOP, launch_qkv_scatter(h->slabs, S, slab_floats(M, 3 * C), (int)slab_ld(3 * C), h->lw(l, L_INB),
                                        h->qkv, h->kv, l, h->page_table, h->max_pages, h->ctx_len, lens, B, T, C, st));
        } else {
            PROF(PC_ROWOP, launch_bias_act(h->slabs, S, slab_floats(M, 3 * C), (int)slab_ld(3 * C), h->lw(l, L_INB), h->qkv,
                                     3 * C, M, 3 * C, ACT_NONE, st));
        }
        }
        if (kv_only_last && scatter && l + 1 == c.n_layer) break;
        if (use_cache_attn) {
            PROF(PC_ATTN_PAGED, launch_attn_paged(h->qkv, h->kv, l, h->page_table, h->max_pages, h->ctx_len, lens, h->att, B, T,
                                       C, 0, st, &h->attn_split));
        } else {
            PROF(PC_ATTN_DENSE, launch_attn_dense(h->qkv, lens, nullptr, h->att, B, T, c.n_head, h->dh, 0, st));
        }
        MGEA_TRY(gemm(h, h->att, C, h->lw(l, L_OUTW), M, C, C, &S, st));
        if (post) {
            PROF(PC_ROWOP, launch_bias_res_ln(h->slabs, S, slab_floats(M, C), (int)slab_ld(C), h->lw(l, L_OUTB), h->x,
                                        nullptr, h->lw(l, L_LN1W), h->lw(l, L_LN1B), c.ln_eps, M, C, 1, st));
        } else {
            PROF(PC_ROWOP, launch_bias_res_ln(h->slabs, S, slab_floats(M, C), (int)slab_ld(C), h->lw(l, L_OUTB), h->x,
                                        h->xn, h->lw(l, L_LN2W), h->lw(l, L_LN2B), c.ln_eps, M, C, 0, st));
        }
        if (gemm_direct_epilogue_ok((int)M, F)) {   // bias + activation inside the GEMM epilogue (no slab round trip)
            PROF(PC_GEMM, launch_gemm_f32_bias_act(post ? h->x : h->xn, C, h->lw(l, L_FC1W), C, h->lw(l, L_FC1B), h->hbuf, F, M, F,
                                                   C, post ? ACT_RELU : ACT_GELU, st));
        } else {
            MGEA_TRY(gemm(h, post ? h->x : h->xn, C, h->lw(l, L_FC1W), M, F, C, &S, st));
            PROF(PC_ROWOP, launch_bias_act(h->slabs, S, slab_floats(M, F), (int)slab_ld(F), h->lw(l, L_FC1B), h->hbuf, F, M, F,
                                     post ? ACT_RELU : ACT_GELU, st));
        }
        MGEA_TRY(gemm(h, h->hbuf, F, h->lw(l, L_FC2W), M, C, F, &S, st));
        if (post) {
            PROF(PC_ROWOP, launch_bias_res_ln(h->slabs, S, slab_floats(M, C), (int)slab_ld(C), h->lw(l, L_FC2B), h->x,
                                        nullptr, h->lw(l, L_LN2W), h->lw(l, L_LN2B), c.ln_eps, M, C, 1, st));
        } else {
            const bool last = l + 1 == c.n_layer;
            PROF(PC_ROWOP, launch_bias_res_ln(h->slabs, S, slab_floats(M, C), (int)slab_ld(C), h->lw(l, L_FC2B), h->x,
                                        h->xn, last ? nullptr : h->lw(l + 1, L_LN1W),
                                        last ? nullptr : h->lw(l + 1, L_LN1B), c.ln_eps, M, C, 0, st));
        }
    }
    return MGEA_OK;
}

// Everything one fused decode pass touches besides the weights.
struct Bufs {
    float *x, *qkv, *att, *hbuf, *stats, *pmax_val;
    int32_t *pmax_idx, *page_table, *ctx_len, *cur_ids, *done, *row_step, *sampled, *ids_hist;
    float* logits;
};

Bufs main_bufs(mgea_decoder* h) {
    return Bufs{h->x, h->qkv, h->att, h->hbuf, h->stats, h->pmax_val, h->pmax_idx, h->page_table, h->ctx_len, h->cur_ids,
                h->done, h->row_step, h->sampled, h->ids_hist, h->logits};
}

// Fused path for M = B*T <= MGEA_FUSED_MAX_ROWS rows in the KV-cache block mode: 5 launches per layer
// (gemm_skinny.hip); x carries per-row LayerNorm partial statistics between kernels.
bool fused_geometry(const mgea_decoder_config& c) {
    return c.block_mode == MGEA_BLOCK_PRELN_GELU && (c.d_model % 128) == 0 && c.d_model >= 256 && c.d_model <= 1024;
}
bool fused_ok(const mgea_decoder* h, int M) {
    return fused_geometry(h->cfg) && M <= MGEA_FUSED_MAX_ROWS && M <= h->ws_tokens && !h->force_unfused && h->wt;
}

// single-token decode steps of <= 2 rows (the reference's serving case is B = 1): wave-level dot products on the
// row-major arena weights instead of 16 x 16 MFMA tiles (gemv_small.hip)
bool gemv_ok(const mgea_decoder* h, int M, int T, const int32_t* lens, bool use_cache_attn) {
    const auto& c = h->cfg;
    return !h->no_gemv && !h->f16 && T == 1 && !lens && use_cache_attn && gemv_shape_ok(M, c.d_model, c.d_model) &&
           gemv_shape_ok(M, c.d_model, c.d_ff);
}

int run_blocks_fused(mgea_decoder* h, const Bufs& u, int B, int T, const int32_t* lens, bool use_cache_attn, hipStream_t st,
                     bool kv_only_last = false) {
    const auto& c = h->cfg;
    const int C = c.d_model, F = c.d_ff, M = B * T;
    const bool gv = gemv_ok(h, M, T, lens, use_cache_attn);
    int n_part = 2, part_cnt = C / 2;  // the embedding kernel leaves the whole-row statistics as two equal halves
    for (int l = 0; l < c.n_layer; ++l) {
        SkinnyArgs a{};
        a.M = M; a.eps = c.ln_eps; a.w_f16 = h->f16;
        if (h->f16) a.ln_g = h->lw(l, L_LN1W);
        // ln1 + in_proj + KV append
        a.A = u.x; a.lda = C; a.W = h->tw(l, 0); a.bias = h->qkv_c2(l); a.N = 3 * C; a.K = C;
        a.ln_c1 = h->qkv_c1(l); a.stats_in = u.stats; a.n_part = n_part; a.part_cnt = part_cnt;
        a.out = u.qkv; a.ldo = 3 * C;
        a.pool = h->kv; a.layer = l; a.page_table = u.page_table; a.max_pages = h->max_pages; a.ctx_len = u.ctx_len;
        a.lens = lens; a.T = T; a.C = C;
        if (gv) {
            a.W = h->lw(l, L_INW); a.bias = h->lw(l, L_INB); a.ln_c1 = nullptr; a.ln_g = h->lw(l, L_LN1W); a.ln_b = h->lw(l, L_LN1B);
            PROF(PC_GEMM, launch_gemv(EPI_QKV, a, st));
        } else {
            PROF(PC_GEMM, launch_skinny(EPI_QKV, a, st));
        }
        if (kv_only_last && l + 1 == c.n_layer) break;       // (run_blocks: the logits are dropped, the last block's K | V are appended)
        if (use_cache_attn) {
            PROF(PC_ATTN_PAGED, launch_attn_paged(u.qkv, h->kv, l, u.page_table, h->max_pages, u.ctx_len, lens, u.att, B, T, C, 1, st, &h->attn_split));
        } else {
            PROF(PC_ATTN_DENSE, launch_attn_dense(u.qkv, lens, nullptr, u.att, B, T, c.n_head, h->dh, 1, st));
        }
        // out_proj + residual (+ stats for ln2)
        SkinnyArgs o{};
        o.M = M; o.eps = c.ln_eps; o.w_f16 = h->f16;
        o.A = u.att; o.lda = C; o.W = h->tw(l, 1); o.bias = h->lw(l, L_OUTB); o.N = C; o.K = C;
        o.out = u.x; o.ldo = C; o.stats_out = u.stats;
        if (gv) {
            o.W = h->lw(l, L_OUTW);
            PROF(PC_GEMM, launch_gemv(EPI_RES, o, st));
        } else {
            PROF(PC_GEMM, launch_skinny(EPI_RES, o, st));
        }
        n_part = C / 16; part_cnt = 16;
        // ln2 + mlp.0 + GELU
        SkinnyArgs f{};
        f.M = M; f.eps = c.ln_eps; f.w_f16 = h->f16;
        if (h->f16) f.ln_g = h->lw(l, L_LN2W);
        f.A = u.x; f.lda = C; f.W = h->tw(l, 2); f.bias = h->fc1_c2(l); f.N = F; f.K = C;
        f.ln_c1 = h->fc1_c1(l); f.stats_in = u.stats; f.n_part = n_part; f.part_cnt = part_cnt;
        f.out = u.hbuf; f.ldo = F; f.act = ACT_GELU;
        if (gv) {
            f.W = h->lw(l, L_FC1W); f.bias = h->lw(l, L_FC1B); f.ln_c1 = nullptr; f.ln_g = h->lw(l, L_LN2W); f.ln_b = h->lw(l, L_LN2B);
            PROF(PC_GEMM, launch_gemv(EPI_ACT, f, st));
        } else {
            PROF(PC_GEMM, launch_skinny(EPI_ACT, f, st));
        }
        // mlp.2 + residual (+ stats for the next ln1)
        SkinnyArgs r{};
        r.M = M; r.eps = c.ln_eps; r.w_f16 = h->f16;
        r.A = u.hbuf; r.lda = F; r.W = h->tw(l, 3); r.bias = h->lw(l, L_FC2B); r.N = C; r.K = F;
        r.out = u.x; r.ldo = C; r.stats_out = u.stats;
        if (gv) {
            r.W = h->lw(l, L_FC2W);
            PROF(PC_GEMM, launch_gemv(EPI_RES, r, st));
        } else {
            PROF(PC_GEMM, launch_skinny(EPI_RES, r, st));
        }
    }
    return MGEA_OK;
}

// pd != NULL: the EOS id (like the other sampler scalars) is read from that device record -- the captured graph's form
StepState step_state(mgea_decoder* h, const Bufs& u, int eos, const SamplerParams* pd) {
    StepState s;
    s.cur_ids = u.cur_ids;
    s.ctx_len = u.ctx_len;
    s.done = u.done;
    s.row_step = u.row_step;
    s.n_done = h->n_done;
    s.ids_out = u.ids_hist;
    s.n_steps = h->ids_hist_stride;
    s.eos_id = eos;
    s.params = pd;
    return s;
}

// One fused decode step (T = 1) over the rows of `u`.
// primed: x already holds the embedding (+ LN statistics) of cur_ids -- generate() keeps that invariant by
// fusing the next step's embedding into this step's tail, so a replayed step is 32 launches.
// pd: device-resident sampler scalars (generate()) or NULL (mgea_decoder_step: `sc` by value).  Only sc.top_k == 1
// (greedy or not) shapes the launch sequence.
int enqueue_step_fused(mgea_decoder* h, const Bufs& u, int B, const mgea_sampler_config& sc, const SamplerParams* pd,
                       float* logits_out, hipStream_t st, bool primed) {
    const auto& c = h->cfg;
    const int C = c.d_model, V = c.vocab;
    const bool greedy = sc.top_k == 1;
    // [embed,] 6 x (qkv, attention, out-proj, fc1, fc2), head (+ per-tile argmax), finalize [+ next embed]
    const int abs_pos = c.pos_mode == MGEA_POS_ABSOLUTE;
    if (!primed)
        PROF(PC_ROWOP, launch_embed_stats(u.cur_ids, nullptr, u.ctx_len, h->w(T_TOK), h->w(T_POS), u.x, u.stats, B, 1, C, V,
                                          c.seq_len, abs_pos, h->err_flag, st));
    MGEA_TRY(run_blocks_fused(h, u, B, 1, nullptr, true, st));
    SkinnyArgs a{};
    a.w_f16 = h->f16;
    a.M = B; a.A = u.x; a.lda = C; a.W = h->head_tw(); a.bias = h->head_b(); a.N = V; a.K = C;
    a.out = logits_out ? logits_out : (greedy ? nullptr : u.logits);
    a.ldo = V; a.pmax_val = u.pmax_val; a.pmax_idx = u.pmax_idx;
    if (gemv_ok(h, B, 1, nullptr, true) && gemv_shape_ok(B, V, C)) {   // partial count = ceil(V / 16) = skinny_logits_tiles(B <= 2, V, C)
        a.W = h->head_w();
        PROF(PC_GEMM, launch_gemv(EPI_LOGITS, a, st));
    } else {
        PROF(PC_GEMM, launch_skinny(EPI_LOGITS, a, st));
    }
    if (greedy && primed) {
        PROF(PC_SAMPLE, launch_argmax_advance_embed(u.pmax_val, u.pmax_idx, skinny_logits_tiles(B, V, C), step_state(h, u, sc.eos_id, pd),
                                                    u.sampled, h->w(T_TOK), h->w(T_POS), u.x, u.stats, B, C, V, c.seq_len,
                                                    abs_pos, st));
    } else if (greedy) {
        PROF(PC_SAMPLE, launch_argmax_advance(u.pmax_val, u.pmax_idx, skinny_logits_tiles(B, V, C), step_state(h, u, sc.eos_id, pd),
                                              u.sampled, B, st));
    } else {
        if (primed) {   // sampler + loop bookkeeping + next step's embedding in one launch
            TailArgs t{step_state(h, u, sc.eos_id, pd), h->w(T_TOK), h->w(T_POS), u.x, u.stats, C, V, c.seq_len, abs_pos};
            PROF(PC_SAMPLE, launch_sample(a.out, B, V, sc, pd, u.row_step, 0, u.sampled, nullptr, st, &t));
        } else {
            PROF(PC_SAMPLE, launch_sample(a.out, B, V, sc, pd, u.row_step, 0, u.sampled, nullptr, st));
            PROF(PC_ROWOP, launch_advance(u.sampled, step_state(h, u, sc.eos_id, pd), B, st));
        }
    }
    return MGEA_OK;
}

// one decode step on cur_ids (T = 1) for the whole batch; logits_out optional
int enqueue_step(mgea_decoder* h, int B, const mgea_sampler_config& sc, const SamplerParams* pd, float* logits_out,
                 hipStream_t st, bool primed = false) {
    const auto& c = h->cfg;
    const int C = c.d_model, V = c.vocab;
    const bool post = c.block_mode == MGEA_BLOCK_POSTLN_RELU;
    const bool greedy = sc.top_k == 1;
    if (fused_ok(h, B)) return enqueue_step_fused(h, main_bufs(h), B, sc, pd, logits_out, st, primed);
    const Bufs u = main_bufs(h);
    PROF(PC_ROWOP, launch_embed_ln(h->cur_ids, nullptr, h->ctx_len, h->w(T_TOK), h->w(T_POS), h->x, h->xn,
                             post ? nullptr : h->lw(0, L_LN1W), post ? nullptr : h->lw(0, L_LN1B), c.ln_eps, B, 1, C,
                             V, c.seq_len, c.pos_mode == MGEA_POS_ABSOLUTE, h->err_flag, st));
    MGEA_TRY(run_blocks(h, B, 1, nullptr, true, true, st));
    int S = 1;
    MGEA_TRY(gemm(h, h->x, C, h->head_w(), B, V, C, &S, st));
    float* lg = logits_out ? logits_out : (greedy ? nullptr : h->logits);
    PROF(PC_SAMPLE, launch_logits_argmax(h->slabs, S, slab_floats(B, V), (int)slab_ld(V), h->head_b(), lg, B, V,
                                  greedy ? h->sampled : nullptr, st));
    if (!greedy) PROF(PC_SAMPLE, launch_sample(lg, B, V, sc, pd, h->row_step, 0, h->sampled, nullptr, st));
    PROF(PC_ROWOP, launch_advance(h->sampled, step_state(h, u, sc.eos_id, pd), B, st));
    return MGEA_OK;
}

// The decode step of generate(): x arrives primed on the fused path.
int enqueue_gen_step(mgea_decoder* h, int B, const mgea_sampler_config& sc, hipStream_t st) {
    if (!fused_ok(h, B)) return enqueue_step(h, B, sc, h->samp_dev, nullptr, st, false);
    return enqueue_step_fused(h, main_bufs(h), B, sc, h->samp_dev, nullptr, st, true);
}

// embedding (+ LN statistics) of cur_ids into the buffers the next generate() step will read
int prime_gen(mgea_decoder* h, int B, hipStream_t st) {
    if (!fused_ok(h, B)) return MGEA_OK;
    const auto& c = h->cfg;
    const Bufs u = main_bufs(h);
    return launch_embed_stats(u.cur_ids, nullptr, u.ctx_len, h->w(T_TOK), h->w(T_POS), u.x, u.stats, B, 1, c.d_model, c.vocab,
                              c.seq_len, c.pos_mode == MGEA_POS_ABSOLUTE, h->err_flag, st);
}

// The captured decode step for (B, greedy): from the cache, or captured + instantiated now (least recently used
// entry evicted beyond MAX_GRAPHS).
constexpr size_t MAX_GRAPHS = 16;   // two per (batch, greedy | sampled): the single step and the 8-step graph
// steps > 1: that many consecutive decode steps in one graph (switch decoder_graph_steps; the per-step state is in device memory, so the
// steps of a graph are as independent of the host as the graphs are of each other)
int step_graph(mgea_decoder* h, int B, const mgea_sampler_config& sc, hipStream_t st, hipGraphExec_t* out, int steps = 1) {
    const bool greedy = sc.top_k == 1;
    for (auto& g : h->graphs)
        if (g.batch == B && g.greedy == greedy && g.steps == steps) {
            g.last_use = ++h->use_clock;
            if (steps == 1) h->counters[0] = g.nodes;
            *out = g.exec;
            return MGEA_OK;
        }
    if (h->graphs.size() >= MAX_GRAPHS) {
        size_t lru = 0;
        for (size_t i = 1; i < h->graphs.size(); ++i)
            if (h->graphs[i].last_use < h->graphs[lru].last_use) lru = i;
        MGEA_CHECK_HIP(hipStreamSynchronize(st));   // an evicted exec may still be replaying
        (void)hipGraphExecDestroy(h->graphs[lru].exec);
        (void)hipGraphDestroy(h->graphs[lru].graph);
        h->graphs.erase(h->graphs.begin() + (long)lru);
    }
    MGEA_CHECK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    int rc = MGEA_OK;
    for (int k = 0; k < steps && rc == MGEA_OK; ++k) rc = enqueue_gen_step(h, B, sc, st);
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(st, &g);
    if (rc != MGEA_OK) {
        if (g) (void)hipGraphDestroy(g);
        return rc;
    }
    MGEA_CHECK_HIP(e);
    hipGraphExec_t ex = nullptr;
    const hipError_t ei = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    if (ei != hipSuccess) {
        (void)hipGraphDestroy(g);
        MGEA_CHECK_HIP(ei);
    }
    size_t nn = 0;
    (void)hipGraphGetNodes(g, nullptr, &nn);
    h->graphs.push_back({B, greedy, steps, g, ex, (int64_t)nn, ++h->use_clock});
    if (steps == 1) h->counters[0] = (int64_t)nn;
    h->counters[2] += 1;   // lifetime captures + instantiations
    h->counters[4] = (int64_t)h->graphs.size();
    *out = ex;
    return MGEA_OK;
}

// ---- fp16 engines: prefill of a big batch on the f16 matrix cores --------------------------------------------------------------
// The exact-fp32 kernels run a [64, 1024] prefill at 0.67 of THEIR peak, which is 1/16 of the f16 MFMA rate.  When the cache is empty
// and the batch is big enough for the persistent 256 x 256 GEMM (M / 256 * d_model / 256 >= 256 tiles), an fp16 engine runs the GPT
// block (api_cache.py:51-74, 87-106) on the kernels of the bf16 DistilBERT path with _Float16 operands instead:
//   x (fp16) + (mean, rstd) per row  ->  per layer:
//     qkv  = rstd (x W_in'^T - mean c1) + c2            W_in' = f16(W_in diag(ln1 gamma)): LayerNorm folded into the GEMM (epilogue 3)
//     K | V of the real tokens -> fp16 KV pages;  att = flash attention over qkv (non-causal: the reference has no mask), fp16
//     x'   = att W_out^T + b + x, + row statistics      (epilogue 5, identity tables: the residual is the raw x)
//     hid  = gelu(rstd' (x' W_fc1'^T - mean' c1') + c2')  (epilogue 4)
//     x    = hid W_fc2^T + b + x', + row statistics     (epilogue 5)
//   logits = x W_head^T + b as fp32                      (epilogue 6, N = vocab)
// Everything between two GEMM inputs is fp16 (the residual stream too: 11 significant bits, against bf16's 8 in the DistilBERT mode);
// accumulation, LayerNorm statistics and the softmax are fp32.  W_in' / W_fc1' round gamma * f16(W) once more (the decode path applies
// gamma to the activations instead): the two paths serve models that differ by one fp16 rounding of those two matrices -- inside
// the fp16 mode's tolerance (tests/test_gpu_f16.py compares both with the oracle on the rounded matrices).
void free_p16(mgea_decoder* h) {
    auto& p = h->p16;
    void* q[] = {p.x0, p.x1, p.qkv, p.att, p.hid, p.w, p.rowstat, p.stats_part, p.ident, p.vec, p.mask};
    for (void* v : q)
        if (v) (void)hipFree(v);
    p = mgea_decoder::Prefill16();
}

bool prefill16_ok(const mgea_decoder* h, int64_t M, bool cache_attn, const float* logits_out) {
    const auto& c = h->cfg;
    if (!h->f16 || cache_attn || c.block_mode != MGEA_BLOCK_PRELN_GELU || !tune(TUNE_DECODER_PREFILL16)) return false;
    if (h->dh != 64 || c.d_model % 256 != 0 || c.d_ff % 256 != 0 || c.d_model > 2048) return false;
    if (logits_out && c.vocab % 4 != 0) return false;
    // worth it from the size at which the smallest GEMM (N = d_model) fills the chip with 256 x 256 tiles; switch value 2 (tests)
    // takes every size the kernels accept
    const int64_t tiles = (M / 256) * (c.d_model / 256);
    return M >= 512 && M < (1ll << 24) && tiles >= (tune(TUNE_DECODER_PREFILL16) == 2 ? 8 : 256);
}

int ensure_p16(mgea_decoder* h, int64_t M, hipStream_t st) {
    auto& p = h->p16;
    const auto& c = h->cfg;
    const int64_t C = c.d_model, F = c.d_ff, V = c.vocab, NL = c.n_layer;
    if (M > p.rows) {
        MGEA_CHECK_HIP(hipDeviceSynchronize());
        void** bufs[] = {&p.x0, &p.x1, &p.qkv, &p.att, &p.hid};
        for (void** b : bufs) { if (*b) (void)hipFree(*b); *b = nullptr; }
        float** fb[] = {&p.rowstat, &p.stats_part, &p.ident};
        for (float** b : fb) { if (*b) (void)hipFree(*b); *b = nullptr; }
        if (p.mask) { (void)hipFree(p.mask); p.mask = nullptr; }
        const int64_t R = round_up(M, 256);
        const bool ok = hipMalloc(&p.x0, R * C * 2) == hipSuccess && hipMalloc(&p.x1, R * C * 2) == hipSuccess &&
                        hipMalloc(&p.qkv, R * 3 * C * 2) == hipSuccess && hipMalloc(&p.att, R * C * 2) == hipSuccess &&
                        hipMalloc(&p.hid, R * F * 2) == hipSuccess && hipMalloc((void**)&p.rowstat, R * 2 * 4) == hipSuccess &&
                        hipMalloc((void**)&p.stats_part, R * (C / 256) * 2 * 4) == hipSuccess &&
                        hipMalloc((void**)&p.ident, (R * 2 + 2 * C) * 4) == hipSuccess && hipMalloc((void**)&p.mask, R * 4) == hipSuccess;
        if (!ok) {
            set_error("decoder: out of device memory for the fp16 prefill workspace (%lld tokens)", (long long)M);
            free_p16(h);
            return MGEA_ENOMEM;
        }
        std::vector<float> idv((size_t)(R * 2 + 2 * C), 0.f);       // (0, 1) per row, then C ones (gamma), then C zeros (beta)
        for (int64_t r = 0; r < R; ++r) idv[(size_t)r * 2 + 1] = 1.f;
        for (int64_t d = 0; d < C; ++d) idv[(size_t)(R * 2 + d)] = 1.f;
        MGEA_CHECK_HIP(hipMemcpy(p.ident, idv.data(), idv.size() * 4, hipMemcpyHostToDevice));
        p.rows = R;
    }
    if (!p.weights_ready) {
        if (!p.w || !p.vec) {
            if (p.w) { (void)hipFree(p.w); p.w = nullptr; }
            if (p.vec) { (void)hipFree(p.vec); p.vec = nullptr; }
            p.w_off.clear();
            int64_t tot = 0;
            for (int l = 0; l < NL; ++l) {
                p.w_off.push_back(tot); tot += 3 * C * C;
                p.w_off.push_back(tot); tot += C * C;
                p.w_off.push_back(tot); tot += F * C;
                p.w_off.push_back(tot); tot += C * F;
            }
            p.w_off.push_back(tot); tot += round_up(V, 256) * C;     // (rows beyond V are never read: the kernel clamps its row index)
            if (hipMalloc(&p.w, tot * 2) != hipSuccess || hipMalloc((void**)&p.vec, NL * (6 * C + 2 * F) * 4) != hipSuccess) {
                // no half-built state: with p.w set and p.vec null the next prefill would skip this block and fold into a null table
                if (p.w) (void)hipFree(p.w);
                if (p.vec) (void)hipFree(p.vec);
                p.w = nullptr; p.vec = nullptr; p.w_off.clear();
                set_error("decoder: out of device memory for the fp16 prefill matrices");
                return MGEA_ENOMEM;
            }
        }
        for (int l = 0; l < NL; ++l) {
            MGEA_TRY(launch_fold_ln_weights_bf16(h->lw(l, L_INW), h->lw(l, L_LN1W), h->lw(l, L_LN1B), h->lw(l, L_INB), (void*)h->p16_w(4 * l + 0),
                                                 h->p16_vec(l, 0), h->p16_vec(l, 1), (int)(3 * C), (int)C, st, 1));
            MGEA_TRY(launch_f32_to_bf16(h->lw(l, L_OUTW), (void*)h->p16_w(4 * l + 1), C * C, st, 1));
            MGEA_TRY(launch_fold_ln_weights_bf16(h->lw(l, L_FC1W), h->lw(l, L_LN2W), h->lw(l, L_LN2B), h->lw(l, L_FC1B), (void*)h->p16_w(4 * l + 2),
                                                 h->p16_vec(l, 2), h->p16_vec(l, 3), (int)F, (int)C, st, 1));
            MGEA_TRY(launch_f32_to_bf16(h->lw(l, L_FC2W), (void*)h->p16_w(4 * l + 3), C * F, st, 1));
        }
        MGEA_TRY(launch_f32_to_bf16(h->head_w(), (void*)h->p16_w(4 * (int)NL), V * C, st, 1));
        p.weights_ready = true;
    }
    return MGEA_OK;
}

int run_prefill16(mgea_decoder* h, const int32_t* ids, const int32_t* lens, int B, int T, float* logits_out, hipStream_t st, bool kv_only_last) {
    const auto& c = h->cfg;
    const int C = c.d_model, F = c.d_ff, V = c.vocab, M = B * T, npart = C / 256;
    MGEA_TRY(ensure_p16(h, M, st));
    auto& p = h->p16;
    void *xc = p.x0, *xo = p.x1;
    const float *id_g = p.ident + p.rows * 2, *id_b = id_g + C;
    PROF(PC_ROWOP, launch_dec_embed_f16(ids, lens, h->ctx_len, h->w(T_TOK), h->w(T_POS), xc, p.rowstat, lens ? p.mask : nullptr, c.ln_eps, B, T,
                                        C, V, c.seq_len, c.pos_mode == MGEA_POS_ABSOLUTE, h->err_flag, st));
    // K | V of the real tokens reach the fp16 KV pages from inside the attention kernel of their layer (bf16.hip: it has those rows in LDS
    // anyway; round 3 ran a scatter kernel per layer that re-read them from the qkv buffer, on a side stream under the attention).  Only a
    // last block that stops at its K | V (kv_only_last: no attention runs) still uses the scatter kernel.
    for (int l = 0; l < c.n_layer; ++l) {
        const bool last = l + 1 == c.n_layer;
        if (last && kv_only_last) {
            // the logits are dropped (run_blocks): of the last block only K | V -- rows C.. of the stacked projection -- then its scatter
            const BfEpiLn qk{p.rowstat, h->p16_vec(l, 0) + C, nullptr, nullptr, nullptr};
            PROF(PC_GEMM, launch_gemm_bf16(xc, C, (const char*)h->p16_w(4 * l + 0) + (int64_t)C * C * 2, C, h->p16_vec(l, 1) + C, nullptr,
                                           (char*)p.qkv + (int64_t)C * 2, 3 * C, M, 2 * C, C, 3, st, nullptr, &qk, 1));
            PROF(PC_ROWOP, launch_kv_scatter_f16(p.qkv, h->kv, l, h->page_table, h->max_pages, h->ctx_len, lens, B, T, C, st));
            break;
        }
        const BfEpiLn q{p.rowstat, h->p16_vec(l, 0), nullptr, nullptr, nullptr};
        PROF(PC_GEMM, launch_gemm_bf16(xc, C, h->p16_w(4 * l + 0), C, h->p16_vec(l, 1), nullptr, p.qkv, 3 * C, M, 3 * C, C, 3, st, nullptr, &q, 1));
        const KvPages pages{h->kv, l, h->page_table, h->max_pages};
        if (tune(TUNE_DECODER_PREFILL16_PAGES)) {
            PROF(PC_ATTN_DENSE, launch_attn_bf16(p.qkv, lens ? p.mask : nullptr, p.att, B, T, c.n_head, h->dh, st, 1, &pages));
        } else {
            PROF(PC_ROWOP, launch_kv_scatter_f16(p.qkv, h->kv, l, h->page_table, h->max_pages, h->ctx_len, lens, B, T, C, st));
            PROF(PC_ATTN_DENSE, launch_attn_bf16(p.qkv, lens ? p.mask : nullptr, p.att, B, T, c.n_head, h->dh, st, 1));
        }
        const BfEpiLn o{p.ident, nullptr, id_g, id_b, p.stats_part};
        PROF(PC_GEMM, launch_gemm_bf16(p.att, C, h->p16_w(4 * l + 1), C, h->lw(l, L_OUTB), xc, xo, C, M, C, C, 5, st, nullptr, &o, 1));
        PROF(PC_ROWOP, launch_ln_rowstat(p.stats_part, p.rowstat, M, npart, C, c.ln_eps, st));
        const BfEpiLn f1{p.rowstat, h->p16_vec(l, 2), nullptr, nullptr, nullptr};
        PROF(PC_GEMM, launch_gemm_bf16(xo, C, h->p16_w(4 * l + 2), C, h->p16_vec(l, 3), nullptr, p.hid, F, M, F, C, 4, st, nullptr, &f1, 1));
        const BfEpiLn f2{p.ident, nullptr, id_g, id_b, last ? nullptr : p.stats_part};
        GemmBf16Info rv{0, 0, 1};                            // reads FC1's big output: walk the tiles backwards (bf16.hip)
        PROF(PC_GEMM, launch_gemm_bf16(p.hid, F, h->p16_w(4 * l + 3), F, h->lw(l, L_FC2B), xo, xc, C, M, C, F, 5, st, &rv, &f2, 1));
        if (!last) PROF(PC_ROWOP, launch_ln_rowstat(p.stats_part, p.rowstat, M, npart, C, c.ln_eps, st));
    }
    if (logits_out)
        PROF(PC_GEMM, launch_gemm_bf16(xc, C, h->p16_w(4 * c.n_layer), C, h->head_b(), nullptr, logits_out, V, M, V, C, 6, st, nullptr, nullptr, 1));
    h->counters[5] += 1;
    return MGEA_OK;
}

int do_reset(mgea_decoder* h, int B, int max_len, hipStream_t st) {
    const auto& c = h->cfg;
    MGEA_REQUIRE(B > 0 && B <= c.max_batch, MGEA_ECAPACITY, "batch %d exceeds max_batch %d", B, c.max_batch);
    MGEA_REQUIRE(max_len > 0 && max_len <= c.max_ctx, MGEA_ECAPACITY, "context %d exceeds max_ctx %d", max_len, c.max_ctx);
    const int ppr = ceil_div(max_len, MGEA_KV_PAGE_TOKENS);
    MGEA_REQUIRE((int64_t)B * ppr <= h->kv.n_pages, MGEA_ECAPACITY, "KV pool too small: need %d pages", B * ppr);
    // page allocation: logical page j of row b -> physical j*B + b (the rows' j-th pages are
    // neighbours, so a decode step streams one contiguous region per page index)
    std::vector<int32_t> pt((size_t)c.max_batch * h->max_pages, 0);
    for (int b = 0; b < B; ++b)
        for (int j = 0; j < ppr; ++j) pt[(size_t)b * h->max_pages + j] = j * B + b;
    h->kv.arith_batch = B;   // the rule above, for kernels that would rather compute a page id than load it (attn_paged.hip)
    MGEA_CHECK_HIP(hipMemcpyAsync(h->page_table, pt.data(), pt.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    MGEA_CHECK_HIP(hipStreamSynchronize(st));  // pt is a stack-lifetime host buffer
    const size_t nb = (size_t)c.max_batch * sizeof(int32_t);
    MGEA_CHECK_HIP(hipMemsetAsync(h->ctx_len, 0, nb, st));
    MGEA_CHECK_HIP(hipMemsetAsync(h->cur_ids, 0, nb, st));
    MGEA_CHECK_HIP(hipMemsetAsync(h->done, 0, nb, st));
    MGEA_CHECK_HIP(hipMemsetAsync(h->row_step, 0, nb, st));
    MGEA_CHECK_HIP(hipMemsetAsync(h->n_done, 0, 16, st));
    h->cur_batch = B;
    h->reserved_len = ppr * MGEA_KV_PAGE_TOKENS < c.max_ctx ? ppr * MGEA_KV_PAGE_TOKENS : c.max_ctx;
    h->host_max_len = 0;
    h->host_min_len = 0;
    return MGEA_OK;
}

int do_forward(mgea_decoder* h, const int32_t* ids, const int32_t* lens, int B, int T, float* logits_out,
               hipStream_t st) {
    const auto& c = h->cfg;
    MGEA_REQUIRE(ids, MGEA_EINVAL, "forward: ids is NULL");
    MGEA_REQUIRE(h->cur_batch > 0, MGEA_EINVAL, "forward: call mgea_decoder_reset first");
    MGEA_REQUIRE(B == h->cur_batch, MGEA_EINVAL, "forward: batch %d differs from the reset batch %d", B, h->cur_batch);
    MGEA_REQUIRE(T > 0, MGEA_EINVAL, "forward: T must be positive");
    // the reference fails with a broadcast RuntimeError when T exceeds the position table (api_cache.py:99)
    MGEA_REQUIRE(T <= c.seq_len, MGEA_EINVAL, "T=%d exceeds the position table (%d rows)", T, c.seq_len);
    const bool post = c.block_mode == MGEA_BLOCK_POSTLN_RELU;
    if (!post)
        MGEA_REQUIRE(h->host_max_len + T <= h->reserved_len, MGEA_ECAPACITY,
                     "context %d + %d new tokens exceeds the reserved %d", h->host_max_len, T, h->reserved_len);
    const int C = c.d_model, V = c.vocab;
    const int64_t M = (int64_t)B * T;
    MGEA_REQUIRE(M < (1ll << 30), MGEA_EINVAL, "forward: too many tokens");
    MGEA_TRY(ensure_ws(h, M));
    const bool cache_attn = (!post) && h->host_max_len > 0;
    const bool p16 = !fused_ok(h, (int)M) && prefill16_ok(h, M, cache_attn, logits_out);
    const bool kv_only_last = !logits_out && !post && !tune(TUNE_DECODER_PREFILL_FULL);   // logits dropped: the last block stops at its K | V
    if (p16) {
        MGEA_TRY(run_prefill16(h, ids, lens, B, T, logits_out, st, kv_only_last));   // computes the logits itself (fp32 output of its head GEMM)
    } else if (fused_ok(h, (int)M)) {
        MGEA_TRY(launch_embed_stats(ids, lens, h->ctx_len, h->w(T_TOK), h->w(T_POS), h->x, h->stats, B, T, C, V,
                                    c.seq_len, c.pos_mode == MGEA_POS_ABSOLUTE, h->err_flag, st));
        MGEA_TRY(run_blocks_fused(h, main_bufs(h), B, T, lens, cache_attn, st, kv_only_last));
    } else {
        MGEA_TRY(launch_embed_ln(ids, lens, h->ctx_len, h->w(T_TOK), h->w(T_POS), h->x, h->xn,
                                 post ? nullptr : h->lw(0, L_LN1W), post ? nullptr : h->lw(0, L_LN1B), c.ln_eps, B, T, C,
                                 V, c.seq_len, (!post) && c.pos_mode == MGEA_POS_ABSOLUTE, h->err_flag, st));
        MGEA_TRY(run_blocks(h, B, T, lens, cache_attn, !post, st, kv_only_last));
    }
    if (p16) {
    } else if (logits_out && fused_ok(h, (int)M)) {
        SkinnyArgs a{};  // x is k-tiled on the fused path: the head is the skinny LOGITS kernel
        a.w_f16 = h->f16;
        a.M = (int)M; a.A = h->x; a.lda = C; a.W = h->head_tw(); a.bias = h->head_b(); a.N = V; a.K = C;
        a.out = logits_out; a.ldo = V;
        MGEA_TRY(launch_skinny(EPI_LOGITS, a, st));
    } else if (logits_out) {
        for (int64_t r0 = 0; r0 < M; r0 += 4096) {
            const int rows = (int)((M - r0) < 4096 ? (M - r0) : 4096);
            if (V % 4 == 0 && gemm_direct_epilogue_ok(rows, V)) {   // bias inside the GEMM: no slab write + read of rows x V floats.  (Same sums as the slab form wherever that
                                                                    // would not split K: from 256 tiles on.  A remainder chunk of 128..255 tiles used to run with K split; its
                                                                    // logits differ from that form in the last bits of the fp32 sums, within the 5e-6 the long-prompt test observes.)
                PROF(PC_GEMM, launch_gemm_f32_bias_act(h->x + r0 * C, C, h->head_w(), C, h->head_b(), logits_out + r0 * V, V, rows, V, C, ACT_NONE, st));
                continue;
            }
            int S = 1;
            MGEA_TRY(gemm(h, h->x + r0 * C, C, h->head_w(), rows, V, C, &S, st));
            MGEA_TRY(launch_bias_act(h->slabs, S, slab_floats(rows, V), (int)slab_ld(V), h->head_b(),
                                     logits_out + r0 * V, V, rows, V, ACT_NONE, st));
        }
    }
    if (!post) {
        MGEA_TRY(launch_add_lens(h->ctx_len, lens, T, B, st));
        h->host_max_len += T;
    }
    MGEA_TRY(launch_take_last(ids, lens, h->cur_ids, B, T, st));
    return MGEA_OK;
}

}  // namespace

extern "C" {

int mgea_decoder_arena_layout(const mgea_decoder_config* cfg, int64_t* offsets_floats, int32_t* n_tensors,
                              int64_t* total_floats) {
    MGEA_TRY(validate(cfg));
    std::vector<int64_t> offs;
    int64_t total = 0;
    const int n = arena_layout(*cfg, &offs, &total);
    if (offsets_floats)
        for (int i = 0; i < n; ++i) offsets_floats[i] = offs[i];
    if (n_tensors) *n_tensors = n;
    if (total_floats) *total_floats = total;
    return MGEA_OK;
}

// (Re)derive the fragment-ordered matrices of the fused decode path from the arena; synchronous.
static int build_tiled_weights(mgea_decoder* h, hipStream_t st) {
    const auto& c = h->cfg;
    const int C = c.d_model, F = c.d_ff, V = c.vocab;
    if (!h->wt) {
        int64_t total = 0;
        h->wt_off.clear();
        for (int l = 0; l < c.n_layer; ++l) {
            h->wt_off.push_back(total); total += wtile_floats(3 * C, C);
            h->wt_off.push_back(total); total += wtile_floats(C, C);
            h->wt_off.push_back(total); total += wtile_floats(F, C);
            h->wt_off.push_back(total); total += wtile_floats(C, F);
        }
        h->wt_off.push_back(total); total += wtile_floats(V, C);
        if (hipMalloc((void**)&h->wt, (size_t)total * (h->f16 ? 2 : 4)) != hipSuccess ||
            hipMalloc((void**)&h->lnv, (size_t)c.n_layer * (6 * C + 2 * F) * sizeof(float)) != hipSuccess) {
            if (h->wt) (void)hipFree(h->wt);
            h->wt = nullptr;
            set_error("decoder_create: allocation of the decode-layout weights (%lld MB) failed", (long long)(total * (h->f16 ? 2 : 4) >> 20));
            return MGEA_ENOMEM;
        }
    }
    if (h->f16) {
        // private model copy: the caller's arena with the five matrix kinds rounded to fp16 (fp32 storage) ...
        float* own = h->arena_own;
        MGEA_CHECK_HIP(hipMemcpyAsync(own, h->arena_src, (size_t)h->arena_total * sizeof(float), hipMemcpyDeviceToDevice, st));
        const int mats[4] = {L_INW, L_OUTW, L_FC1W, L_FC2W};
        const int64_t msz[4] = {3ll * C * C, (int64_t)C * C, (int64_t)F * C, (int64_t)C * F};
        for (int l = 0; l < c.n_layer; ++l)
            for (int j = 0; j < 4; ++j) MGEA_TRY(launch_round_f16_inplace(own + h->off[2 + l * L_COUNT + mats[j]], msz[j], st));
        MGEA_TRY(launch_round_f16_inplace(own + h->off[2 + c.n_layer * L_COUNT], (int64_t)V * C, st));
        // ... and its fp16 fragments for the decode path; LayerNorm's gamma is applied on the activation side (gemm_skinny.hip)
        _Float16* wt = static_cast<_Float16*>(h->wt);
        for (int l = 0; l < c.n_layer; ++l) {
            MGEA_TRY(launch_tile_weights_f16(h->lw(l, L_INW), 3 * C, C, wt + h->wt_off[4 * l + 0], st));
            MGEA_TRY(launch_ln_vectors(h->lw(l, L_INW), h->lw(l, L_LN1W), h->lw(l, L_LN1B), h->lw(l, L_INB), 3 * C, C,
                                       const_cast<float*>(h->qkv_c1(l)), const_cast<float*>(h->qkv_c2(l)), st));
            MGEA_TRY(launch_tile_weights_f16(h->lw(l, L_OUTW), C, C, wt + h->wt_off[4 * l + 1], st));
            MGEA_TRY(launch_tile_weights_f16(h->lw(l, L_FC1W), F, C, wt + h->wt_off[4 * l + 2], st));
            MGEA_TRY(launch_ln_vectors(h->lw(l, L_FC1W), h->lw(l, L_LN2W), h->lw(l, L_LN2B), h->lw(l, L_FC1B), F, C,
                                       const_cast<float*>(h->fc1_c1(l)), const_cast<float*>(h->fc1_c2(l)), st));
            MGEA_TRY(launch_tile_weights_f16(h->lw(l, L_FC2W), C, F, wt + h->wt_off[4 * l + 3], st));
        }
        MGEA_TRY(launch_tile_weights_f16(h->head_w(), V, C, wt + h->wt_off[4 * c.n_layer], st));
        MGEA_CHECK_HIP(hipStreamSynchronize(st));
        return MGEA_OK;
    }
    float* wt = static_cast<float*>(h->wt);
    for (int l = 0; l < c.n_layer; ++l) {
        MGEA_TRY(launch_ln_fold(h->lw(l, L_INW), h->lw(l, L_LN1W), h->lw(l, L_LN1B), h->lw(l, L_INB), 3 * C, C,
                                wt + h->wt_off[4 * l + 0], const_cast<float*>(h->qkv_c1(l)), const_cast<float*>(h->qkv_c2(l)), st));
        MGEA_TRY(launch_tile_weights(h->lw(l, L_OUTW), C, C, wt + h->wt_off[4 * l + 1], st));
        MGEA_TRY(launch_ln_fold(h->lw(l, L_FC1W), h->lw(l, L_LN2W), h->lw(l, L_LN2B), h->lw(l, L_FC1B), F, C,
                                wt + h->wt_off[4 * l + 2], const_cast<float*>(h->fc1_c1(l)), const_cast<float*>(h->fc1_c2(l)), st));
        MGEA_TRY(launch_tile_weights(h->lw(l, L_FC2W), C, F, wt + h->wt_off[4 * l + 3], st));
    }
    MGEA_TRY(launch_tile_weights(h->head_w(), V, C, wt + h->wt_off[4 * c.n_layer], st));
    MGEA_CHECK_HIP(hipStreamSynchronize(st));
    return MGEA_OK;
}

int mgea_decoder_create(const mgea_decoder_config* cfg, const float* arena_dev, mgea_decoder** out) {
    MGEA_TRY(validate(cfg));
    MGEA_REQUIRE(arena_dev && out, MGEA_EINVAL, "decoder_create: NULL argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device visible: the MI355X path has no CPU fallback");
        return MGEA_ENODEVICE;
    }
    mgea_decoder* h = new mgea_decoder();
    h->cfg = *cfg;
    h->arena = h->arena_src = arena_dev;
    h->f16 = cfg->dtype == MGEA_DTYPE_F16;
    int64_t total = 0;
    arena_layout(*cfg, &h->off, &total);
    h->arena_total = total;
    h->dh = cfg->d_model / cfg->n_head;
    h->force_unfused = tune(TUNE_DECODER_UNFUSED) == 1;   // A/B switches (tools/README.md), latched per engine
    h->no_gemv = tune(TUNE_DECODER_NOGEMV) == 1;
    h->no_graph = tune(TUNE_DECODER_NOGRAPH) == 1;
    { DeviceInfo di; MGEA_TRY(device_info(&di)); }   // cached now: launchers ask for it inside graph capture
    h->pages_per_row_cap = ceil_div(cfg->max_ctx, MGEA_KV_PAGE_TOKENS);
    h->max_pages = h->pages_per_row_cap;
    h->kv.n_pages = cfg->max_batch * h->pages_per_row_cap;
    h->kv.H = cfg->n_head;
    h->kv.dh = h->dh;
    h->kv.f16 = h->f16 ? 1 : 0;
    h->kv.layer_stride = (int64_t)h->kv.n_pages * 2 * cfg->n_head * h->kv.page_elems();
    const size_t pool_bytes = (size_t)h->kv.layer_stride * cfg->n_layer * h->kv.elt_bytes();
    auto fail = [&](int code, const char* what) {
        set_error("decoder_create: %s", what);
        mgea_decoder_destroy(h);
        return code;
    };
    if (h->f16) {
        if (hipMalloc((void**)&h->arena_own, (size_t)total * sizeof(float)) != hipSuccess)
            return fail(MGEA_ENOMEM, "allocation of the fp16-rounded model copy failed");
        h->arena = h->arena_own;   // filled by build_tiled_weights below
    }
    if (cfg->block_mode == MGEA_BLOCK_PRELN_GELU) {
        if (hipMalloc((void**)&h->kv.base, pool_bytes) != hipSuccess) return fail(MGEA_ENOMEM, "KV pool allocation failed");
        if (hipMemset(h->kv.base, 0, pool_bytes) != hipSuccess) return fail(MGEA_EHIP, "KV pool memset failed");
    }
    const size_t nb = (size_t)cfg->max_batch * sizeof(int32_t);
    h->ids_hist_stride = cfg->max_ctx;
    if (hipMalloc((void**)&h->page_table, nb * h->max_pages) != hipSuccess || hipMalloc((void**)&h->ctx_len, nb) != hipSuccess ||
        hipMalloc((void**)&h->cur_ids, nb) != hipSuccess || hipMalloc((void**)&h->done, nb) != hipSuccess ||
        hipMalloc((void**)&h->row_step, nb) != hipSuccess || hipMalloc((void**)&h->sampled, nb) != hipSuccess ||
        hipMalloc((void**)&h->n_done, 16) != hipSuccess || hipMalloc((void**)&h->samp_dev, sizeof(SamplerParams)) != hipSuccess ||
        hipMalloc((void**)&h->err_flag, 16) != hipSuccess ||
        hipMalloc((void**)&h->ids_hist, nb * h->ids_hist_stride) != hipSuccess)
        return fail(MGEA_ENOMEM, "state allocation failed");
    h->attn_split.max_split = MGEA_ATTN_MAX_SPLIT;
    h->attn_split.max_items = MGEA_ATTN_SPLIT_ITEMS;   // attn_split_count(): (row, head) pairs, one query each
    if (hipMalloc((void**)&h->attn_split.part, (size_t)MGEA_ATTN_SPLIT_ITEMS * MGEA_ATTN_MAX_SPLIT * attn_part_floats(h->dh) * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&h->attn_split.count, MGEA_ATTN_SPLIT_ITEMS * sizeof(int32_t)) != hipSuccess)
        return fail(MGEA_ENOMEM, "state allocation failed");
    (void)hipMemset(h->attn_split.count, 0, MGEA_ATTN_SPLIT_ITEMS * sizeof(int32_t));
    (void)hipMemset(h->page_table, 0, nb * h->max_pages);
    (void)hipMemset(h->ctx_len, 0, nb);
    (void)hipMemset(h->done, 0, nb);
    (void)hipMemset(h->row_step, 0, nb);
    (void)hipMemset(h->cur_ids, 0, nb);
    (void)hipMemset(h->n_done, 0, 16);
    (void)hipMemset(h->samp_dev, 0, sizeof(SamplerParams));
    (void)hipMemset(h->err_flag, 0, 16);
    const int rc = ensure_ws(h, cfg->max_batch > 64 ? cfg->max_batch : 64);
    if (rc != MGEA_OK) {
        mgea_decoder_destroy(h);
        return rc;
    }
    if (fused_geometry(*cfg)) {
        const int rc2 = build_tiled_weights(h, nullptr);
        if (rc2 != MGEA_OK) {
            mgea_decoder_destroy(h);
            return rc2;
        }
    }
    *out = h;
    return MGEA_OK;
}

int mgea_decoder_refresh_weights(mgea_decoder* h, void* stream) {
    MGEA_REQUIRE(h, MGEA_EINVAL, "decoder handle is NULL");
    std::lock_guard<std::mutex> lk(h->mu);
    h->p16.weights_ready = false;   // rebuilt from the refreshed model copy at the next big-batch prefill
    if (!fused_geometry(h->cfg)) return MGEA_OK;
    return build_tiled_weights(h, (hipStream_t)stream);
}

int mgea_decoder_destroy(mgea_decoder* h) {
    if (!h) return MGEA_OK;
    (void)hipDeviceSynchronize();
    drop_graphs(h);
    free_ws(h);
    free_p16(h);
    void* p[] = {h->kv.base, h->page_table, h->ctx_len, h->cur_ids, h->done, h->row_step, h->n_done, h->sampled, h->ids_hist, h->wt, h->lnv,
                 h->samp_dev, h->err_flag, h->arena_own, h->attn_split.part, h->attn_split.count};
    for (void* q : p)
        if (q) (void)hipFree(q);
    delete h;
    return MGEA_OK;
}

int mgea_decoder_reset(mgea_decoder* h, int32_t batch, int32_t max_len, void* stream) {
    MGEA_REQUIRE(h, MGEA_EINVAL, "decoder handle is NULL");
    std::lock_guard<std::mutex> lk(h->mu);
    return do_reset(h, batch, max_len, (hipStream_t)stream);
}

int mgea_decoder_forward(mgea_decoder* h, const int32_t* ids_dev, const int32_t* lens_dev, int32_t B, int32_t T,
                         float* logits_out_dev, void* stream) {
    MGEA_REQUIRE(h, MGEA_EINVAL, "decoder handle is NULL");
    std::lock_guard<std::mutex> lk(h->mu);
    return do_forward(h, ids_dev, lens_dev, B, T, logits_out_dev, (hipStream_t)stream);
}

int mgea_decoder_step(mgea_decoder* h, const int32_t* ids_in_dev, const mgea_sampler_config* s, int32_t* ids_out_dev,
                      float* logits_out_dev, void* stream) {
    MGEA_REQUIRE(h && s, MGEA_EINVAL, "decoder_step: NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t st = (hipStream_t)stream;
    MGEA_REQUIRE(h->cfg.block_mode == MGEA_BLOCK_PRELN_GELU, MGEA_EINVAL, "decoder_step needs the KV-cache block mode");
    MGEA_REQUIRE(h->cur_batch > 0, MGEA_EINVAL, "decoder_step: call reset/forward first");
    MGEA_REQUIRE(h->host_max_len + 1 <= h->reserved_len, MGEA_ECAPACITY, "context %d + 1 exceeds the reserved %d",
                 h->host_max_len, h->reserved_len);
    const int B = h->cur_batch;
    if (ids_in_dev)
        MGEA_CHECK_HIP(hipMemcpyAsync(h->cur_ids, ids_in_dev, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    MGEA_TRY(enqueue_step(h, B, *s, nullptr, logits_out_dev, st));   // eager single step: the scalars travel by value
    h->host_max_len += 1;
    if (ids_out_dev)
        MGEA_CHECK_HIP(hipMemcpyAsync(ids_out_dev, h->sampled, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    return MGEA_OK;
}

int mgea_decoder_generate(mgea_decoder* h, const int32_t* prompt_ids_dev, const int32_t* lens_dev, int32_t B,
                          int32_t Tp, int32_t n_steps, const mgea_sampler_config* s, int32_t* ids_out_dev,
                          void* stream) {
    MGEA_REQUIRE(h && s && prompt_ids_dev && ids_out_dev, MGEA_EINVAL, "decoder_generate: NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t st = (hipStream_t)stream;
    const auto& c = h->cfg;
    MGEA_REQUIRE(c.block_mode == MGEA_BLOCK_PRELN_GELU, MGEA_EINVAL, "decoder_generate needs the KV-cache block mode");
    MGEA_REQUIRE(n_steps >= 0 && Tp > 0, MGEA_EINVAL, "decoder_generate: bad n_steps / Tp");
    MGEA_REQUIRE(Tp + n_steps <= c.max_ctx, MGEA_ECAPACITY, "prompt %d + %d steps exceeds max_ctx %d", Tp, n_steps, c.max_ctx);
    MGEA_REQUIRE(s->temperature > 0.f, MGEA_EINVAL, "temperature must be > 0");
    MGEA_TRY(do_reset(h, B, Tp + n_steps, st));
    MGEA_TRY(do_forward(h, prompt_ids_dev, lens_dev, B, Tp, nullptr, st));  // prefill, logits dropped (api_cache.py:163)
    if (n_steps == 0) return MGEA_OK;

    // the request's sampler scalars -> device memory (stream-ordered), then the cached step graph of this batch size:
    // all per-step state lives in device memory, so one graph serves every step of every request
    MGEA_TRY(launch_set_sampler_params(h->samp_dev, *s, st));
    hipGraphExec_t gexec = nullptr, gexec_k = nullptr;
    if (!h->no_graph) MGEA_TRY(step_graph(h, B, *s, st, &gexec));
    // several steps per graph launch (switch decoder_graph_steps, a divisor of 16 so that the EOS poll below keeps its rhythm)
    int K = h->no_graph || h->prof_stride > 0 ? 1 : tune(TUNE_DECODER_GRAPH_STEPS);
    if (K != 2 && K != 4 && K != 8 && K != 16) K = 1;
    if (K > 1 && n_steps >= K) MGEA_TRY(step_graph(h, B, *s, st, &gexec_k, K));
    MGEA_TRY(prime_gen(h, B, st));   // x <- embedding of the re-fed last prompt token (api_cache.py:167)
    int launched = 0;
    int32_t host_done = 0;
    while (launched < n_steps) {
        const int i = launched;
        if (h->prof_stride > 0 && (i % h->prof_stride) == h->prof_stride / 2) {
            h->prof_now = true;  // this step runs eagerly with HIP events around every launch
            const int rc = enqueue_gen_step(h, B, *s, st);
            h->prof_now = false;
            MGEA_TRY(rc);
            ++launched;
        } else if (h->no_graph) {
            MGEA_TRY(enqueue_gen_step(h, B, *s, st));
            ++launched;
        } else if (gexec_k && i % K == 0 && i + K <= n_steps) {
            MGEA_CHECK_HIP(hipGraphLaunch(gexec_k, st));
            launched += K;
        } else {
            MGEA_CHECK_HIP(hipGraphLaunch(gexec, st));
            ++launched;
        }
        if (s->eos_id >= 0 && (launched % 16) == 0) {  // stop once every row has drawn EOS (api_cache.py:181)
            MGEA_CHECK_HIP(hipMemcpyAsync(&host_done, h->n_done, sizeof(int32_t), hipMemcpyDeviceToHost, st));
            MGEA_CHECK_HIP(hipStreamSynchronize(st));
            if (host_done >= B) break;
        }
    }
    h->host_max_len += launched;
    h->counters[1] = launched;
    // rows: ids_hist[b, 0:launched]; steps never run are -1
    MGEA_CHECK_HIP(hipMemsetAsync(ids_out_dev, 0xff, (size_t)B * n_steps * sizeof(int32_t), st));
    MGEA_CHECK_HIP(hipMemcpy2DAsync(ids_out_dev, (size_t)n_steps * sizeof(int32_t), h->ids_hist,
                                    (size_t)h->ids_hist_stride * sizeof(int32_t), (size_t)launched * sizeof(int32_t), B,
                                    hipMemcpyDeviceToDevice, st));
    return MGEA_OK;
}

int mgea_decoder_context_lengths(mgea_decoder* h, int32_t* lens_out_dev, void* stream) {
    MGEA_REQUIRE(h && lens_out_dev, MGEA_EINVAL, "context_lengths: NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    MGEA_CHECK_HIP(hipMemcpyAsync(lens_out_dev, h->ctx_len, (size_t)(h->cur_batch > 0 ? h->cur_batch : 0) * sizeof(int32_t),
                                  hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return MGEA_OK;
}

int mgea_decoder_profile(mgea_decoder* h, int32_t stride) {
    MGEA_REQUIRE(h && stride >= 0, MGEA_EINVAL, "decoder_profile: bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    h->prof_stride = stride;
    return MGEA_OK;
}

int mgea_decoder_profile_read(mgea_decoder* h, double* ms_by_class, int64_t* launches_by_class, int32_t n_classes) {
    MGEA_REQUIRE(h && ms_by_class && launches_by_class && n_classes >= PC_COUNT, MGEA_EINVAL, "decoder_profile_read: bad argument");
    std::lock_guard<std::mutex> lk(h->mu);
    MGEA_CHECK_HIP(hipDeviceSynchronize());
    for (int i = 0; i < n_classes; ++i) { ms_by_class[i] = 0.0; launches_by_class[i] = 0; }
    for (auto& r : h->prof) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            ms_by_class[r.cls] += ms;
            launches_by_class[r.cls] += 1;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    h->prof.clear();
    return MGEA_OK;
}

int mgea_decoder_error_flags(mgea_decoder* h, int32_t* flags_out, void* stream) {
    MGEA_REQUIRE(h && flags_out, MGEA_EINVAL, "decoder_error_flags: NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t st = (hipStream_t)stream;
    int32_t v = 0;
    MGEA_CHECK_HIP(hipMemcpyAsync(&v, h->err_flag, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    MGEA_CHECK_HIP(hipStreamSynchronize(st));
    if (v) MGEA_CHECK_HIP(hipMemsetAsync(h->err_flag, 0, sizeof(int32_t), st));
    *flags_out = v;
    return MGEA_OK;
}

int mgea_decoder_stats(mgea_decoder* h, int64_t* out) {
    MGEA_REQUIRE(h && out, MGEA_EINVAL, "decoder_stats: NULL argument");
    for (int i = 0; i < 8; ++i) out[i] = h->counters[i];
    return MGEA_OK;
}

}  // extern "C"
