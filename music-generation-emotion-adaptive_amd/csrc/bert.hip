// DistilBERT(+LoRA, pre-merged) sequence-classifier forward behind the C ABI.  Replaces the
// `model(**inputs).logits` call of emotion_analysis/inference.py:16-20 (third-party
// transformers DistilBertForSequenceClassification loaded by emotion_analysis/modeling.py:14-21):
//   LN(word[ids] + pos[:S]) -> 6 x { qkv = h Wqkv^T + b (q_lin|k_lin|v_lin stacked: one GEMM);
//   softmax(q k^T / sqrt(dh) + key mask) v; h = LN(out_lin(ctx) + h);
//   h = LN(lin2(gelu_erf(lin1(h))) + h) } -> h[:,0] -> pre_classifier -> ReLU -> classifier.
// Prefill-only: every contraction is an MFMA GEMM (gemm_f32.hip) or the MFMA flash attention
// (attn_dense.hip); LayerNorm / bias / GELU / residual live in the slab epilogues (rowops.hip).
#include <mutex>
#include <vector>

#include "common.h"

using namespace mgea;

namespace {
enum { B_WORD = 0, B_POS, B_ELNW, B_ELNB, B_HEAD0,
       BL_QKVW = 0, BL_QKVB, BL_OUTW, BL_OUTB, BL_SALNW, BL_SALNB, BL_L1W, BL_L1B, BL_L2W, BL_L2B, BL_OLNW, BL_OLNB,
       BL_COUNT };

int bert_layout(const mgea_bert_config& c, std::vector<int64_t>* offs, int64_t* total) {
    const int64_t V = c.vocab, D = c.dim, Hd = c.hidden, P = c.max_pos, NL = c.num_labels;
    std::vector<int64_t> sizes = {V * D, P * D, D, D};
    for (int i = 0; i < c.n_layers; ++i) {
        const int64_t s[BL_COUNT] = {3 * D * D, 3 * D, D * D, D, D, D, Hd * D, Hd, D * Hd, D, D, D};
        for (int j = 0; j < BL_COUNT; ++j) sizes.push_back(s[j]);
    }
    sizes.push_back(D * D);
    sizes.push_back(D);
    sizes.push_back(NL * D);
    sizes.push_back(NL);
    int64_t o = 0;
    if (offs) offs->clear();
    for (int64_t s : sizes) {
        if (offs) offs->push_back(o);
        o += round_up(s, 64);
    }
    *total = o;
    return (int)sizes.size();
}

constexpr int64_t BF16_MIN_TOKENS = 512;   // bf16 engines: calls with fewer tokens run on the exact-fp32 kernels

int validate(const mgea_bert_config* c) {
    MGEA_REQUIRE(c, MGEA_EINVAL, "bert config is NULL");
    MGEA_REQUIRE(c->vocab > 0 && c->max_pos > 0 && c->dim > 0 && c->n_heads > 0 && c->n_layers > 0 && c->hidden > 0 &&
                     c->num_labels > 0 && c->max_tokens > 0,
                 MGEA_EINVAL, "bert config: non-positive dimension");
    MGEA_REQUIRE(c->dim % c->n_heads == 0, MGEA_EINVAL, "dim %d not divisible by n_heads %d", c->dim, c->n_heads);
    const int dh = c->dim / c->n_heads;
    MGEA_REQUIRE(dh == 32 || dh == 64, MGEA_EINVAL, "head_dim %d not supported (32 or 64)", dh);
    MGEA_REQUIRE(c->dim % 32 == 0 && c->hidden % 32 == 0 && c->dim <= 4096, MGEA_EINVAL, "dim/hidden must be multiples of 32, dim <= 4096");
    MGEA_REQUIRE(c->dtype == MGEA_DTYPE_F32 || c->dtype == MGEA_DTYPE_BF16, MGEA_EINVAL, "bert dtype %d unknown", c->dtype);
    if (c->dtype == MGEA_DTYPE_BF16)
        MGEA_REQUIRE(dh == 64 && c->dim % 64 == 0 && c->hidden % 64 == 0 && c->dim <= 2048, MGEA_EINVAL,
                     "bf16 mode needs head_dim 64 and dim/hidden multiples of 64 (dim <= 2048)");
    return MGEA_OK;
}
}  // namespace

struct mgea_bert {
    mgea_bert_config cfg{};
    const float* arena = nullptr;
    std::vector<int64_t> off;
    std::mutex mu;
    float *h = nullptr, *qkv = nullptr, *ctx = nullptr, *ffn = nullptr, *slabs = nullptr, *pooled = nullptr,
          *pooled2 = nullptr;
    void *wb = nullptr, *hb = nullptr, *qkvb = nullptr, *ctxb = nullptr, *ffnb = nullptr, *tmpb = nullptr;  // bf16 mode
    // what the last forward ran (mgea_bert_stats): forwards so far; folded-LayerNorm pipeline or not; bf16 GEMM launches by kernel
    // (persistent / ring / small), persistent launches that cut their tail tiles in halves, by epilogue (0..5); LayerNorm kernels
    int64_t n_forwards = 0, last_fold = 0, last_persistent = 0, last_ring = 0, last_small = 0, last_half_tiles = 0, last_ln_kernels = 0, last_cls_only = 0;
    int64_t last_epi[6] = {0, 0, 0, 0, 0, 0};
    int64_t last_rows = 0;         // rows the last forward ran its GEMMs on (padded: B S; packed: the real tokens)
    int32_t* err_flag = nullptr;   // sticky device flags (bit 0: a token id outside the vocabulary was clamped), as the decoder's
    // bf16 mode, folded-LayerNorm pipeline (big batches: every GEMM on the persistent kernel): W diag(gamma) copies, c1 / c2 vectors,
    // per-tile row sums and the two (mean, rstd) tables
    void* wfold = nullptr;
    float *fvec = nullptr, *stats_part = nullptr, *rowstat_sa = nullptr, *rowstat_out = nullptr;
    float* ident = nullptr;                                  // [M][2] (0, 1) rows, then D ones, then D zeros: the identity LayerNorm of layer 0's residual
    const char* fc1f(int l) const { return (const char*)wfold + (int64_t)l * cfg.hidden * cfg.dim * 2; }
    const char* qkvf(int l) const { return (const char*)wfold + ((int64_t)cfg.n_layers * cfg.hidden + (int64_t)(l - 1) * 3 * cfg.dim) * cfg.dim * 2; }
    float* fc1c(int l, int which) const { return fvec + ((int64_t)l * 2 + which) * cfg.hidden; }
    float* qkvc(int l, int which) const { return fvec + (int64_t)cfg.n_layers * 2 * cfg.hidden + ((int64_t)(l - 1) * 2 + which) * 3 * cfg.dim; }
    const char* wbf(int64_t off_floats) const { return (const char*)wb + off_floats * 2; }
    int64_t slab_cap = 0;
    const float* w(int i) const { return arena + off[i]; }
    const float* lw(int l, int j) const { return arena + off[B_HEAD0 + l * BL_COUNT + j]; }
    const float* hw(int j) const { return arena + off[B_HEAD0 + cfg.n_layers * BL_COUNT + j]; }
};

extern "C" {

int mgea_bert_arena_layout(const mgea_bert_config* cfg, int64_t* offsets_floats, int32_t* n_tensors,
                           int64_t* total_floats) {
    MGEA_TRY(validate(cfg));
    std::vector<int64_t> offs;
    int64_t total = 0;
    const int n = bert_layout(*cfg, &offs, &total);
    if (offsets_floats)
        for (int i = 0; i < n; ++i) offsets_floats[i] = offs[i];
    if (n_tensors) *n_tensors = n;
    if (total_floats) *total_floats = total;
    return MGEA_OK;
}

int mgea_bert_destroy(mgea_bert* h) {
    if (!h) return MGEA_OK;
    (void)hipDeviceSynchronize();
    void* p[] = {h->h, h->qkv, h->ctx, h->ffn, h->slabs, h->pooled, h->pooled2, h->wb, h->hb, h->qkvb, h->ctxb, h->ffnb, h->tmpb, h->wfold, h->fvec, h->stats_part, h->rowstat_sa, h->rowstat_out, h->ident, h->err_flag};
    for (void* q : p)
        if (q) (void)hipFree(q);
    delete h;
    return MGEA_OK;
}

int mgea_bert_create(const mgea_bert_config* cfg, const float* arena_dev, mgea_bert** out) {
    MGEA_TRY(validate(cfg));
    MGEA_REQUIRE(arena_dev && out, MGEA_EINVAL, "bert_create: NULL argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        set_error("no HIP device visible: the MI355X path has no CPU fallback");
        return MGEA_ENODEVICE;
    }
    mgea_bert* h = new mgea_bert();
    h->cfg = *cfg;
    h->arena = arena_dev;
    int64_t total = 0;
    bert_layout(*cfg, &h->off, &total);
    const int64_t M = cfg->max_tokens, D = cfg->dim, Hd = cfg->hidden;
    const int64_t nmax = (3 * D > Hd ? 3 * D : Hd);
    // slabs: big-M GEMMs never split K; the two head GEMMs (M = batch <= max_tokens) may
    int64_t slab = M * slab_ld((int)nmax);
    const int64_t head = 32 * (int64_t)(M < 64 ? M : 64) * slab_ld((int)D);
    slab = slab > head ? slab : head;
    h->slab_cap = slab;
    bool ok;
    if (cfg->dtype == MGEA_DTYPE_BF16) {
        // perf mode: bf16 copy of the whole arena (same offsets), bf16 activations; the classifier head
        // (M = batch rows) stays on the fp32 kernels
        const int64_t small = 32 * 64 * slab_ld((int)D), big = M * slab_ld((int)D);  // split-K (B <= 64) / one slab
        h->slab_cap = small > big ? small : big;
        const int64_t f32_small = 2 * (M < BF16_MIN_TOKENS ? M : BF16_MIN_TOKENS) * slab_ld((int)nmax);   // the exact-fp32 path of small calls (room for a split of K)
        if (f32_small > h->slab_cap) h->slab_cap = f32_small;
        ok = hipMalloc(&h->wb, total * 2) == hipSuccess && hipMalloc(&h->hb, M * D * 2) == hipSuccess &&
             hipMalloc(&h->qkvb, M * 3 * D * 2) == hipSuccess && hipMalloc(&h->ctxb, M * D * 2) == hipSuccess &&
             hipMalloc(&h->ffnb, M * Hd * 2) == hipSuccess && hipMalloc(&h->tmpb, M * D * 2) == hipSuccess &&
             hipMalloc((void**)&h->slabs, h->slab_cap * 4) == hipSuccess && hipMalloc((void**)&h->pooled, M * D * 4) == hipSuccess &&
             hipMalloc((void**)&h->pooled2, M * D * 4) == hipSuccess;
        // small calls (fewer than BF16_MIN_TOKENS tokens: the endpoint's one text per request) run on the exact-fp32 kernels -- the 16-bit
        // GEMMs need hundreds of rows to fill their tiles -- with fp32 activation buffers of that size
        const int64_t Ms = M < BF16_MIN_TOKENS ? M : BF16_MIN_TOKENS;
        if (ok) ok = hipMalloc((void**)&h->h, Ms * D * 4) == hipSuccess && hipMalloc((void**)&h->qkv, Ms * 3 * D * 4) == hipSuccess &&
                     hipMalloc((void**)&h->ctx, Ms * D * 4) == hipSuccess && hipMalloc((void**)&h->ffn, Ms * Hd * 4) == hipSuccess;
        if (ok) ok = launch_f32_to_bf16(arena_dev, h->wb, total, nullptr) == MGEA_OK && hipDeviceSynchronize() == hipSuccess;
        if (ok) {   // folded-LayerNorm pipeline (see mgea_bert_forward): FC1 of every layer, QKV of layers >= 1
            const int64_t L = cfg->n_layers;
            const int64_t wel = (L * Hd + (L - 1) * 3 * D) * D, vel = L * 2 * Hd + (L - 1) * 2 * 3 * D;
            ok = hipMalloc(&h->wfold, wel * 2) == hipSuccess && hipMalloc((void**)&h->fvec, (vel > 0 ? vel : 1) * 4) == hipSuccess &&
                 hipMalloc((void**)&h->stats_part, M * ((D + 255) / 256) * 2 * 4) == hipSuccess &&
                 hipMalloc((void**)&h->rowstat_sa, M * 2 * 4) == hipSuccess && hipMalloc((void**)&h->rowstat_out, M * 2 * 4) == hipSuccess &&
                 hipMalloc((void**)&h->ident, (M * 2 + 2 * D) * 4) == hipSuccess;
            if (ok) {
                std::vector<float> idv((size_t)(M * 2 + 2 * D), 0.f);
                for (int64_t r = 0; r < M; ++r) idv[(size_t)r * 2 + 1] = 1.f;
                for (int64_t d = 0; d < D; ++d) idv[(size_t)(M * 2 + d)] = 1.f;
                ok = hipMemcpy(h->ident, idv.data(), idv.size() * 4, hipMemcpyHostToDevice) == hipSuccess;
            }
            for (int l = 0; ok && l < L; ++l) {
                ok = launch_fold_ln_weights_bf16(h->lw(l, BL_L1W), h->lw(l, BL_SALNW), h->lw(l, BL_SALNB), h->lw(l, BL_L1B), (void*)h->fc1f(l),
                                                 h->fc1c(l, 0), h->fc1c(l, 1), (int)Hd, (int)D, nullptr) == MGEA_OK;
                if (ok && l >= 1)
                    ok = launch_fold_ln_weights_bf16(h->lw(l, BL_QKVW), h->lw(l - 1, BL_OLNW), h->lw(l - 1, BL_OLNB), h->lw(l, BL_QKVB),
                                                     (void*)h->qkvf(l), h->qkvc(l, 0), h->qkvc(l, 1), (int)(3 * D), (int)D, nullptr) == MGEA_OK;
            }
            if (ok) ok = hipDeviceSynchronize() == hipSuccess;
        }
    } else {
        ok = hipMalloc((void**)&h->h, M * D * 4) == hipSuccess && hipMalloc((void**)&h->qkv, M * 3 * D * 4) == hipSuccess &&
             hipMalloc((void**)&h->ctx, M * D * 4) == hipSuccess && hipMalloc((void**)&h->ffn, M * Hd * 4) == hipSuccess &&
             hipMalloc((void**)&h->slabs, slab * 4) == hipSuccess && hipMalloc((void**)&h->pooled, M * D * 4) == hipSuccess &&
             hipMalloc((void**)&h->pooled2, M * D * 4) == hipSuccess;
    }
    if (ok) ok = hipMalloc((void**)&h->err_flag, 16) == hipSuccess && hipMemset(h->err_flag, 0, 16) == hipSuccess;
    if (!ok) {
        set_error("bert_create: out of device memory");
        mgea_bert_destroy(h);
        return MGEA_ENOMEM;
    }
    *out = h;
    return MGEA_OK;
}

}  // extern "C"

// The forward behind both entry points.  Padded input: ids [B, S] (+ key mask), M = B S rows.  PACKED input (cu != NULL, round 4): the
// real tokens of the B sequences back to back -- ids / pos_ids [M], sequence b = rows cu[b] .. cu[b + 1] - 1, S = the longest sequence --
// so that the row-wise GEMMs, the LayerNorm statistics and the attention run on sum(lengths) rows instead of B x S: with the reference's
// tokenizer padding a batch to its longest prompt (emotion_analysis/inference.py:16, padding=True) 44 % of the [256, 128] bench batch is
// padding whose rows nobody reads.  Same logits: a row's GEMM / LayerNorm results do not depend on which other rows are in the batch, and a
// sequence attends to exactly its own real keys in both forms.
static int bert_forward_impl(mgea_bert* h, const int32_t* ids_dev, const int32_t* mask_dev, int32_t B, int32_t S, int64_t M64,
                             const int32_t* cu, const int32_t* pos_ids, float* logits_out_dev, int32_t* argmax_out_dev, hipStream_t st) {
    const auto& c = h->cfg;
    MGEA_REQUIRE(B > 0 && S > 0, MGEA_EINVAL, "bert_forward: bad shape B=%d S=%d", B, S);
    MGEA_REQUIRE(S <= c.max_pos, MGEA_EINVAL, "sequence length %d exceeds max_position_embeddings %d", S, c.max_pos);
    MGEA_REQUIRE(M64 > 0 && M64 <= c.max_tokens, MGEA_ECAPACITY, "%lld tokens exceed max_tokens %d", (long long)M64, c.max_tokens);
    const int M = (int)M64, D = c.dim, Hd = c.hidden, NL = c.num_labels, dh = D / c.n_heads;
    if (cu) {
        MGEA_REQUIRE(!mask_dev && pos_ids && M >= B, MGEA_EINVAL, "bert_forward_packed: packed rows carry positions, no mask, and at least one token per sequence");
        // the 16-bit attention takes a sequence as ONE query block / key stage (<= 256 tokens); the exact-fp32 kernels take any length
        MGEA_REQUIRE(!(c.dtype == MGEA_DTYPE_BF16 && M >= BF16_MIN_TOKENS) || S <= 256, MGEA_EINVAL,
                     "bert_forward_packed on the 16-bit kernels: sequences of at most 256 tokens (longest here: %d)", S);
    }
    auto gemm = [&](const float* A, int lda, const float* W, int m, int n, int k, int* Sout) -> int {
        const int s = pick_split_k(m, n, k, h->slab_cap);
        MGEA_REQUIRE((int64_t)s * slab_floats(m, n) <= h->slab_cap, MGEA_ECAPACITY, "internal: bert slab workspace too small");
        const int rc = launch_gemm_f32(A, lda, W, k, h->slabs, m, n, k, s, st);
        if (rc < 0) return rc;
        *Sout = rc;
        return MGEA_OK;
    };
    int Sk = 1;
    // LAST LAYER, [CLS] ROWS ONLY.  The classifier reads hidden_state[:, 0] of the last layer and nothing else of it
    // (emotion_analysis/modeling.py:14-21 -> DistilBertForSequenceClassification), so of that layer only the keys and values of every
    // position are needed (the [CLS] query attends to them); its query, attention, out-projection, both LayerNorms and the FFN are
    // computed for the B [CLS] rows, in fp32 on the arena's fp32 matrices in both modes (like the classifier head).  Same function of
    // the inputs as computing every position and discarding all but one -- 1 / n_layers of the forward less about 2 / 3 of a QKV GEMM
    // (switch bert_full_last_layer = 1 computes every position; both are tested against the goldens).  The [B, 2 D + hidden] fp32
    // scratch is the FFN buffer, idle in that layer.
    const bool use16_ = c.dtype == MGEA_DTYPE_BF16 && M >= BF16_MIN_TOKENS;
    const int64_t f32_rows = c.dtype == MGEA_DTYPE_BF16 ? (c.max_tokens < BF16_MIN_TOKENS ? c.max_tokens : BF16_MIN_TOKENS) : c.max_tokens;
    const int64_t cls_cap = use16_ ? (int64_t)c.max_tokens * Hd / 2 : f32_rows * Hd;   // floats in ffnb / ffn
    // (cls_tail's fp32 GEMMs go through the slab workspace: [B, hidden] floats must fit too -- a geometry with hidden > S * dim did not,
    // on a bf16 engine whose slab holds about M * dim floats, and got ECAPACITY where the every-position form ran: ADVICE r3)
    bool cls_last = !tune(TUNE_BERT_FULL_LAST_LAYER) && S >= 4 && (int64_t)B * (2 * D + Hd) <= cls_cap &&
                    slab_floats(B, D > Hd ? D : Hd) <= h->slab_cap;
    const int last = c.n_layers - 1;
    auto cls_tail = [&](int l, const void* kv, int kv_bf16) -> int {   // pooled [B, D] = the layer's input rows at [CLS]; kv = qkv buffer with K | V filled
        float* cls = reinterpret_cast<float*>(use16_ ? h->ffnb : (void*)h->ffn);
        float *cq = cls, *cctx = cls + (int64_t)B * D, *chid = cls + (int64_t)2 * B * D;
        MGEA_TRY(gemm(h->pooled, D, h->lw(l, BL_QKVW), B, D, D, &Sk));                     // q = x W_q^T + b_q (rows 0..D-1 of the stacked matrix)
        MGEA_TRY(launch_bias_act(h->slabs, Sk, slab_floats(B, D), (int)slab_ld(D), h->lw(l, BL_QKVB), cq, D, B, D, ACT_NONE, st));
        MGEA_TRY(launch_attn_cls(cq, kv, kv_bf16, mask_dev, cctx, B, S, c.n_heads, dh, st, cu));
        MGEA_TRY(gemm(cctx, D, h->lw(l, BL_OUTW), B, D, D, &Sk));
        MGEA_TRY(launch_bias_res_ln(h->slabs, Sk, slab_floats(B, D), (int)slab_ld(D), h->lw(l, BL_OUTB), h->pooled, nullptr,
                                    h->lw(l, BL_SALNW), h->lw(l, BL_SALNB), c.ln_eps, B, D, 1, st));
        MGEA_TRY(gemm(h->pooled, D, h->lw(l, BL_L1W), B, Hd, D, &Sk));
        MGEA_TRY(launch_bias_act(h->slabs, Sk, slab_floats(B, Hd), (int)slab_ld(Hd), h->lw(l, BL_L1B), chid, Hd, B, Hd, ACT_GELU, st));
        MGEA_TRY(gemm(chid, Hd, h->lw(l, BL_L2W), B, D, Hd, &Sk));
        MGEA_TRY(launch_bias_res_ln(h->slabs, Sk, slab_floats(B, D), (int)slab_ld(D), h->lw(l, BL_L2B), h->pooled, nullptr,
                                    h->lw(l, BL_OLNW), h->lw(l, BL_OLNB), c.ln_eps, B, D, 1, st));
        return MGEA_OK;
    };
    h->last_cls_only = 0;
    const bool use16 = c.dtype == MGEA_DTYPE_BF16 && M >= BF16_MIN_TOKENS;   // (bf16 engines: small calls take the exact-fp32 kernels below)
    if (c.dtype == MGEA_DTYPE_BF16 && !use16) {
        h->last_fold = h->last_persistent = h->last_ring = h->last_small = h->last_half_tiles = h->last_ln_kernels = 0;
        for (int64_t& e : h->last_epi) e = 0;
    }
    if (use16) {
        // perf mode: bf16 MFMA GEMMs with fused bias / GELU / residual epilogues, bf16 flash attention
        auto wb = [&](int l, int j) { return (const void*)h->wbf(h->off[B_HEAD0 + l * BL_COUNT + j]); };
        h->last_fold = h->last_persistent = h->last_ring = h->last_small = h->last_half_tiles = h->last_ln_kernels = 0;
        for (int64_t& e : h->last_epi) e = 0;
        auto bgemm = [&](const void* A, int lda, const void* W, int ldw, const float* bias, const void* res, void* C, int ldc, int m, int n,
                         int k, int epi, const BfEpiLn* ln = nullptr) -> int {
            GemmBf16Info gi{0, 0, k > n ? 1 : 0};   // the FFN down-projection (K = hidden > N) reads the up-projection's big output: walk it backwards
            MGEA_TRY(launch_gemm_bf16(A, lda, W, ldw, bias, res, C, ldc, m, n, k, epi, st, &gi, ln));
            (gi.kernel == 2 ? h->last_persistent : gi.kernel == 1 ? h->last_ring : h->last_small) += 1;
            h->last_half_tiles += gi.half_tiles;
            h->last_epi[epi] += 1;
            return MGEA_OK;
        };
        if (cu) MGEA_TRY(launch_bert_embed_ln_bf16(ids_dev, h->w(B_WORD), h->w(B_POS), h->w(B_ELNW), h->w(B_ELNB), c.ln_eps, h->hb, M, 1, D, c.vocab, st,
                                                   h->err_flag, pos_ids));
        else MGEA_TRY(launch_bert_embed_ln_bf16(ids_dev, h->w(B_WORD), h->w(B_POS), h->w(B_ELNW), h->w(B_ELNB), c.ln_eps, h->hb, B,
                                                S, D, c.vocab, st, h->err_flag));
        // Folded-LayerNorm pipeline: when every GEMM of a layer runs on the persistent 256 x 256 kernel (big batches), no LayerNorm
        // kernel runs at all.  The residual GEMMs write the RAW sums (x + sublayer(x)) and per-tile row sums; a 1-launch reduction
        // turns those into (mean, rstd) per row; the consumers apply the LayerNorm themselves: the next GEMM as rstd (A W'^T - mean
        // c1) + c2 with W' = W diag(gamma) (epilogues 3 / 4), the next residual GEMM by normalising the residual row on the way in
        // (epilogue 5), the classifier head on its B CLS rows.  Saves a read + write of [M, D] per LayerNorm (12 x 20 us at the bench
        // shape).  hb and tmpb alternate as the raw buffers; layer 0 starts from the materialised embedding LayerNorm.
        const bool fold = !tune(TUNE_BERT_BF16_NOFOLD) && D % 256 == 0 && Hd % 256 == 0 && gemm_bf16_is_persistent(M, 3 * D, D) && gemm_bf16_is_persistent(M, D, D) &&
                          gemm_bf16_is_persistent(M, Hd, D) && gemm_bf16_is_persistent(M, D, Hd);
        h->last_fold = fold ? 1 : 0;
        if (fold && cls_last && !gemm_bf16_is_persistent(M, 2 * D, D)) cls_last = false;   // (the K | V GEMM must stay on the kernel with the folding epilogue)
        h->last_cls_only = cls_last ? 1 : 0;
        const int64_t kv_off = (int64_t)D * D * 2;          // bytes: rows D.. of a stacked bf16 [3 D, D] matrix
        if (fold) {
            const int npart = D / 256;
            const float *id_g = h->ident + (int64_t)c.max_tokens * 2, *id_b = id_g + D;
            for (int l = 0; l < c.n_layers; ++l) {
                const bool first = l == 0;
                if (cls_last && l == last) {                // K | V of every position, the rest of the layer on the [CLS] rows
                    if (first) {
                        MGEA_TRY(bgemm(h->hb, D, (const char*)wb(l, BL_QKVW) + kv_off, D, h->lw(l, BL_QKVB) + D, nullptr, (char*)h->qkvb + D * 2, 3 * D, M,
                                       2 * D, D, 0));
                        MGEA_TRY(launch_gather_cls_bf16(h->hb, h->pooled, B, S, D, st, cu));
                    } else {
                        BfEpiLn q{h->rowstat_out, h->qkvc(l, 0) + D, nullptr, nullptr, nullptr};
                        MGEA_TRY(bgemm(h->hb, D, h->qkvf(l) + kv_off, D, h->qkvc(l, 1) + D, nullptr, (char*)h->qkvb + D * 2, 3 * D, M, 2 * D, D, 3, &q));
                        MGEA_TRY(launch_gather_cls_ln_bf16(h->hb, h->rowstat_out, h->lw(l - 1, BL_OLNW), h->lw(l - 1, BL_OLNB), h->pooled, B, S, D, st, cu));
                    }
                    MGEA_TRY(cls_tail(l, h->qkvb, 1));
                    break;
                }
                if (first) {
                    MGEA_TRY(bgemm(h->hb, D, wb(l, BL_QKVW), D, h->lw(l, BL_QKVB), nullptr, h->qkvb, 3 * D, M, 3 * D, D, 0));
                } else {
                    BfEpiLn q{h->rowstat_out, h->qkvc(l, 0), nullptr, nullptr, nullptr};
                    MGEA_TRY(bgemm(h->hb, D, h->qkvf(l), D, h->qkvc(l, 1), nullptr, h->qkvb, 3 * D, M, 3 * D, D, 3, &q));
                }
                MGEA_TRY(launch_attn_bf16(h->qkvb, mask_dev, h->ctxb, B, S, c.n_heads, dh, st, 0, nullptr, cu));
                // out-proj + LayerNorm_out(l-1)(raw hb) as the residual (layer 0: hb is already normalised) -> raw tmpb + row sums
                BfEpiLn o{first ? h->ident : h->rowstat_out, nullptr, first ? id_g : h->lw(l - 1, BL_OLNW), first ? id_b : h->lw(l - 1, BL_OLNB),
                          h->stats_part};
                MGEA_TRY(bgemm(h->ctxb, D, wb(l, BL_OUTW), D, h->lw(l, BL_OUTB), h->hb, h->tmpb, D, M, D, D, 5, &o));
                MGEA_TRY(launch_ln_rowstat(h->stats_part, h->rowstat_sa, M, npart, D, c.ln_eps, st));
                BfEpiLn f1{h->rowstat_sa, h->fc1c(l, 0), nullptr, nullptr, nullptr};
                MGEA_TRY(bgemm(h->tmpb, D, h->fc1f(l), D, h->fc1c(l, 1), nullptr, h->ffnb, Hd, M, Hd, D, 4, &f1));
                // FC2 + LayerNorm_sa(l)(raw tmpb) as the residual -> raw hb + row sums
                BfEpiLn f2{h->rowstat_sa, nullptr, h->lw(l, BL_SALNW), h->lw(l, BL_SALNB), h->stats_part};
                MGEA_TRY(bgemm(h->ffnb, Hd, wb(l, BL_L2W), Hd, h->lw(l, BL_L2B), h->tmpb, h->hb, D, M, D, Hd, 5, &f2));
                MGEA_TRY(launch_ln_rowstat(h->stats_part, h->rowstat_out, M, npart, D, c.ln_eps, st));
            }
            if (!cls_last)
                MGEA_TRY(launch_gather_cls_ln_bf16(h->hb, h->rowstat_out, h->lw(c.n_layers - 1, BL_OLNW), h->lw(c.n_layers - 1, BL_OLNB), h->pooled, B, S,
                                                   D, st, cu));
        } else {
        for (int l = 0; l < c.n_layers; ++l) {
            if (cls_last && l == last) {
                MGEA_TRY(bgemm(h->hb, D, (const char*)wb(l, BL_QKVW) + kv_off, D, h->lw(l, BL_QKVB) + D, nullptr, (char*)h->qkvb + D * 2, 3 * D, M, 2 * D,
                               D, 0));
                MGEA_TRY(launch_gather_cls_bf16(h->hb, h->pooled, B, S, D, st, cu));
                MGEA_TRY(cls_tail(l, h->qkvb, 1));
                break;
            }
            MGEA_TRY(bgemm(h->hb, D, wb(l, BL_QKVW), D, h->lw(l, BL_QKVB), nullptr, h->qkvb, 3 * D, M, 3 * D, D, 0));
            MGEA_TRY(launch_attn_bf16(h->qkvb, mask_dev, h->ctxb, B, S, c.n_heads, dh, st, 0, nullptr, cu));
            MGEA_TRY(bgemm(h->ctxb, D, wb(l, BL_OUTW), D, h->lw(l, BL_OUTB), h->hb, h->tmpb, D, M, D, D, 2));
            MGEA_TRY(launch_layernorm_bf16(h->tmpb, h->lw(l, BL_SALNW), h->lw(l, BL_SALNB), h->hb, M, D, c.ln_eps, st));
            h->last_ln_kernels += 2;
            MGEA_TRY(bgemm(h->hb, D, wb(l, BL_L1W), D, h->lw(l, BL_L1B), nullptr, h->ffnb, Hd, M, Hd, D, 1));
            MGEA_TRY(bgemm(h->ffnb, Hd, wb(l, BL_L2W), Hd, h->lw(l, BL_L2B), h->hb, h->tmpb, D, M, D, Hd, 2));
            MGEA_TRY(launch_layernorm_bf16(h->tmpb, h->lw(l, BL_OLNW), h->lw(l, BL_OLNB), h->hb, M, D, c.ln_eps, st));
        }
        if (!cls_last) MGEA_TRY(launch_gather_cls_bf16(h->hb, h->pooled, B, S, D, st, cu));
        }
    } else {
    h->last_cls_only = cls_last ? 1 : 0;
    if (cu) MGEA_TRY(launch_bert_embed_ln(ids_dev, h->w(B_WORD), h->w(B_POS), h->w(B_ELNW), h->w(B_ELNB), c.ln_eps, h->h, M, 1, D, c.vocab, st,
                                          h->err_flag, pos_ids));
    else MGEA_TRY(launch_bert_embed_ln(ids_dev, h->w(B_WORD), h->w(B_POS), h->w(B_ELNW), h->w(B_ELNB), c.ln_eps, h->h, B, S,
                                       D, c.vocab, st, h->err_flag));
    for (int l = 0; l < c.n_layers; ++l) {
        if (cls_last && l == last) {               // K | V of every position (columns D.. of the stacked projection), the rest on the [CLS] rows
            const float *wkv = h->lw(l, BL_QKVW) + (int64_t)D * D, *bkv = h->lw(l, BL_QKVB) + D;
            if (gemm_direct_epilogue_ok(M, 2 * D)) {
                MGEA_TRY(launch_gemm_f32_bias_act(h->h, D, wkv, D, bkv, h->qkv + D, 3 * D, M, 2 * D, D, ACT_NONE, st));
            } else {
                MGEA_TRY(gemm(h->h, D, wkv, M, 2 * D, D, &Sk));
                MGEA_TRY(launch_bias_act(h->slabs, Sk, slab_floats(M, 2 * D), (int)slab_ld(2 * D), bkv, h->qkv + D, 3 * D, M, 2 * D, ACT_NONE, st));
            }
            MGEA_TRY(launch_gather_rows(h->h, D, h->pooled, D, B, S, D, st, cu));
            MGEA_TRY(cls_tail(l, h->qkv, 0));
            break;
        }
        if (gemm_direct_epilogue_ok(M, 3 * D)) {   // bias (+ GELU below) inside the GEMM epilogue: no slab round trip
            MGEA_TRY(launch_gemm_f32_bias_act(h->h, D, h->lw(l, BL_QKVW), D, h->lw(l, BL_QKVB), h->qkv, 3 * D, M, 3 * D, D,
                                              ACT_NONE, st));
        } else {
            MGEA_TRY(gemm(h->h, D, h->lw(l, BL_QKVW), M, 3 * D, D, &Sk));
            MGEA_TRY(launch_bias_act(h->slabs, Sk, slab_floats(M, 3 * D), (int)slab_ld(3 * D), h->lw(l, BL_QKVB), h->qkv,
                                     3 * D, M, 3 * D, ACT_NONE, st));
        }
        MGEA_TRY(launch_attn_dense(h->qkv, nullptr, mask_dev, h->ctx, B, S, c.n_heads, dh, 0, st, cu));
        MGEA_TRY(gemm(h->ctx, D, h->lw(l, BL_OUTW), M, D, D, &Sk));
        MGEA_TRY(launch_bias_res_ln(h->slabs, Sk, slab_floats(M, D), (int)slab_ld(D), h->lw(l, BL_OUTB), h->h, nullptr,
                                    h->lw(l, BL_SALNW), h->lw(l, BL_SALNB), c.ln_eps, M, D, 1, st));
        if (gemm_direct_epilogue_ok(M, Hd)) {
            MGEA_TRY(launch_gemm_f32_bias_act(h->h, D, h->lw(l, BL_L1W), D, h->lw(l, BL_L1B), h->ffn, Hd, M, Hd, D, ACT_GELU, st));
        } else {
            MGEA_TRY(gemm(h->h, D, h->lw(l, BL_L1W), M, Hd, D, &Sk));
            MGEA_TRY(launch_bias_act(h->slabs, Sk, slab_floats(M, Hd), (int)slab_ld(Hd), h->lw(l, BL_L1B), h->ffn, Hd, M, Hd,
                                     ACT_GELU, st));
        }
        MGEA_TRY(gemm(h->ffn, Hd, h->lw(l, BL_L2W), M, D, Hd, &Sk));
        MGEA_TRY(launch_bias_res_ln(h->slabs, Sk, slab_floats(M, D), (int)slab_ld(D), h->lw(l, BL_L2B), h->h, nullptr,
                                    h->lw(l, BL_OLNW), h->lw(l, BL_OLNB), c.ln_eps, M, D, 1, st));
    }
    if (!cls_last) MGEA_TRY(launch_gather_rows(h->h, D, h->pooled, D, B, S, D, st, cu));
    }
    // pooled = h[:, 0]  ->  pre_classifier -> ReLU -> classifier (fp32 in both modes)
    MGEA_TRY(gemm(h->pooled, D, h->hw(0), B, D, D, &Sk));
    MGEA_TRY(launch_bias_act(h->slabs, Sk, slab_floats(B, D), (int)slab_ld(D), h->hw(1), h->pooled2, D, B, D, ACT_RELU, st));
    MGEA_TRY(gemm(h->pooled2, D, h->hw(2), B, NL, D, &Sk));
    MGEA_TRY(launch_logits_argmax(h->slabs, Sk, slab_floats(B, NL), (int)slab_ld(NL), h->hw(3), logits_out_dev, B, NL,
                                  argmax_out_dev, st));
    h->n_forwards += 1;
    h->last_rows = M;
    return MGEA_OK;
}

extern "C" {

int mgea_bert_forward(mgea_bert* h, const int32_t* ids_dev, const int32_t* mask_dev, int32_t B, int32_t S,
                      float* logits_out_dev, int32_t* argmax_out_dev, void* stream) {
    MGEA_REQUIRE(h && ids_dev, MGEA_EINVAL, "bert_forward: NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    return bert_forward_impl(h, ids_dev, mask_dev, B, S, (int64_t)B * S, nullptr, nullptr, logits_out_dev, argmax_out_dev, (hipStream_t)stream);
}

int mgea_bert_forward_packed(mgea_bert* h, const int32_t* ids_dev, const int32_t* pos_ids_dev, const int32_t* cu_seqlens_dev, int32_t B,
                             int32_t n_tokens, int32_t max_len, float* logits_out_dev, int32_t* argmax_out_dev, void* stream) {
    MGEA_REQUIRE(h && ids_dev && pos_ids_dev && cu_seqlens_dev, MGEA_EINVAL, "bert_forward_packed: NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    return bert_forward_impl(h, ids_dev, nullptr, B, max_len, n_tokens, cu_seqlens_dev, pos_ids_dev, logits_out_dev, argmax_out_dev,
                             (hipStream_t)stream);
}

int mgea_bert_error_flags(mgea_bert* h, int32_t* flags_out, void* stream) {
    MGEA_REQUIRE(h && flags_out, MGEA_EINVAL, "bert_error_flags: NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    hipStream_t st = (hipStream_t)stream;
    int32_t v = 0;
    MGEA_CHECK_HIP(hipMemcpyAsync(&v, h->err_flag, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    MGEA_CHECK_HIP(hipStreamSynchronize(st));
    if (v) MGEA_CHECK_HIP(hipMemsetAsync(h->err_flag, 0, sizeof(int32_t), st));
    *flags_out = v;
    return MGEA_OK;
}

int mgea_bert_stats(mgea_bert* h, int64_t* out) {
    MGEA_REQUIRE(h && out, MGEA_EINVAL, "bert_stats: NULL argument");
    std::lock_guard<std::mutex> lk(h->mu);
    for (int i = 0; i < 16; ++i) out[i] = 0;
    out[0] = h->n_forwards; out[1] = h->last_fold; out[2] = h->last_persistent; out[3] = h->last_ring; out[4] = h->last_small;
    out[5] = h->last_half_tiles; out[6] = h->last_ln_kernels; out[7] = h->last_cls_only;
    for (int e = 0; e < 6; ++e) out[8 + e] = h->last_epi[e];
    out[14] = h->last_rows;
    return MGEA_OK;
}

}  // extern "C"
