/*
 * mgea.h -- C ABI of the MI355X-native (gfx950) transformer-inference hot path of
 * RohitMurali18/Music-Generation-Emotion-Adaptive.
 *
 * The reference has no FFI / plugin interface: its hot path is reached by plain
 * Python attribute access from api_cache.py.  This header is therefore the boundary a maintainer
 * would bind from Python (ctypes stub in INTEGRATION.md); every entry point names the reference
 * code it replaces.  Conventions:
 *   - plain pointers and sizes only; every `*_dev` pointer is a device (HBM) pointer owned by the
 *     caller (in practice a torch.Tensor's data_ptr()) and only borrowed for the call, except the
 *     weight arena, which must outlive the handle created on it;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     all work is enqueued on it, nothing synchronises unless stated;
 *   - every function returns 0 on success or a negative MGEA_E* code; mgea_last_error() gives
 *     the thread-local message.  No exception crosses this boundary;
 *   - handles are internally locked: concurrent calls on one handle serialise (the reference's
 *     endpoint runs in FastAPI's thread pool, api_cache.py:186-187).
 */
#ifndef MGEA_H
#define MGEA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGEA_OK          0
#define MGEA_EINVAL     -1   /* bad shape / argument (python: ValueError / RuntimeError) */
#define MGEA_ENOMEM     -2   /* device allocation failed */
#define MGEA_EHIP       -3   /* a HIP call failed */
#define MGEA_ECAPACITY  -4   /* batch or context exceeds what the handle reserved */
#define MGEA_ENODEVICE  -5   /* no gfx950 device visible */

/* Storage / arithmetic modes.  Which engine accepts which:
 *   decoder (mgea_decoder_config.dtype): F32, F16        DistilBERT (mgea_bert_config.dtype): F32, BF16
 * F32 is the parity mode of both (bit-exact greedy ids / labels against the reference's fp32 CPU path). */
#define MGEA_DTYPE_F32   0   /* parity mode: fp32 storage, exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) */
#define MGEA_DTYPE_BF16  1   /* DistilBERT perf mode: bf16 weights + activations, fp32 accumulate (bf16 MFMA); calls of fewer than 512
                                tokens (one text per request) run on the exact-fp32 kernels of the same engine */
#define MGEA_DTYPE_F16   2   /* decoder perf mode: fp16 projection matrices + fp16 KV pages, fp32 accumulate (f16 MFMA).  DECODE steps and
                                prefills of up to 512 (row, token) pairs keep the residual stream, LayerNorm, softmax and logits in fp32.
                                A BIG prefill into an empty cache (>= 256 tiles of 256 x 256: e.g. [64, 1024]; switch decoder_prefill16)
                                runs on the f16 matrix cores with fp16 activations between the GEMMs -- the residual stream too, saturated
                                at +-65504 where it is written; accumulation, LayerNorm statistics, softmax and logits stay fp32.  Checked
                                against the oracle on synthetic weights only (tests/test_gpu_f16.py); a trained checkpoint whose residual
                                stream leaves the fp16 range would be clipped there: parity on trained weights is unpinned. */

#define MGEA_BLOCK_PRELN_GELU  0  /* api_cache.py:51-74 GPTBlock (KV-cache model, the default) */
#define MGEA_BLOCK_POSTLN_RELU 1  /* generate_music/generate.py:25-35 nn.TransformerEncoder twin */

#define MGEA_POS_REFERENCE 0 /* pos_emb[:T] per call: every decode step uses row 0 (api_cache.py:99) */
#define MGEA_POS_ABSOLUTE  1 /* true positions (build-defined extra, parity unpinned) */

#define MGEA_KV_PAGE_TOKENS 64 /* tokens per KV page (one wave64 tile) */

const char* mgea_last_error(void);
int mgea_version(void);
/* A/B and test switches (tools/README.md lists them).  The table is filled once, when the library is loaded, from the
 * MGEA_<NAME> environment variables; these two calls read / change an entry at run time.  Nothing on a launch path reads the
 * environment, and no switch is needed in production. */
int mgea_tune_set(const char* name, int32_t value);
int mgea_tune_get(const char* name, int32_t* value_out);
/* number of visible HIP devices, or a negative error */
int mgea_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Decoder: replaces GPTBlock / GPTWithKV / sample_kvcache (api_cache.py:39-106, 159-184).
 * ------------------------------------------------------------------------------------------ */
typedef struct mgea_decoder mgea_decoder;

typedef struct mgea_decoder_config {
    int32_t vocab;      /* len(tok2id), <= 14336 (sampler limit) api_cache.py:109 */
    int32_t seq_len;    /* rows of the position table         api_cache.py:36  */
    int32_t d_model;    /*                                    api_cache.py:37  */
    int32_t n_head;     /* reference hard-codes 8             api_cache.py:112 */
    int32_t n_layer;    /*                                    api_cache.py:31-32 */
    int32_t d_ff;       /* 4*d_model in the reference         api_cache.py:83  */
    int32_t max_batch;  /* rows the KV pool is reserved for */
    int32_t max_ctx;    /* tokens per row the KV pool is reserved for (may exceed seq_len: decode
                           steps use position row 0, so the cache can outgrow the table) */
    int32_t dtype;      /* MGEA_DTYPE_* */
    int32_t block_mode; /* MGEA_BLOCK_* */
    int32_t pos_mode;   /* MGEA_POS_* */
    float   ln_eps;     /* 1e-5 (nn.LayerNorm default) */
} mgea_decoder_config;

/* Sampler: replaces api_cache.py:169-178.  top_k == 1 is the greedy path (exact argmax, ties to
 * the lowest id); otherwise softmax(logits/temperature + (-1e10 outside top-k)) -- exactly top_k
 * entries survive like topk + scatter_ (api_cache.py:172-175), equal logits at the boundary by lowest id -- optionally cut
 * to the top_p nucleus, then one multinomial draw per row from a Philox4x32-10 stream keyed by
 * (seed, row, step) -- matches torch.multinomial in distribution only. */
typedef struct mgea_sampler_config {
    float    temperature;  /* > 0 */
    int32_t  top_k;        /* 0 = no top-k cut */
    float    top_p;        /* <= 0 or >= 1 = no nucleus cut (build-defined; not in the reference) */
    int32_t  eos_id;       /* -1 = none; a row that draws eos_id stops (api_cache.py:181) */
    uint64_t seed;
} mgea_sampler_config;

/* Weight arena: ONE contiguous fp32 device buffer holding every tensor (one RCCL broadcast moves
 * it).  Canonical tensor order, each tensor 256-byte aligned, torch layouts ([out, in] Linear):
 *   0 tok_emb [V,C]   1 pos_emb [L,C]
 *   per layer i (12 tensors, base 2+12*i): ln1_w ln1_b in_proj_w[3C,C] in_proj_b[3C]
 *       out_proj_w[C,C] out_proj_b ln2_w ln2_b fc1_w[F,C] fc1_b fc2_w[C,F] fc2_b
 *   then head_w [V,C], head_b [V].
 * (names after remap_state_dict, api_cache.py:118-134) */
int mgea_decoder_arena_layout(const mgea_decoder_config* cfg, int64_t* offsets_floats /* [n] or NULL */,
                              int32_t* n_tensors, int64_t* total_floats);

int mgea_decoder_create(const mgea_decoder_config* cfg, const float* arena_dev, mgea_decoder** out);
int mgea_decoder_destroy(mgea_decoder* h);
/* create() derives a decode-layout copy of the projection matrices from the arena.  If the caller
 * rewrites the arena afterwards (new checkpoint into the same tensor), call this before the next step. */
int mgea_decoder_refresh_weights(mgea_decoder* h, void* stream);

/* Forget all cached tokens and reserve KV pages for `batch` rows of up to `max_len` tokens. */
int mgea_decoder_reset(mgea_decoder* h, int32_t batch, int32_t max_len, void* stream);

/* model(idx, past_kv) (api_cache.py:87-106): append T new tokens per row to the cache and run the
 * NL blocks with every new token attending to the whole cache, no mask.  ids_dev [B,T] int32;
 * lens_dev [B] int32 or NULL (ragged rows: only the first lens[b] tokens of row b are real; the
 * rest are ignored and never cached, so each row equals its solo run).  logits_out_dev [B,T,V]
 * fp32 or NULL (the sampler's prefill discards them, api_cache.py:163): without a logits buffer nothing reads the last block's
 * output, and the call ends once that block's K | V are in the cache (switch decoder_prefill_full = 1 runs the whole block). */
int mgea_decoder_forward(mgea_decoder* h, const int32_t* ids_dev, const int32_t* lens_dev,
                         int32_t B, int32_t T, float* logits_out_dev, void* stream);

/* One decode step of the sampling loop (api_cache.py:167-179): feed ids_in_dev [B] (NULL = the
 * ids the previous step/generate produced), sample, write ids_out_dev [B] (NULL allowed) and
 * optionally the pre-temperature logits [B,V]. */
int mgea_decoder_step(mgea_decoder* h, const int32_t* ids_in_dev, const mgea_sampler_config* s,
                      int32_t* ids_out_dev, float* logits_out_dev, void* stream);

/* sample_kvcache (api_cache.py:159-184) for a batch: reset, prefill (logits dropped), then
 * n_steps decode steps, the first of which re-feeds each row's last prompt token.  The step is
 * captured once per (batch size, greedy | sampled) into a hipGraph and replayed; the sampler's
 * scalars (seed, temperature, top_k, top_p, eos_id) live in device memory, so a new request with
 * other values reuses the graph (mgea_decoder_stats counts instantiations).  ids_out_dev
 * [B, n_steps] int32; entries after a row's EOS are -1.  Host-synchronises only if eos_id >= 0
 * (to stop early once all rows ended). */
int mgea_decoder_generate(mgea_decoder* h, const int32_t* prompt_ids_dev, const int32_t* lens_dev,
                          int32_t B, int32_t Tp, int32_t n_steps, const mgea_sampler_config* s,
                          int32_t* ids_out_dev, void* stream);

/* Current cached length of each row -> lens_out_dev [B] (device int32). */
int mgea_decoder_context_lengths(mgea_decoder* h, int32_t* lens_out_dev, void* stream);

/* Per-kernel-class timing for bench.py's roofline leg: with stride n > 0 every n-th decode step of
 * generate() runs eagerly (not from the graph) with a hipEvent pair around each launch, recorded on
 * the launch stream; stride 0 switches it off.  profile_read() synchronises, sums the event pairs
 * into ms_by_class / launches_by_class (classes: 0 gemm, 1 row epilogues, 2 paged attention,
 * 3 dense attention, 4 logits+argmax / sampler; n_classes >= 5) and clears the records. */
int mgea_decoder_profile(mgea_decoder* h, int32_t stride);
int mgea_decoder_profile_read(mgea_decoder* h, double* ms_by_class, int64_t* launches_by_class,
                              int32_t n_classes);

/* out[0] kernels in the step graph last used, [1] graph replays of the last generate(), [2] graph
 * captures + instantiations over the handle's lifetime, [4] graphs cached now, [5] forwards that ran on the f16 matrix-core
 * prefill path (MGEA_DTYPE_F16 engines, empty cache, batch * T big enough: csrc/decoder.hip run_prefill16); others 0. */
int mgea_decoder_stats(mgea_decoder* h, int64_t* out /* [8] */);

/* Token ids outside [0, vocab) make nn.Embedding raise IndexError in the reference (api_cache.py:99).
 * Here they are clamped on the device and recorded in a sticky flag word, so that no call has to
 * synchronise to validate its input: this call synchronises `stream`, returns the flags (bit 0 = an id
 * was clamped since the last call) in *flags_out (host) and clears them. */
int mgea_decoder_error_flags(mgea_decoder* h, int32_t* flags_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * DistilBERT(+LoRA) classifier forward: replaces the model call inside
 * emotion_analysis/inference.py:16-20 (transformers DistilBertForSequenceClassification).
 * ------------------------------------------------------------------------------------------ */
typedef struct mgea_bert mgea_bert;

typedef struct mgea_bert_config {
    int32_t vocab, max_pos, dim, n_heads, n_layers, hidden, num_labels;
    int32_t max_tokens; /* B*S capacity of the activation workspace */
    int32_t dtype;      /* MGEA_DTYPE_* */
    float   ln_eps;     /* 1e-12 */
} mgea_bert_config;

/* Arena order (fp32, 256-byte aligned): word_emb[V,D] pos_emb[P,D] emb_ln_w emb_ln_b;
 * per layer (12 tensors): qkv_w[3D,D] (q_lin,k_lin,v_lin rows stacked, LoRA already merged)
 * qkv_b[3D] out_w[D,D] out_b sa_ln_w sa_ln_b lin1_w[Hd,D] lin1_b lin2_w[D,Hd] lin2_b
 * out_ln_w out_ln_b; then pre_w[D,D] pre_b cls_w[labels,D] cls_b. */
int mgea_bert_arena_layout(const mgea_bert_config* cfg, int64_t* offsets_floats, int32_t* n_tensors,
                           int64_t* total_floats);
int mgea_bert_create(const mgea_bert_config* cfg, const float* arena_dev, mgea_bert** out);
int mgea_bert_destroy(mgea_bert* h);
/* ids_dev [B,S] int32, mask_dev [B,S] int32 0/1 or NULL -> logits_out_dev [B,labels] fp32 and/or
 * argmax_out_dev [B] int32 (either may be NULL).  The classifier reads hidden_state[:, 0] of the last layer only: for S >= 4 that layer
 * projects K | V for every position and runs its query, attention, out-projection and FFN for the B [CLS] rows (same logits; switch
 * bert_full_last_layer = 1 computes every position). */
int mgea_bert_forward(mgea_bert* h, const int32_t* ids_dev, const int32_t* mask_dev, int32_t B,
                      int32_t S, float* logits_out_dev, int32_t* argmax_out_dev, void* stream);
/* PACKED forward: the real tokens of the B sequences back to back instead of [B, S] rows padded to the batch's longest
 * prompt (what the tokenizer call of emotion_analysis/inference.py:16 -- padding=True -- hands to the model).  ids_dev / pos_ids_dev
 * [n_tokens] int32 (token id; position of the token inside its sequence), cu_seqlens_dev [B + 1] int32 (sequence b = rows cu[b] ..
 * cu[b + 1] - 1; cu[0] = 0, cu[B] = n_tokens), max_len = the longest sequence.  Every row-wise GEMM, LayerNorm statistic
 * and attention tile then runs on real tokens only; same logits as the padded call with the corresponding prefix mask (a row's results
 * do not depend on the other rows of the batch, and a sequence attends to exactly its own keys in both forms).  Both engine modes take it:
 * F32 engines (and BF16 engines below their 512-token routing threshold) on the exact-fp32 kernels with sequences of any length up to
 * max_pos, BF16 engines from 512 tokens on with sequences of at most 256 tokens.  The caller guarantees a consistent cu_seqlens
 * (non-decreasing, cu[B] = n_tokens, cu[b + 1] - cu[b] in 1..max_len): it is device memory and is not read back. */
int mgea_bert_forward_packed(mgea_bert* h, const int32_t* ids_dev, const int32_t* pos_ids_dev, const int32_t* cu_seqlens_dev, int32_t B,
                             int32_t n_tokens, int32_t max_len, float* logits_out_dev, int32_t* argmax_out_dev, void* stream);
/* Token ids handed over as DEVICE memory are not read back before the forward (that would put a host sync in front of every call):
 * an id outside [0, vocab) -- nn.Embedding raises IndexError in the reference (emotion_analysis/inference.py:16-17) -- is clamped by the
 * embedding kernel and recorded in a sticky device flag.  This call synchronises `stream`, returns the flags (bit 0 = an id was clamped
 * since the last call) in *flags_out (host) and clears them; same contract as mgea_decoder_error_flags. */
int mgea_bert_error_flags(mgea_bert* h, int32_t* flags_out, void* stream);
/* What the handle ran (so that a test can assert WHICH kernels produced the numbers it checks): out[0] forwards so far; of the
 * last forward: [1] 1 = folded-LayerNorm bf16 pipeline, [2] / [3] / [4] bf16 GEMM launches on the persistent 256 x 256 kernel /
 * a ring kernel / the 128 x 128 kernel, [5] persistent launches that cut their left-over tiles into 128-row halves,
 * [6] LayerNorm kernel launches, [7] 1 when the last layer ran for the [CLS] rows only (K | V of every position, the rest on B rows:
 * the classifier reads nothing else; switch bert_full_last_layer = 1 computes every position), [8 + e] bf16 GEMM launches with
 * epilogue e (0..5), [14] rows the GEMMs ran on (padded call: B S; packed call: the real tokens); others 0. */
int mgea_bert_stats(mgea_bert* h, int64_t* out /* [16] */);

/* W[out,in] += scale * B[out,r] @ A[r,in] in place (peft LoRA fold, W' = W + (alpha/r) B A;
 * Scripts/finetuneDistillBert.ipynb:787-795). */
int mgea_lora_merge(float* w_dev, const float* a_dev, const float* b_dev, int32_t out_dim,
                    int32_t in_dim, int32_t r, float scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Op-level entry points (the kernels the engines are built from), exported so the parity
 * tests can check each kernel against the oracle in isolation.  All fp32, row-major.
 * ------------------------------------------------------------------------------------------ */
/* out[M, N] = A[M,K] @ W[N,K]^T, exact-fp32 MFMA, optional split-K (deterministic slab reduce).
 * K % 32 == 0.  workspace_dev: >= mgea_op_gemm_workspace_floats(M,N,split_k) floats. */
int64_t mgea_op_gemm_workspace_floats(int32_t M, int32_t N, int32_t split_k);
int mgea_op_gemm_f32(const float* a_dev, const float* w_dev, const float* bias_dev /* NULL ok */,
                     float* out_dev, int32_t M, int32_t N, int32_t K, int32_t split_k,
                     float* workspace_dev, void* stream);
/* y = LayerNorm(x) over the last dim C (C % 4 == 0, C <= 4096). */
int mgea_op_layernorm(const float* x_dev, const float* w_dev, const float* b_dev, float* y_dev,
                      int32_t M, int32_t C, float eps, void* stream);
/* Non-causal attention over a packed qkv buffer [B*T, 3C] (q | k | v, head h at column h*dh);
 * key validity = (t < lens[b] if lens) && (mask[b,t] != 0 if mask).  out [B*T, C]. */
int mgea_op_attention_f32(const float* qkv_dev, const int32_t* lens_dev, const int32_t* mask_dev,
                          float* out_dev, int32_t B, int32_t T, int32_t n_head, int32_t head_dim,
                          void* stream);
/* bf16 perf-mode kernels (MGEA_DTYPE_BF16 engines); bf16 buffers are raw 16-bit storage.
 * gemm: out[M,N] = epi(a[M,K] @ w[N,K]^T + bias), fp32 accumulate; epi 0 bias, 1 bias+GELU,
 * 2 bias+residual(res_dev [M,N] bf16).  K % 64 == 0, N % 4 == 0.  attention: head_dim 64 only. */
int mgea_op_f32_to_bf16(const float* src_dev, void* dst_dev, int64_t n, void* stream);
int mgea_op_gemm_bf16(const void* a_dev, const void* w_dev, const float* bias_dev, const void* res_dev,
                      void* out_dev, int32_t M, int32_t N, int32_t K, int32_t epi, void* stream);
/* The same GEMM with the LayerNorm-folding epilogues of the big-batch DistilBERT pipeline (persistent 256 x 256 kernel only:
 * M >= 512, N % 256 == 0; csrc/common.h BfEpiLn), so that the kernel combination the engine picks at B = 256 / S = 128 can be
 * checked in isolation:
 *   epi 3 / 4: out = rstd_row (a w'^T - mean_row c1) + c2 [+ GELU]: a = RAW rows, w' = bf16(W diag(gamma)), bias_dev = c2,
 *              rowstat_dev [M][2] = (mean, rstd) of the a rows (mgea_op_fold_ln_bf16 makes w' / c1 / c2);
 *   epi 5:     out = a w^T + bias + LayerNorm(res row; rowstat_dev, ln_g_dev, ln_b_dev); stats_out_dev [M][N / 256][2] (or NULL)
 *              receives per 256-column tile (sum, M2 about the tile mean) of every output row, which mgea_op_ln_rowstat turns
 *              into the next (mean, rstd);
 *   epi 0..2 as mgea_op_gemm_bf16 (the LayerNorm pointers are ignored).
 * info_out [2] (host, or NULL): [0] the kernel that ran (0 128 x 128 register-staged, 1 ring, 2 persistent), [1] 1 when the
 * persistent kernel cut its left-over tiles into two independent 128-row halves (tiles % CUs <= CUs / 2). */
int mgea_op_gemm_bf16_ln(const void* a_dev, const void* w_dev, const float* bias_dev, const void* res_dev, void* out_dev,
                         int32_t M, int32_t N, int32_t K, int32_t epi, const float* rowstat_dev, const float* c1_dev,
                         const float* ln_g_dev, const float* ln_b_dev, float* stats_out_dev, int32_t* info_out, void* stream);
int mgea_op_ln_rowstat(const float* part_dev, float* rowstat_dev, int32_t M, int32_t n_part, int32_t C, float eps,
                       void* stream);
/* wf_out_dev [N,K] bf16 = bf16(W diag(gamma)); c1[n] = sum_k of the ROUNDED wf[n,k]; c2[n] = bias[n] + sum_k W[n,k] beta[k]. */
int mgea_op_fold_ln_bf16(const float* w_dev, const float* gamma_dev, const float* beta_dev, const float* bias_dev, int32_t N,
                         int32_t K, void* wf_out_dev, float* c1_out_dev, float* c2_out_dev, void* stream);
int mgea_op_attention_bf16(const void* qkv_dev, const int32_t* mask_dev, void* out_dev, int32_t B, int32_t T,
                           int32_t n_head, int32_t head_dim, void* stream);
int mgea_op_layernorm_bf16(const void* x_dev, const float* w_dev, const float* b_dev, void* y_dev, int32_t M,
                           int32_t C, float eps, void* stream);
/* Layouts of the fused decode path.  The skinny GEMM reads both operands in MFMA-fragment order so that
 * every wave load is 1 KB of consecutive bytes (see csrc/common.h):
 *   tile_weights: W [N,K] row-major -> out_dev [mgea_op_tiled_weight_floats(N,K)] (rows padded to 32);
 *   tile_rows:    [M<=512, N] row-major <-> the k-tiled activation buffer (whole 64-row groups of 64 * N floats), to_tiled != 0
 *                 converts row-major -> tiled.  K % 32 == 0, N % 32 == 0. */
int64_t mgea_op_tiled_weight_floats(int32_t N, int32_t K);
int mgea_op_tile_weights(const float* w_dev, int32_t N, int32_t K, float* out_dev, void* stream);
int mgea_op_tile_rows(const float* src_dev, float* dst_dev, int32_t M, int32_t N, int32_t to_tiled, void* stream);
/* LayerNorm folded into the matrix it feeds: wt_out_dev = tiles of gamma[k] * W[n,k], c1[n] = the row sums of those
 * products, c2[n] = sum_k beta[k] * W[n,k] + bias[n], so that LN(x) @ W^T + bias = rstd * (x @ W'^T - mean * c1) + c2. */
int mgea_op_fold_ln(const float* w_dev, const float* gamma_dev, const float* beta_dev, const float* bias_dev,
                    int32_t N, int32_t K, float* wt_out_dev, float* c1_out_dev, float* c2_out_dev, void* stream);
/* Fused skinny GEMM (decode step, M <= 512): out = epilogue(A @ W^T + bias); epi 1 = residual add into out +
 * LayerNorm partial stats, 2 = activation (0 none, 1 GELU, 2 ReLU), 3 = LM head (logits [M,N] row-major in out_dev
 * or NULL, per-tile (max, argmax) partials in stats_out_dev).  a_dev and out_dev are k-tiled activation buffers,
 * w_dev is a tiled weight (above).  Folded LayerNorm of A when ln_c1_dev != NULL: w_dev / ln_c1_dev / bias_dev are
 * mgea_op_fold_ln's wt / c1 / c2 and stats_in_dev [M][n_part][2] holds partial (mean, M2) over part_cnt columns
 * each (n_part even, <= 64).  dbg = 0 (tools/skinny_bench.py, tools/skinny_phases.py). */
int mgea_op_skinny(int32_t epi, const float* a_dev, const float* w_dev, const float* bias_dev,
                   const float* ln_c1_dev, const float* stats_in_dev, int32_t n_part,
                   int32_t part_cnt, float* out_dev, float* stats_out_dev, int32_t M, int32_t N, int32_t K,
                   int32_t act, int32_t dbg, void* stream);
/* Number P of (max, argmax) partials per row that mgea_op_skinny(epi = 3) of this shape writes: stats_out_dev receives the maxima as
 * float [row][P] (row stride P) followed, 64 * P floats in, by the int32 argmax indices in the same layout.  (The decode-step head of
 * api_cache.py:105 runs as ONE balanced round of the chip where the shape allows -- csrc/head_gemm.hip, P = the CU count -- and on
 * the generic skinny kernel otherwise; switch head_balanced.) */
int mgea_op_skinny_logits_partials(int32_t M, int32_t N, int32_t K);
/* Sampler on a logits matrix [B,V]; step selects the Philox counter.  probs_out_dev [B,V] or NULL
 * receives the pre-multinomial distribution. */
int mgea_op_sample(const float* logits_dev, int32_t B, int32_t V, const mgea_sampler_config* s,
                   int64_t step, int32_t* ids_out_dev, float* probs_out_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MGEA_H */
