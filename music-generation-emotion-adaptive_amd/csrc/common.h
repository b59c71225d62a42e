// Shared host/device helpers for the gfx950 kernels.  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/mgea.h"

namespace mgea {

// ---- error plumbing (thread-local message behind mgea_last_error) --------------------------
void set_error(const char* fmt, ...);
const char* get_error();

#define MGEA_CHECK_HIP(expr)                                                          \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            ::mgea::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),  \
                              __FILE__, __LINE__);                                    \
            return MGEA_EHIP;                                                         \
        }                                                                             \
    } while (0)

#define MGEA_REQUIRE(cond, code, ...)            \
    do {                                         \
        if (!(cond)) {                           \
            ::mgea::set_error(__VA_ARGS__);      \
            return (code);                       \
        }                                        \
    } while (0)

#define MGEA_TRY(expr)               \
    do {                             \
        int _rc = (expr);            \
        if (_rc != MGEA_OK) return _rc; \
    } while (0)

// ---- A/B and test switches (tools/README.md) ------------------------------------------------
// One process-global table, filled ONCE when the library is loaded from the MGEA_<NAME> environment variables and changed at
// run time only through mgea_tune_set() (tests, tools): nothing on a launch path reads the environment.
enum { TUNE_BF16_GEMM_TILE = 0,   // 0 = the launcher's choice; 1 128x128 / 2 256x128 / 3 256x256 ring kernels; 4 = the persistent 256x256 kernel
       TUNE_BF16_GEMM_SMALL,      // 1: the register-staged 128x128 kernel for every shape
       TUNE_BF16_GEMM_TAIL,       // persistent kernel, tiles left after the full rounds: 0 whole tiles, 1 two 128-row halves, 2 = 1 + staggered order
       TUNE_BF16_GEMM_PHASES,     // whole tiles of the persistent kernel: 4 phases of 16 MFMAs per K-tile, 2 (default) phases of 32, or 1 = software-pipelined, one barrier per K-tile
       TUNE_BF16_GEMM_REVERSE,    // 1 (default): the FFN down-projection walks its tiles from the end of each XCD's run (A = the up-projection's output)
       TUNE_BERT_BF16_NOFOLD,     // 1: bf16 DistilBERT with LayerNorm kernels instead of the folded-LayerNorm pipeline
       TUNE_DECODER_PREFILL_FULL, // 1: a decoder forward whose logits are dropped still runs the whole last block (default: it stops at that block's K | V)
       TUNE_BERT_FULL_LAST_LAYER, // 1: the last DistilBERT layer computes every position (as the reference does) instead of K | V for all + the rest for the [CLS] rows only
       TUNE_DECODER_UNFUSED, TUNE_DECODER_NOGEMV, TUNE_DECODER_NOGRAPH,   // read by mgea_decoder_create()
       TUNE_ATTN16_WIDE,          // 16-bit flash attention with 8 waves / 256-key stages: 0 never, 1 from 512 tokens on (default), 2 always
       TUNE_DECODER_PREFILL16_PAGES,     // 1 (default): the fp16 prefill writes K | V pages from inside its attention kernel (the rows are in LDS there); 0: a scatter kernel per layer re-reads them from the qkv buffer
       TUNE_DECODER_PREFILL16,    // fp16 engines, big-batch prefill: 0 keep the exact-fp32 kernels (A/B), 1 f16 matrix cores when the batch
                                  // fills the chip (default), 2 whenever the kernels accept the shape (tests)
       TUNE_HEAD_BALANCED,        // 1 (default): the LM head of a 3..64-row decode step on head_balanced_kernel (one workgroup per CU, equal unit counts); 0: the generic skinny kernel
       TUNE_ATTN16_PIPE,          // 1 (default): the 16-bit flash attention runs full key stages software-pipelined (next tile's Q K^T under this tile's exponentials); 0: rolled loop
       TUNE_SKINNY_ONE_PER_CU,    // 1: skinny GEMM launches of <= 256 workgroups ask for > 80 KB of LDS, so that no two share a CU (A/B; default 0)
       TUNE_SAMPLER_WAVE_SELECT,  // 1 (default): top_k <= 64 finds its boundary wave by wave (ballots only) and merges 4 x 64 candidates after one barrier; 0: block-wide bisection, a barrier per bit
       TUNE_ATTN_SPLIT,           // the decode step's attention of at most this many (row, head) pairs spreads each pair's KV pages over several workgroups, the last one to arrive merges their partials (default 64 = 8 rows of 8 heads: measured +3..5 % tokens/s at 1-8 rows, nothing at 16, -3.5 % at 32; 0: always one workgroup per pair)
       TUNE_DECODER_GRAPH_STEPS,  // decode steps per hipGraph launch of mgea_decoder_generate: 1, 2, 4, 8 (default) or 16; the single-step graph serves the remainder (round 4: 289.4 -> 287.6 us per step at B = 64, 168.3 -> 163.4 at B = 1)
       TUNE_ATTN_ARITH_PAGES,     // 1 (default): the decode attention computes physical page ids (j * batch + b, the decoder's own allocation) instead of loading them from the page table -- one dependent scalar load less per page; 0: always the table
       TUNE_COUNT };
int tune(int key);

// per-device facts the launchers need (cached per device id; one process may drive several devices)
struct DeviceInfo { int dev; int n_cu; };
int device_info(DeviceInfo* out);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel instantiation, device): `done` is the instantiation's own bit mask
int set_max_dynamic_lds(const void* fn, int bytes, int dev, uint64_t* done);

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// "k-tiled" activation layout of the fused decode path, in MFMA-fragment order: a [<=64, N] matrix is cut
// into 32-wide k-chunks of 2048 floats; inside a chunk the 16-byte group holding (row, k..k+3) sits at
//   [row / 16][h = (k / 4) % 2][lane = (k / 8) % 4 * 16 + row % 16]
// which is exactly the order in which the 64 lanes of a wave consume it in the skinny GEMM (lane = 16 g + c
// feeds row c with k = k0 + 8 g + 4 h + s): one operand load instruction reads 1 KB of CONSECUTIVE bytes.
// Measured on MI355X: a wave load whose lanes each fetch 16 B from a different 128-B line is served at about
// one lane per clock (~16 B/clk, 36 GB/s per CU); consecutive lanes reach the 64 B/clk of the L1.
// Rows beyond 64 (decode batches of up to MGEA_FUSED_MAX_ROWS rows) continue in further 64-row groups of 64 * N floats each.
constexpr int MGEA_FUSED_MAX_ROWS = 512;
__host__ __device__ static inline int64_t tiled_off(int row, int n, int N) {
    const int r = row & 63;
    return (int64_t)(row >> 6) * 64 * N + (int64_t)(n >> 5) * 2048 +
           ((((r >> 4) * 2 + ((n >> 2) & 1)) * 64 + ((n >> 3) & 3) * 16 + (r & 15)) << 2) + (n & 3);
}
// The same idea for the weights of the skinny GEMMs (W [N, K] row-major in the arena): per (16-row tile,
// 32-wide k-chunk) one 2 KB block [h][lane = 16 g + c][4] holding W[tile * 16 + c][chunk * 32 + 8 g + 4 h + s];
// rows are padded to a multiple of 32 with zeros.  Built once per engine from the arena (launch_tile_weights).
static inline int64_t wtile_floats(int N, int K) { return round_up(N, 32) * (int64_t)K; }
int launch_tile_weights(const float* W, int N, int K, float* out, hipStream_t st, const float* gamma = nullptr);
// fp16 twin (MGEA_DTYPE_F16): per (16-row tile, 32-wide k-chunk) one 1 KB block [lane = 16 g + c][8] of _Float16 holding
// W[tile * 16 + c][chunk * 32 + 8 g + 0..7] -- the A/B fragment of v_mfma_f32_16x16x32_f16; out has wtile_floats(N, K) halves
int launch_tile_weights_f16(const float* W, int N, int K, void* out, hipStream_t st);
// c1[n] = sum_k gamma[k] * W[n,k], c2[n] = sum_k beta[k] * W[n,k] + bias[n] (fp64 sums of exact products): the folded-LayerNorm
// vectors when gamma is applied on the activation side (fp16 mode)
int launch_ln_vectors(const float* W, const float* gamma, const float* beta, const float* bias, int N, int K, float* c1, float* c2,
                      hipStream_t st);
// x[i] <- float(half(x[i])) in place (round-to-nearest-even): the fp16 engine's private fp32 copy of the matrices
int launch_round_f16_inplace(float* x, int64_t n, hipStream_t st);
// LayerNorm folded into the matrix it feeds: out = tiles of gamma[k] * W[n,k]; c1[n] = sum_k of those products,
// c2[n] = sum_k beta[k] * W[n,k] + bias[n] (sums in fp64, once per engine)
int launch_ln_fold(const float* W, const float* gamma, const float* beta, const float* bias, int N, int K, float* out,
                   float* c1, float* c2, hipStream_t st);
// row-major [M, N] <-> k-tiled (to_tiled != 0: src row-major, dst tiled), for tests and tools
int launch_tile_rows(const float* src, float* dst, int M, int N, int to_tiled, hipStream_t st);

// ---- kernel launchers (defined in the .hip files) -------------------------------------------

// P[z][M, ldp] = A[M, k-slice z] @ W[N, k-slice z]^T   (raw partial products, no bias)
// ldp = round_up(N, 64); slab stride = M * ldp floats.  K % 32 == 0.
int launch_gemm_f32(const float* A, int lda, const float* W, int ldw, float* P, int M, int N, int K,
                    int split_k, hipStream_t st);
int launch_gemm_f32_bias_act(const float* A, int lda, const float* W, int ldw, const float* bias, float* out, int ldo,
                             int M, int N, int K, int act, hipStream_t st);
// chooses split_k so the grid fills the chip (deterministic function of the shape)
int pick_split_k(int M, int N, int K, int64_t cap_floats = -1);   // cap: slab workspace of the caller (floats)
bool gemm_direct_epilogue_ok(int M, int N);
static inline int64_t slab_ld(int N) { return round_up(N, 64); }
static inline int64_t slab_floats(int M, int N) { return (int64_t)M * slab_ld(N); }

// Row epilogues over S slabs P (slab stride `ps`, leading dim `ldp`): v = sum_s P[s][m][n] + bias[n]
enum { ACT_NONE = 0, ACT_GELU = 1, ACT_RELU = 2 };
// out[m, n] = act(v)
int launch_bias_act(const float* P, int S, int64_t ps, int ldp, const float* bias, float* out, int ldo,
                    int M, int N, int act, hipStream_t st);
// pre-LN :  x += v;  xn = LN(x; lnw, lnb) (xn written only if lnw != NULL)
// post-LN:  x = LN(x + v; lnw, lnb)
int launch_bias_res_ln(const float* P, int S, int64_t ps, int ldp, const float* bias, float* x, float* xn,
                       const float* lnw, const float* lnb, float eps, int M, int C, int post_ln,
                       hipStream_t st);
int launch_layernorm(const float* x, const float* w, const float* b, float* y, int M, int C, float eps,
                     hipStream_t st);

// Paged KV pool: [n_layer][n_pages][K | V][H][one page = 64 tokens x dh elements]; elements are fp32 (parity mode) or
// fp16 (MGEA_DTYPE_F16).  Inside a page, with G = the elements of one 16-byte group (4 floats / 8 halves):
//   K: [dh / G][64 tokens][G]  (token-major inside a d-group: in the decode kernel lane = token, every wave load
//                               instruction is 1 KiB of consecutive bytes)
//   V: [64 tokens][dh]         (a wave instruction reads 1 KiB = several whole rows)
struct KvPool {
    void* base;
    int32_t n_pages;      // physical pages per layer
    int32_t H, dh;
    int64_t layer_stride; // elements
    int32_t f16;          // 0: float elements, 1: _Float16 elements
    int32_t arith_batch;  // > 0: the page table holds logical page j of row b -> physical j * arith_batch + b (what mgea_decoder_reset
                          // writes): a kernel may compute that instead of loading it; 0: only the table says
    __host__ __device__ int64_t page_elems() const { return (int64_t)MGEA_KV_PAGE_TOKENS * dh; }
    __host__ __device__ int elt_bytes() const { return f16 ? 2 : 4; }
};

// decoder embedding: x[m] = tok_emb[ids[m]] + pos_emb[pos]; xn = LN(x) if lnw.
// rows m = b*T + t; rows with t >= lens[b] are zero-filled.  pos = t (+ ctx_len[b] if absolute).
// err_flag (device int32 or NULL): bit 0 is set when a real token id lies outside [0, vocab) (it is clamped).
int launch_embed_ln(const int32_t* ids, const int32_t* lens, const int32_t* ctx_len, const float* tok_emb,
                    const float* pos_emb, float* x, float* xn, const float* lnw, const float* lnb, float eps,
                    int B, int T, int C, int vocab, int pos_rows, int absolute_pos, int32_t* err_flag, hipStream_t st);
// BERT embedding: h[m] = LN(word[ids[m]] + pos[t])
int launch_bert_embed_ln(const int32_t* ids, const float* word, const float* pos, const float* lnw,
                         const float* lnb, float eps, float* h, int B, int S, int D, int vocab,
                         hipStream_t st, int32_t* err_flag = nullptr, const int32_t* pos_ids = nullptr);   // pos_ids: packed rows (B = rows, S = 1)
// qkv epilogue of the decoder: v = sum P + bias over [M, 3C]; q -> qbuf[m, C] (and k|v -> kvbuf
// [m, 2C] if kvbuf != NULL, for the no-cache attention); k, v of real tokens -> KV pages of `layer`
// at position ctx_len[b] + t.
int launch_qkv_scatter(const float* P, int S, int64_t ps, int ldp, const float* bias, float* qkv_out,
                       const KvPool& pool, int layer, const int32_t* page_table, int max_pages,
                       const int32_t* ctx_len, const int32_t* lens, int B, int T, int C, hipStream_t st);
// decode / extend attention over the paged cache: query rows m = b*T + t attend to
// ctx_len[b] + (lens ? lens[b] : T) cached tokens.  q read from qkv[m, 0:C] (row stride 3C).
// split (small batches): scratch of the split-context form -- part [max_items][max_split][attn_part_floats(dh)] floats, count
// [max_items] int32 zeroed once (the kernel leaves them zero); NULL, or a shape attn_split_count() answers 1 for: one workgroup per
// (row, head, query).
constexpr int MGEA_ATTN_MAX_SPLIT = 16, MGEA_ATTN_SPLIT_ITEMS = 256;
__host__ __device__ constexpr int attn_part_floats(int dh) { return dh + 2; }   // acc[dh], max, sum
struct AttnSplit { float* part; int32_t* count; int max_split, max_items; };
int attn_split_count(int B, int H, int T, int max_pages);
int launch_attn_paged(const float* qkv, const KvPool& pool, int layer, const int32_t* page_table,
                      int max_pages, const int32_t* ctx_len, const int32_t* lens, float* out, int B, int T,
                      int C, int tiled_out, hipStream_t st, const AttnSplit* split = nullptr);
// dense non-causal attention over the qkv buffer itself (prefill without past, BERT)
int launch_attn_dense(const float* qkv, const int32_t* lens, const int32_t* mask, float* out, int B, int T,
                      int H, int dh, int tiled_out, hipStream_t st, const int32_t* cu = nullptr);   // cu: packed rows (T = the longest sequence)

// The sampler's scalars as the kernels read them from DEVICE memory (one 32-byte record per engine): a captured
// decode-step graph holds only the pointer, so one graph serves every request whatever its seed / temperature /
// top-k / top-p / EOS id (mgea_sampler_config, api_cache.py:160,204).
struct SamplerParams {
    float temperature; int32_t top_k; float top_p; int32_t eos_id;
    uint32_t seed_lo, seed_hi; int32_t pad0, pad1;
};
static inline SamplerParams sampler_params(const mgea_sampler_config& s) {
    return SamplerParams{s.temperature, s.top_k, s.top_p, s.eos_id, (uint32_t)s.seed, (uint32_t)(s.seed >> 32), 0, 0};
}
constexpr int MGEA_SAMPLER_MAX_VOCAB = 14336;   // the sampler keeps a row in registers: 256 threads x 56 logits

// logits row epilogue: v = sum P + bias; optional store to logits[m, V]; greedy argmax path writes
// next ids and advances the per-row state (see decoder.hip).
struct StepState {
    int32_t* cur_ids;   // [B] token fed to the next step
    int32_t* ctx_len;   // [B]
    int32_t* done;      // [B]
    int32_t* row_step;  // [B] index of the step each row is producing (also the Philox counter)
    int32_t* n_done;    // [1]
    int32_t* ids_out;   // [B, n_steps] or NULL
    int32_t  n_steps;
    int32_t  eos_id;            // used when params == NULL
    const SamplerParams* params;   // device record whose eos_id wins (NULL: eos_id above)
    __host__ __device__ int eos() const { return params ? params->eos_id : eos_id; }
};
// one query per sequence (the [CLS] position, fp32 q [B, D]) against the K | V columns of a packed qkv buffer [B * S, 3 D] (fp32 or bf16) -> fp32 [B, D]
int launch_attn_cls(const float* q, const void* qkv, int qkv_bf16, const int32_t* mask, float* out, int B, int S, int H, int dh, hipStream_t st,
                    const int32_t* cu = nullptr);
int launch_logits_argmax(const float* P, int S, int64_t ps, int ldp, const float* bias, float* logits,
                         int M, int V, int32_t* argmax_out, hipStream_t st);
// sampler over logits [B,V] (top_k != 1); writes ids[b]; probs_out optional
// tail != NULL: the sampler also does the loop bookkeeping of the row and the NEXT step's embedding (advance_embed_row)
struct TailArgs {
    StepState s;
    const float* tok_emb; const float* pos_emb;
    float* x; float* stats;     // k-tiled residual stream and its LayerNorm partials (fused decode path)
    int C, vocab, pos_rows, absolute_pos;
};
// params_dev != NULL: the scalars come from that device record instead of `s`
int launch_sample(const float* logits, int B, int V, const mgea_sampler_config& s, const SamplerParams* params_dev,
                  const int32_t* row_step_dev, int64_t step_host, int32_t* ids_out, float* probs_out, hipStream_t st,
                  const TailArgs* tail = nullptr);
int launch_set_sampler_params(SamplerParams* params_dev, const mgea_sampler_config& s, hipStream_t st);
// after ids for this step are in `sampled` [B]: apply EOS/done logic, write ids_out[b, step],
// cur_ids, ctx_len += 1, row_step += 1
int launch_advance(const int32_t* sampled, const StepState& s, int B, hipStream_t st);
// ctx_len[b] += (lens ? lens[b] : T)
int launch_add_lens(int32_t* ctx_len, const int32_t* lens, int T, int B, hipStream_t st);
// cur_ids[b] = ids[b, (lens ? lens[b] : T) - 1]
int launch_take_last(const int32_t* ids, const int32_t* lens, int32_t* cur_ids, int B, int T, hipStream_t st);
int launch_gather_rows(const float* src, int ld_src, float* dst, int ld_dst, int rows, int row_step, int C,
                       hipStream_t st, const int32_t* row_idx = nullptr);   // row_idx: source row of output row r (instead of r * row_step)
int launch_lora_merge(float* w, const float* a, const float* b, int out_dim, int in_dim, int r, float scale,
                      hipStream_t st);

// ---- bf16 perf-mode kernels (bf16.hip); bf16 buffers travel as void* ---------------------------
int launch_f32_to_bf16(const float* src, void* dst, int64_t n, hipStream_t st, int f16 = 0);   // f16: _Float16 instead of bf16
// C = epi(A @ W^T + bias): epi 0 bias, 1 bias + GELU, 2 bias + residual(res, ld = ldc)
// LayerNorm folded around the bf16 GEMMs (bf16.hip, epilogues 3-5 of launch_gemm_bf16; persistent 256 x 256 kernel only).
//   epi 3 / 4 (LNFOLD / +GELU): A = RAW rows, W = W diag(gamma) in bf16, c1[n] = sum_k W'[n, k], bias slot = c2 = b + W beta;
//                               rowstat = (mean, rstd) of the A rows, [M][2]; out = rstd (A W'^T - mean c1) + c2
//   epi 5 (RES_LN): out = A W^T + bias + LayerNorm(res row) with the residual's rowstat / ln_g / ln_b (an already normalised
//                   residual comes with identity tables: mean 0, rstd 1, gamma 1, beta 0); stats_out (or nullptr) receives (sum, M2 =
//                   sum of squared deviations from the tile's own mean) of every output row per 256-column tile, [M][N / 256][2],
//                   from which launch_ln_rowstat makes the next (mean, rstd).
struct BfEpiLn { const float* rowstat; const float* c1; const float* ln_g; const float* ln_b; float* stats_out; };
// What a launch_gemm_bf16 call ran (optional out-parameter; the engines count these for mgea_bert_stats): kernel 0 = the
// register-staged 128 x 128 kernel, 1 = a ring kernel, 2 = the persistent phase-interleaved 256 x 256 kernel, 3 = the two-workgroups-per-CU
// 256 x 128 kernel (switch bf16_gemm_tile = 5 only); half_tiles = 1 when the persistent
// kernel cuts the tiles left over after its full rounds into two 128-row halves (bf16.hip, "HALF-TILE TAIL").
// reverse (IN): ask the persistent kernel to walk its tiles from the end of every XCD's run (switch bf16_gemm_reverse; for a GEMM whose A
// operand is the previous kernel's large output: see the kernel).
struct GemmBf16Info { int kernel; int half_tiles; int reverse; };
int launch_gemm_bf16(const void* A, int lda, const void* W, int ldw, const float* bias, const void* res, void* C,
                     int ldc, int M, int N, int K, int epi, hipStream_t st, GemmBf16Info* info = nullptr, const BfEpiLn* ln = nullptr,
                     int f16 = 0);   // f16 != 0: _Float16 operands / outputs (persistent kernel, epilogues 3 / 4 / 5, and 6 = fp32 output)
// true when launch_gemm_bf16 would run this shape on the persistent 256 x 256 kernel (the only one with epilogues 3-5)
bool gemm_bf16_is_persistent(int M, int N, int K);
int launch_ln_rowstat(const float* part, float* rowstat, int M, int n_part, int C, float eps, hipStream_t st);
int launch_fold_ln_weights_bf16(const float* W, const float* gamma, const float* beta, const float* b, void* Wf, float* c1, float* c2,
                                int N, int K, hipStream_t st, int f16 = 0);
// cu (here and below; NULL = padded [B, S] rows): packed input, the rows of sequence b are cu[b] .. cu[b + 1] - 1
int launch_gather_cls_ln_bf16(const void* h, const float* rowstat, const float* g, const float* be, float* out, int B, int S, int D,
                              hipStream_t st, const int32_t* cu = nullptr);
int launch_layernorm_bf16(const void* x, const float* w, const float* b, void* y, int M, int C, float eps, hipStream_t st);
int launch_bert_embed_ln_bf16(const int32_t* ids, const float* word, const float* pos, const float* lnw, const float* lnb,
                              float eps, void* h, int B, int S, int D, int vocab, hipStream_t st, int32_t* err_flag = nullptr,
                              const int32_t* pos_ids = nullptr);
int launch_gather_cls_bf16(const void* h, float* out, int B, int S, int D, hipStream_t st, const int32_t* cu = nullptr);
// pages (fp16 instantiation only, or NULL): the K | V rows of every key whose mask bit is set also go to the fp16 KV pages of `layer` at
// position t (decoder prefill into an empty cache: the attention kernel has those rows in LDS anyway)
struct KvPages { KvPool pool; int layer; const int32_t* page_table; int max_pages; };
int launch_attn_bf16(const void* qkv, const int32_t* mask, void* out, int B, int T, int H, int dh, hipStream_t st, int f16 = 0,
                     const KvPages* pages = nullptr, const int32_t* cu = nullptr);

// fp16 big-batch prefill of the decoder (bf16.hip): embedding rows as fp16 + their (mean, rstd); K | V of fp16 qkv rows -> fp16 KV pages
int launch_dec_embed_f16(const int32_t* ids, const int32_t* lens, const int32_t* ctx_len, const float* tok_emb, const float* pos_emb,
                         void* x, float* rowstat, int32_t* mask_out, float eps, int B, int T, int C, int vocab, int pos_rows,
                         int absolute_pos, int32_t* err_flag, hipStream_t st);
int launch_kv_scatter_f16(const void* qkv, const KvPool& pool, int layer, const int32_t* page_table, int max_pages, const int32_t* ctx_len,
                          const int32_t* lens, int B, int T, int C, hipStream_t st);

// ---- fused skinny GEMM (decode step, M <= MGEA_FUSED_MAX_ROWS): gemm_skinny.hip --------------------------------
enum { EPI_QKV = 0, EPI_RES = 1, EPI_ACT = 2, EPI_LOGITS = 3 };

struct SkinnyArgs {
    const float* A; int lda;
    const float* W;            // [N, K] in the tiled weight layout (launch_tile_weights); fp16 tiles (launch_tile_weights_f16) if w_f16
    int w_f16;                 // MGEA_DTYPE_F16 engines: W holds _Float16 fragments, A is rounded to fp16 on load, f16 MFMA, fp32 accumulate
    const float* bias;         // [N] or NULL
    int M, N, K;
    // folded LayerNorm (ln_c1 != NULL): W holds gamma * W, ln_c1 [N] = its row sums, bias = beta @ W^T + bias
    // (launch_ln_fold); per-row partial stats [64][n_part][2], each over `part_cnt` elements
    const float* ln_c1; float eps;
    const float* ln_g; const float* ln_b;   // launch_gemv: LayerNorm gamma / beta applied directly (W row-major, unfolded);
                                            // w_f16 with ln_c1: gamma is applied to A on load (W stays the plain rounded matrix,
                                            // ln_c1 = sum_k gamma_k W[n,k], bias = sum_k beta_k W[n,k] + b[n])
    const float* stats_in; int n_part; int part_cnt;
    float inv_n_part, inv_k;   // 1 / n_part and 1 / (n_part * part_cnt), filled by the launcher (no division in the kernel)
    // outputs
    float* out; int ldo;       // QKV: qkv_out [M, N]; RES: x [M, N] (in place); ACT: out [M, ldo]; LOGITS: logits or NULL
    float* stats_out;          // RES: [64][N/16][2]
    int act;
    // QKV scatter
    KvPool pool; int layer; const int32_t* page_table; int max_pages; const int32_t* ctx_len;
    const int32_t* lens; int T; int C;
    // LOGITS
    float* pmax_val; int32_t* pmax_idx;   // [64][n_tiles]
    int dbg;                   // ablation bits for tools/skinny_bench.py (0 in production)
    int nw;                    // waves per workgroup (set by the launcher)
};
int launch_skinny(int epi, const SkinnyArgs& a, hipStream_t st);
// M <= 2 rows (single-stream decode): wave-level dot products on the ROW-MAJOR arena weights (gemv_small.hip)
bool gemv_shape_ok(int M, int N, int K);
int launch_gemv(int epi, const SkinnyArgs& a, hipStream_t st);
int skinny_logits_tiles(int M, int N, int K);   // (max, argmax) partials per row that launch_skinny(EPI_LOGITS) of this shape writes
// head_gemm.hip: the LM head of a decode step as one balanced round of the chip; launch returns 1 when the shape is not its own
int head_balanced_partials(int M, int N, int K);
int launch_head_balanced(const SkinnyArgs& a, hipStream_t st);
int launch_embed_stats(const int32_t* ids, const int32_t* lens, const int32_t* ctx_len, const float* tok_emb,
                       const float* pos_emb, float* x, float* stats, int B, int T, int C, int vocab, int pos_rows,
                       int absolute_pos, int32_t* err_flag, hipStream_t st);
int launch_argmax_advance(const float* pval, const int32_t* pidx, int n_tiles, const StepState& s, int32_t* sampled,
                          int B, hipStream_t st);
int launch_argmax_advance_embed(const float* pval, const int32_t* pidx, int n_tiles, const StepState& s, int32_t* sampled,
                                const float* tok_emb, const float* pos_emb, float* x, float* stats, int B, int C, int vocab,
                                int pos_rows, int absolute_pos, hipStream_t st);

}  // namespace mgea

// ---- device helpers --------------------------------------------------------------------------
#ifdef __HIPCC__
namespace mgea {
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// non-temporal 16-byte load for data streamed once (KV pages): does not displace L2/MALL lines
__device__ __forceinline__ float4 ldnt4(const float* p) {
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
// 8 floats -> 8 halves, round-to-nearest-even (v_cvt_f16_f32)
__device__ __forceinline__ h16x8 to_h8(float4 a, float4 b) {
    return (h16x8){(_Float16)a.x, (_Float16)a.y, (_Float16)a.z, (_Float16)a.w, (_Float16)b.x, (_Float16)b.y, (_Float16)b.z, (_Float16)b.w};
}
// the 4 consecutive elements d .. d+3 (d % 4 == 0) of head `head`, token slot `slot`, of the K (isv = 0) or V page `phys`
__device__ __forceinline__ void kv_store4(const KvPool& pool, int layer, int phys, int isv, int head, int slot, int d, float4 v) {
    const int64_t pe = pool.page_elems();
    const int64_t page = layer * pool.layer_stride + ((int64_t)(phys * 2 + isv) * pool.H + head) * pe;
    if (pool.f16) {
        _Float16* p = static_cast<_Float16*>(pool.base) + page + (isv ? slot * pool.dh + d : ((d >> 3) * MGEA_KV_PAGE_TOKENS + slot) * 8 + (d & 7));
        *reinterpret_cast<h16x4*>(p) = (h16x4){(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    } else {
        float* p = static_cast<float*>(pool.base) + page + (isv ? slot * pool.dh + d : ((d >> 2) * MGEA_KV_PAGE_TOKENS + slot) * 4);
        st4(p, v);
    }
}

// End of a decode step for row b, by one workgroup of >= 256 threads whose thread 0 holds the new token `tok` (threads >= 256 only take part
// in the barriers: the sampler's workgroup has 1024):
// the sampler-loop bookkeeping of api_cache.py:179-181 (append, EOS stop) and the NEXT step's embedding
// x[b] = tok_emb[fed] + pos_emb[pos] (k-tiled) with its LayerNorm statistics (two equal half-row partials).
// st_* = the row's state as loaded by thread 0 at kernel start.  sh: >= 6 floats of shared scratch.
__device__ __forceinline__ void advance_embed_row(int b, int tok, const mgea::TailArgs& t, int32_t* sampled, int st_step, int st_fed,
                                                  int st_len, int st_done, float* sh) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* shi = reinterpret_cast<int*>(sh);
    if (tid == 0) {
        const mgea::StepState& s = t.s;
        sampled[b] = tok;
        int out = -1, fed = st_fed, len = st_len;
        if (!st_done) {
            out = tok;
            fed = tok;
            s.cur_ids[b] = tok;
            len += 1;
            s.ctx_len[b] = len;
            if (tok == s.eos()) {
                s.done[b] = 1;
                atomicAdd(s.n_done, 1);
            }
        }
        if (s.ids_out && st_step < s.n_steps) s.ids_out[(int64_t)b * s.n_steps + st_step] = out;
        s.row_step[b] = st_step + 1;
        shi[4] = fed < 0 ? 0 : (fed >= t.vocab ? t.vocab - 1 : fed);
        const int pos = t.absolute_pos ? len : 0;   // reference: a decode step adds pos_emb[:1] = row 0 (api_cache.py:99)
        shi[5] = pos < t.pos_rows ? pos : t.pos_rows - 1;
    }
    __syncthreads();
    const int id = shi[4], pos = shi[5];
    const int C = t.C, nf4 = C >> 2;
    float4 v[4];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + i * 256;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < nf4 && tid < 256) {
            v[i] = add4(ld4(t.tok_emb + (int64_t)id * C + f * 4), ld4(t.pos_emb + (int64_t)pos * C + f * 4));
            st4(t.x + mgea::tiled_off(b, f * 4, C), v[i]);
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    auto bsum = [&](float val) {
        val = wave_sum(val);
        __syncthreads();
        if (lane == 0 && wave < 4) sh[wave] = val;
        __syncthreads();
        return (sh[0] + sh[1]) + (sh[2] + sh[3]);
    };
    const float mean = bsum(sum) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + i * 256;
        if (f < nf4 && tid < 256) {
            const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    }
    const float m2 = bsum(q);
    if (tid == 0) st4(t.stats + (int64_t)b * 4, make_float4(mean, 0.5f * m2, mean, 0.5f * m2));   // as embed_stats_kernel
}
}  // namespace mgea
#endif
