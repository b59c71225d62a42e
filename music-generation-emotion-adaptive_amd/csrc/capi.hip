// C-ABI glue: error string, device probe, LoRA fold and the op-level entry points the parity
// tests use to check each kernel against the oracle in isolation (include/mgea.h).
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "common.h"

namespace mgea {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* get_error() { return g_err; }

// ---- switches: one table, initialised from the environment when the library is loaded ---------
namespace {
struct TuneEntry { int key; const char* name; const char* env; int dflt; };
// (each row names its key: the table is looked up by key, never by position)
const TuneEntry kTune[] = {
    {TUNE_BF16_GEMM_TILE, "bf16_gemm_tile", "MGEA_BF16_GEMM_TILE", 0},
    {TUNE_BF16_GEMM_SMALL, "bf16_gemm_small", "MGEA_BF16_GEMM_SMALL", 0},
    {TUNE_BF16_GEMM_TAIL, "bf16_gemm_tail", "MGEA_BF16_GEMM_TAIL", 2},
    {TUNE_BF16_GEMM_PHASES, "bf16_gemm_phases", "MGEA_BF16_GEMM_PHASES", 2},
    {TUNE_BF16_GEMM_REVERSE, "bf16_gemm_reverse", "MGEA_BF16_GEMM_REVERSE", 1},
    {TUNE_BERT_BF16_NOFOLD, "bert_bf16_nofold", "MGEA_BERT_BF16_NOFOLD", 0},
    {TUNE_DECODER_PREFILL_FULL, "decoder_prefill_full", "MGEA_DECODER_PREFILL_FULL", 0},
    {TUNE_BERT_FULL_LAST_LAYER, "bert_full_last_layer", "MGEA_BERT_FULL_LAST_LAYER", 0},
    {TUNE_DECODER_UNFUSED, "decoder_unfused", "MGEA_DECODER_UNFUSED", 0},
    {TUNE_DECODER_NOGEMV, "decoder_nogemv", "MGEA_DECODER_NOGEMV", 0},
    {TUNE_DECODER_NOGRAPH, "decoder_nograph", "MGEA_DECODER_NOGRAPH", 0},
    {TUNE_ATTN16_WIDE, "attn16_wide", "MGEA_ATTN16_WIDE", 1},
    {TUNE_DECODER_PREFILL16_PAGES, "decoder_prefill16_pages", "MGEA_DECODER_PREFILL16_PAGES", 1},
    {TUNE_DECODER_PREFILL16, "decoder_prefill16", "MGEA_DECODER_PREFILL16", 1},
    {TUNE_HEAD_BALANCED, "head_balanced", "MGEA_HEAD_BALANCED", 1},
    {TUNE_ATTN16_PIPE, "attn16_pipe", "MGEA_ATTN16_PIPE", 1},
    {TUNE_SKINNY_ONE_PER_CU, "skinny_one_per_cu", "MGEA_SKINNY_ONE_PER_CU", 0},
    {TUNE_SAMPLER_WAVE_SELECT, "sampler_wave_select", "MGEA_SAMPLER_WAVE_SELECT", 1},
    {TUNE_ATTN_SPLIT, "attn_split", "MGEA_ATTN_SPLIT", 64},
    {TUNE_DECODER_GRAPH_STEPS, "decoder_graph_steps", "MGEA_DECODER_GRAPH_STEPS", 8},
    {TUNE_ATTN_ARITH_PAGES, "attn_arith_pages", "MGEA_ATTN_ARITH_PAGES", 1},
};
static_assert(sizeof(kTune) / sizeof(kTune[0]) == TUNE_COUNT, "one table row per switch");
std::atomic<int> g_tune[TUNE_COUNT];
struct TuneInit {
    TuneInit() {
        for (const TuneEntry& t : kTune) {
            const char* e = getenv(t.env);
            g_tune[t.key].store(e && e[0] ? atoi(e) : t.dflt, std::memory_order_relaxed);
        }
    }
} g_tune_init;
int tune_index(const char* name) {
    for (const TuneEntry& t : kTune)
        if (name && !strcmp(name, t.name)) return t.key;
    return -1;
}
}  // namespace
int tune(int key) { return g_tune[key].load(std::memory_order_relaxed); }

int device_info(DeviceInfo* out) {
    static std::mutex mu;
    static int n_cu[64] = {0};
    int dev = 0;
    MGEA_CHECK_HIP(hipGetDevice(&dev));
    MGEA_REQUIRE(dev >= 0 && dev < 64, MGEA_EINVAL, "device id %d out of range", dev);
    std::lock_guard<std::mutex> lk(mu);
    if (!n_cu[dev]) {
        hipDeviceProp_t prop;
        MGEA_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
        n_cu[dev] = prop.multiProcessorCount;
    }
    out->dev = dev;
    out->n_cu = n_cu[dev];
    return MGEA_OK;
}
int set_max_dynamic_lds(const void* fn, int bytes, int dev, uint64_t* done) {
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (*done >> dev & 1) return MGEA_OK;
    MGEA_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    *done |= (uint64_t)1 << dev;
    return MGEA_OK;
}
}  // namespace mgea

using namespace mgea;

extern "C" {

const char* mgea_last_error(void) { return get_error(); }
int mgea_version(void) { return 100; }

int mgea_tune_set(const char* name, int32_t value) {
    const int i = tune_index(name);
    MGEA_REQUIRE(i >= 0, MGEA_EINVAL, "tune_set: unknown switch '%s'", name ? name : "(null)");
    g_tune[i].store(value, std::memory_order_relaxed);
    return MGEA_OK;
}
int mgea_tune_get(const char* name, int32_t* value_out) {
    const int i = tune_index(name);
    MGEA_REQUIRE(i >= 0 && value_out, MGEA_EINVAL, "tune_get: unknown switch '%s'", name ? name : "(null)");
    *value_out = tune(i);
    return MGEA_OK;
}

int mgea_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        set_error("hipGetDeviceCount failed: no HIP device visible");
        return MGEA_ENODEVICE;
    }
    return n;
}

int mgea_lora_merge(float* w_dev, const float* a_dev, const float* b_dev, int32_t out_dim, int32_t in_dim, int32_t r,
                    float scale, void* stream) {
    MGEA_REQUIRE(w_dev && a_dev && b_dev && out_dim > 0 && in_dim > 0 && r > 0, MGEA_EINVAL, "lora_merge: bad argument");
    return launch_lora_merge(w_dev, a_dev, b_dev, out_dim, in_dim, r, scale, (hipStream_t)stream);
}

int64_t mgea_op_gemm_workspace_floats(int32_t M, int32_t N, int32_t split_k) {
    if (split_k < 1) split_k = 1;
    return (int64_t)split_k * slab_floats(M, N);
}

int mgea_op_gemm_f32(const float* a_dev, const float* w_dev, const float* bias_dev, float* out_dev, int32_t M,
                     int32_t N, int32_t K, int32_t split_k, float* workspace_dev, void* stream) {
    MGEA_REQUIRE(a_dev && w_dev && out_dev && workspace_dev, MGEA_EINVAL, "op_gemm: NULL argument");
    hipStream_t st = (hipStream_t)stream;
    if (split_k <= 0) split_k = M > 64 ? 1 : pick_split_k(M, N, K);   // the caller sized the workspace for this (mgea.h)
    const int S = launch_gemm_f32(a_dev, K, w_dev, K, workspace_dev, M, N, K, split_k, st);
    if (S < 0) return S;
    return launch_bias_act(workspace_dev, S, slab_floats(M, N), (int)slab_ld(N), bias_dev, out_dev, N, M, N, ACT_NONE, st);
}

int mgea_op_layernorm(const float* x_dev, const float* w_dev, const float* b_dev, float* y_dev, int32_t M, int32_t C,
                      float eps, void* stream) {
    MGEA_REQUIRE(x_dev && w_dev && b_dev && y_dev, MGEA_EINVAL, "op_layernorm: NULL argument");
    return launch_layernorm(x_dev, w_dev, b_dev, y_dev, M, C, eps, (hipStream_t)stream);
}

int mgea_op_attention_f32(const float* qkv_dev, const int32_t* lens_dev, const int32_t* mask_dev, float* out_dev,
                          int32_t B, int32_t T, int32_t n_head, int32_t head_dim, void* stream) {
    MGEA_REQUIRE(qkv_dev && out_dev, MGEA_EINVAL, "op_attention: NULL argument");
    return launch_attn_dense(qkv_dev, lens_dev, mask_dev, out_dev, B, T, n_head, head_dim, 0, (hipStream_t)stream);
}

int mgea_op_f32_to_bf16(const float* src_dev, void* dst_dev, int64_t n, void* stream) {
    MGEA_REQUIRE(src_dev && dst_dev && n > 0, MGEA_EINVAL, "op_f32_to_bf16: bad argument");
    return launch_f32_to_bf16(src_dev, dst_dev, n, (hipStream_t)stream);
}

int mgea_op_gemm_bf16(const void* a_dev, const void* w_dev, const float* bias_dev, const void* res_dev, void* out_dev,
                      int32_t M, int32_t N, int32_t K, int32_t epi, void* stream) {
    MGEA_REQUIRE(a_dev && w_dev && out_dev, MGEA_EINVAL, "op_gemm_bf16: NULL argument");
    return launch_gemm_bf16(a_dev, K, w_dev, K, bias_dev, res_dev, out_dev, N, M, N, K, epi, (hipStream_t)stream);
}

int mgea_op_gemm_bf16_ln(const void* a_dev, const void* w_dev, const float* bias_dev, const void* res_dev, void* out_dev,
                         int32_t M, int32_t N, int32_t K, int32_t epi, const float* rowstat_dev, const float* c1_dev,
                         const float* ln_g_dev, const float* ln_b_dev, float* stats_out_dev, int32_t* info_out, void* stream) {
    MGEA_REQUIRE(a_dev && w_dev && out_dev, MGEA_EINVAL, "op_gemm_bf16_ln: NULL argument");
    const BfEpiLn ln{rowstat_dev, c1_dev, ln_g_dev, ln_b_dev, stats_out_dev};
    GemmBf16Info gi{-1, 0, 0};
    const int rc = launch_gemm_bf16(a_dev, K, w_dev, K, bias_dev, res_dev, out_dev, N, M, N, K, epi, (hipStream_t)stream, &gi,
                                    epi >= 3 ? &ln : nullptr);
    if (info_out) { info_out[0] = gi.kernel; info_out[1] = gi.half_tiles; }
    return rc;
}

int mgea_op_ln_rowstat(const float* part_dev, float* rowstat_dev, int32_t M, int32_t n_part, int32_t C, float eps, void* stream) {
    MGEA_REQUIRE(part_dev && rowstat_dev && M > 0 && n_part > 0 && C > 0 && C % n_part == 0, MGEA_EINVAL, "op_ln_rowstat: bad argument");
    return launch_ln_rowstat(part_dev, rowstat_dev, M, n_part, C, eps, (hipStream_t)stream);
}

int mgea_op_fold_ln_bf16(const float* w_dev, const float* gamma_dev, const float* beta_dev, const float* bias_dev, int32_t N, int32_t K,
                         void* wf_out_dev, float* c1_out_dev, float* c2_out_dev, void* stream) {
    MGEA_REQUIRE(w_dev && gamma_dev && beta_dev && bias_dev && wf_out_dev && c1_out_dev && c2_out_dev && N > 0 && K > 0, MGEA_EINVAL,
                 "op_fold_ln_bf16: bad argument");
    return launch_fold_ln_weights_bf16(w_dev, gamma_dev, beta_dev, bias_dev, wf_out_dev, c1_out_dev, c2_out_dev, N, K, (hipStream_t)stream);
}

int mgea_op_attention_bf16(const void* qkv_dev, const int32_t* mask_dev, void* out_dev, int32_t B, int32_t T,
                           int32_t n_head, int32_t head_dim, void* stream) {
    MGEA_REQUIRE(qkv_dev && out_dev, MGEA_EINVAL, "op_attention_bf16: NULL argument");
    return launch_attn_bf16(qkv_dev, mask_dev, out_dev, B, T, n_head, head_dim, (hipStream_t)stream);
}

int mgea_op_layernorm_bf16(const void* x_dev, const float* w_dev, const float* b_dev, void* y_dev, int32_t M, int32_t C,
                           float eps, void* stream) {
    MGEA_REQUIRE(x_dev && w_dev && b_dev && y_dev, MGEA_EINVAL, "op_layernorm_bf16: NULL argument");
    return launch_layernorm_bf16(x_dev, w_dev, b_dev, y_dev, M, C, eps, (hipStream_t)stream);
}

/* ablation / micro-benchmark hook for the fused skinny GEMM (tools/skinny_bench.py): EPI_ACT or
 * EPI_RES on caller buffers; dbg bits skip A loads (1), W loads (2), MFMAs (4). */
int64_t mgea_op_tiled_weight_floats(int32_t N, int32_t K) { return wtile_floats(N, K); }

int mgea_op_tile_weights(const float* w_dev, int32_t N, int32_t K, float* out_dev, void* stream) {
    return launch_tile_weights(w_dev, N, K, out_dev, (hipStream_t)stream);
}

int mgea_op_fold_ln(const float* w_dev, const float* gamma_dev, const float* beta_dev, const float* bias_dev, int32_t N, int32_t K,
                    float* wt_out_dev, float* c1_out_dev, float* c2_out_dev, void* stream) {
    return launch_ln_fold(w_dev, gamma_dev, beta_dev, bias_dev, N, K, wt_out_dev, c1_out_dev, c2_out_dev, (hipStream_t)stream);
}

int mgea_op_tile_rows(const float* src_dev, float* dst_dev, int32_t M, int32_t N, int32_t to_tiled, void* stream) {
    return launch_tile_rows(src_dev, dst_dev, M, N, to_tiled, (hipStream_t)stream);
}

int mgea_op_skinny(int32_t epi, const float* a_dev, const float* w_dev, const float* bias_dev, const float* ln_c1_dev,
                   const float* stats_in_dev, int32_t n_part, int32_t part_cnt, float* out_dev,
                   float* stats_out_dev, int32_t M, int32_t N, int32_t K, int32_t act, int32_t dbg, void* stream) {
    SkinnyArgs a{};
    a.A = a_dev; a.lda = K; a.W = w_dev; a.bias = bias_dev; a.M = M; a.N = N; a.K = K;
    a.ln_c1 = ln_c1_dev; a.eps = 1e-5f; a.stats_in = stats_in_dev; a.n_part = n_part; a.part_cnt = part_cnt;
    a.out = out_dev; a.ldo = N; a.stats_out = stats_out_dev; a.act = act; a.dbg = dbg;
    MGEA_REQUIRE(epi == EPI_ACT || epi == EPI_RES || epi == EPI_LOGITS, MGEA_EINVAL, "op_skinny: epilogue %d not exposed", epi);
    if (epi == EPI_LOGITS) {   // LM head: logits [M,N] row-major in out_dev (or NULL); per-tile (max, argmax) partials in stats_out_dev
        MGEA_REQUIRE(stats_out_dev, MGEA_EINVAL, "op_skinny: the LOGITS epilogue writes its partials to stats_out_dev");
        // (tools that pass ablation / timestamp bits in dbg run the generic kernel, whose partial count differs from the balanced head kernel's)
        const int tiles = (dbg & ~(1 << 21)) ? skinny_logits_tiles(M, N, 0) : skinny_logits_tiles(M, N, K);
        a.pmax_val = stats_out_dev;
        a.pmax_idx = reinterpret_cast<int32_t*>(stats_out_dev + (int64_t)64 * tiles);
    }
    return launch_skinny(epi, a, (hipStream_t)stream);
}

int mgea_op_skinny_logits_partials(int32_t M, int32_t N, int32_t K) { return skinny_logits_tiles(M, N, K); }

int mgea_op_sample(const float* logits_dev, int32_t B, int32_t V, const mgea_sampler_config* s, int64_t step,
                   int32_t* ids_out_dev, float* probs_out_dev, void* stream) {
    MGEA_REQUIRE(logits_dev && s, MGEA_EINVAL, "op_sample: NULL argument");
    hipStream_t st = (hipStream_t)stream;
    if (s->top_k == 1 && ids_out_dev) {
        MGEA_TRY(launch_logits_argmax(logits_dev, 1, 0, V, nullptr, nullptr, B, V, ids_out_dev, st));
        if (!probs_out_dev) return MGEA_OK;
        return launch_sample(logits_dev, B, V, *s, nullptr, nullptr, step, nullptr, probs_out_dev, st);
    }
    return launch_sample(logits_dev, B, V, *s, nullptr, nullptr, step, ids_out_dev, probs_out_dev, st);
}

}  // extern "C"
