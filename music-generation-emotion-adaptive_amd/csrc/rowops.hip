// Row-wise kernels around the GEMMs: embedding lookups, LayerNorm, and the split-K slab
// epilogues (bias / activation / residual / LayerNorm / KV-page scatter / logits + argmax).
// All HBM-bound byte movers: 16-byte accesses, one 256-thread block per row (LayerNorm-type)
// or a 2-D grid (element-wise type); slabs are summed in slab order -> deterministic.
//
// Reference semantics: LayerNorm eps 1e-5 (api_cache.py:42,44) / 1e-12 (DistilBERT), exact-erf
// GELU (nn.GELU default, api_cache.py:47), tok_emb + pos_emb[:T] (api_cache.py:99), residual adds
// (api_cache.py:72-73), post-LN order of nn.TransformerEncoderLayer (generate.py:30-32).
#include "common.h"

namespace mgea {

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();  // protect red[] from the previous use
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// Row held as up to 4 float4 per thread (C <= 4096, C % 4 == 0).
struct RowRegs {
    float4 v[4];
};

__device__ __forceinline__ void row_layernorm(RowRegs& r, int C, float eps, const float* __restrict__ w,
                                              const float* __restrict__ b, float* red) {
    const int nf4 = C >> 2;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + i * 256;
        if (f < nf4) s += (r.v[i].x + r.v[i].y) + (r.v[i].z + r.v[i].w);
    }
    const float mean = block_sum_256(s, red) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + i * 256;
        if (f < nf4) {
            const float a = r.v[i].x - mean, bb = r.v[i].y - mean, cc = r.v[i].z - mean, d = r.v[i].w - mean;
            q += (a * a + bb * bb) + (cc * cc + d * d);
        }
    }
    const float var = block_sum_256(q, red) / (float)C;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + i * 256;
        if (f < nf4) {
            const float4 ww = ld4(w + f * 4), bv = ld4(b + f * 4);
            r.v[i].x = (r.v[i].x - mean) * rstd * ww.x + bv.x;
            r.v[i].y = (r.v[i].y - mean) * rstd * ww.y + bv.y;
            r.v[i].z = (r.v[i].z - mean) * rstd * ww.z + bv.z;
            r.v[i].w = (r.v[i].w - mean) * rstd * ww.w + bv.w;
        }
    }
}

__device__ __forceinline__ float4 slab_sum4(const float* __restrict__ P, int S, int64_t ps, int64_t off) {
    float4 a = ld4(P + off);
    for (int s = 1; s < S; ++s) a = add4(a, ld4(P + s * ps + off));
    return a;
}

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ b, float* __restrict__ y, int C,
                                                       float eps) {
    __shared__ float red[4];
    const int64_t m = blockIdx.x;
    RowRegs r;
    const int nf4 = C >> 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + i * 256;
        if (f < nf4) r.v[i] = ld4(x + m * C + f * 4);
    }
    row_layernorm(r, C, eps, w, b, red);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + i * 256;
        if (f < nf4) st4(y + m * C + f * 4, r.v[i]);
    }
}

int launch_layernorm(const float* x, const float* w, const float* b, float* y, int M, int C, float eps,
                     hipStream_t st) {
    MGEA_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && C <= 4096, MGEA_EINVAL, "layernorm: bad shape M=%d C=%d", M, C);
    hipLaunchKernelGGL(layernorm_kernel, dim3(M), dim3(256), 0, st, x, w, b, y, C, eps);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_ln_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ lens,
                                                      const int32_t* __restrict__ ctx_len,
                                                      const float* __restrict__ tok_emb,
                                                      const float* __restrict__ pos_emb, float* __restrict__ x,
                                                      float* __restrict__ xn, const float* __restrict__ lnw,
                                                      const float* __restrict__ lnb, float eps, int T, int C, int vocab,
                                                      int pos_rows, int absolute_pos, int32_t* __restrict__ err_flag) {
    __shared__ float red[4];
    const int64_t m = blockIdx.x;
    const int b = (int)(m / T), t = (int)(m % T);
    const int nf4 = C >> 2;
    const bool real = lens ? (t < lens[b]) : true;
    int id = ids[m];
    if (real && (id < 0 || id >= vocab) && err_flag && threadIdx.x == 0) atomicOr(err_flag, 1);   // see embed_stats_kernel
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // clamp keeps the loads in bounds
    int pos = t + ((absolute_pos && ctx_len) ? ctx_len[b] : 0);
    pos = pos < pos_rows ? pos : pos_rows - 1;
    RowRegs r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + i * 256;
        if (f < nf4) {
            r.v[i] = real ? add4(ld4(tok_emb + (int64_t)id * C + f * 4), ld4(pos_emb + (int64_t)pos * C + f * 4))
                          : make_float4(0.f, 0.f, 0.f, 0.f);
            st4(x + m * C + f * 4, r.v[i]);
        }
    }
    if (lnw) {
        row_layernorm(r, C, eps, lnw, lnb, red);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = threadIdx.x + i * 256;
            if (f < nf4) st4(xn + m * C + f * 4, r.v[i]);
        }
    }
}

int launch_embed_ln(const int32_t* ids, const int32_t* lens, const int32_t* ctx_len, const float* tok_emb,
                    const float* pos_emb, float* x, float* xn, const float* lnw, const float* lnb, float eps,
                    int B, int T, int C, int vocab, int pos_rows, int absolute_pos, int32_t* err_flag, hipStream_t st) {
    MGEA_REQUIRE(C % 4 == 0 && C <= 4096, MGEA_EINVAL, "embed: d_model=%d must be a multiple of 4 and <= 4096", C);
    hipLaunchKernelGGL(embed_ln_kernel, dim3(B * T), dim3(256), 0, st, ids, lens, ctx_len, tok_emb, pos_emb, x, xn,
                       lnw, lnb, eps, T, C, vocab, pos_rows, absolute_pos, err_flag);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

__global__ __launch_bounds__(256) void bert_embed_ln_kernel(const int32_t* __restrict__ ids,
                                                           const float* __restrict__ word,
                                                           const float* __restrict__ pos, const float* __restrict__ lnw,
                                                           const float* __restrict__ lnb, float eps,
                                                           float* __restrict__ h, int S, int D, int vocab,
                                                           int32_t* __restrict__ err_flag, const int32_t* __restrict__ pos_ids) {
    __shared__ float red[4];
    const int64_t m = blockIdx.x;
    const int t = pos_ids ? pos_ids[m] : (int)(m % S);          // packed rows carry their position
    const int nf4 = D >> 2;
    int id = ids[m];
    // an id outside the vocabulary (nn.Embedding raises IndexError) is clamped and reported through the engine's sticky flag
    if ((id < 0 || id >= vocab) && err_flag && threadIdx.x == 0) atomicOr(err_flag, 1);
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    RowRegs r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + i * 256;
        if (f < nf4) r.v[i] = add4(ld4(word + (int64_t)id * D + f * 4), ld4(pos + (int64_t)t * D + f * 4));
    }
    row_layernorm(r, D, eps, lnw, lnb, red);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + i * 256;
        if (f < nf4) st4(h + m * D + f * 4, r.v[i]);
    }
}

int launch_bert_embed_ln(const int32_t* ids, const float* word, const float* pos, const float* lnw,
                         const float* lnb, float eps, float* h, int B, int S, int D, int vocab, hipStream_t st, int32_t* err_flag,
                         const int32_t* pos_ids) {
    MGEA_REQUIRE(D % 4 == 0 && D <= 4096, MGEA_EINVAL, "bert embed: dim=%d must be a multiple of 4 and <= 4096", D);
    hipLaunchKernelGGL(bert_embed_ln_kernel, dim3(B * S), dim3(256), 0, st, ids, word, pos, lnw, lnb, eps, h, S, D,
                       vocab, err_flag, pos_ids);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

// ------------------------------------------------------------------------------------------
// out[m, n] = act(sum_s P[s][m][n] + bias[n]);  grid (M, ceil(N/1024)), float4 per thread.
template <int ACT>
__global__ __launch_bounds__(256) void bias_act_kernel(const float* __restrict__ P, int S, int64_t ps, int ldp,
                                                      const float* __restrict__ bias, float* __restrict__ out,
                                                      int ldo, int N) {
    const int64_t m = blockIdx.x;
    const int n = (blockIdx.y * 256 + threadIdx.x) * 4;
    if (n >= N) return;
    float4 v = slab_sum4(P, S, ps, m * ldp + n);
    if (n + 3 < N) {
        if (bias) v = add4(v, ld4(bias + n));
        if (ACT == ACT_GELU) v = make_float4(gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w));
        if (ACT == ACT_RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        if ((ldo & 3) == 0) {
            st4(out + m * ldo + n, v);
        } else {
            float* o = out + m * ldo + n;
            o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
        }
    } else {  // ragged tail (N % 4 != 0)
        const float e[4] = {v.x, v.y, v.z, v.w};
        for (int j = 0; j < 4 && n + j < N; ++j) {
            float t = e[j] + (bias ? bias[n + j] : 0.f);
            if (ACT == ACT_GELU) t = gelu_erf(t);
            if (ACT == ACT_RELU) t = fmaxf(t, 0.f);
            out[m * ldo + n + j] = t;
        }
    }
}

int launch_bias_act(const float* P, int S, int64_t ps, int ldp, const float* bias, float* out, int ldo, int M,
                    int N, int act, hipStream_t st) {
    dim3 grid(M, ceil_div(N, 1024));
    if (act == ACT_GELU)
        hipLaunchKernelGGL(bias_act_kernel<ACT_GELU>, grid, dim3(256), 0, st, P, S, ps, ldp, bias, out, ldo, N);
    else if (act == ACT_RELU)
        hipLaunchKernelGGL(bias_act_kernel<ACT_RELU>, grid, dim3(256), 0, st, P, S, ps, ldp, bias, out, ldo, N);
    else
        hipLaunchKernelGGL(bias_act_kernel<ACT_NONE>, grid, dim3(256), 0, st, P, S, ps, ldp, bias, out, ldo, N);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bias_res_ln_kernel(const float* __restrict__ P, int S, int64_t ps, int ldp,
                                                         const float* __restrict__ bias, float* __restrict__ x,
                                                         float* __restrict__ xn, const float* __restrict__ lnw,
                                                         const float* __restrict__ lnb, float eps, int C,
                                                         int post_ln) {
    __shared__ float red[4];
    const int64_t m = blockIdx.x;
    const int nf4 = C >> 2;
    RowRegs r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + i * 256;
        if (f < nf4) {
            float4 v = slab_sum4(P, S, ps, m * ldp + f * 4);
            if (bias) v = add4(v, ld4(bias + f * 4));
            r.v[i] = add4(ld4(x + m * C + f * 4), v);
            if (!post_ln) st4(x + m * C + f * 4, r.v[i]);
        }
    }
    if (lnw) {
        row_layernorm(r, C, eps, lnw, lnb, red);
        float* dst = post_ln ? x : xn;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = threadIdx.x + i * 256;
            if (f < nf4) st4(dst + m * C + f * 4, r.v[i]);
        }
    }
}

int launch_bias_res_ln(const float* P, int S, int64_t ps, int ldp, const float* bias, float* x, float* xn,
                       const float* lnw, const float* lnb, float eps, int M, int C, int post_ln, hipStream_t st) {
    MGEA_REQUIRE(C % 4 == 0 && C <= 4096, MGEA_EINVAL, "bias_res_ln: bad C=%d", C);
    MGEA_REQUIRE(!post_ln || lnw, MGEA_EINVAL, "post-LN epilogue needs LayerNorm weights");
    hipLaunchKernelGGL(bias_res_ln_kernel, dim3(M), dim3(256), 0, st, P, S, ps, ldp, bias, x, xn, lnw, lnb, eps, C,
                       post_ln);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

// ------------------------------------------------------------------------------------------
// qkv epilogue + KV page scatter (page layouts: KvPool in common.h; fp32 or fp16 elements).
__global__ __launch_bounds__(256) void qkv_scatter_kernel(const float* __restrict__ P, int S, int64_t ps, int ldp,
                                                         const float* __restrict__ bias,
                                                         float* __restrict__ qkv_out, KvPool pool, int layer,
                                                         const int32_t* __restrict__ page_table, int max_pages,
                                                         const int32_t* __restrict__ ctx_len,
                                                         const int32_t* __restrict__ lens, int T, int C) {
    const int64_t m = blockIdx.x;
    const int b = (int)(m / T), t = (int)(m % T);
    const bool real = lens ? (t < lens[b]) : true;
    const int pos = ctx_len[b] + t;
    const int page = pos >> 6, slot = pos & 63;
    const bool cache_ok = real && page < max_pages;
    const int phys = cache_ok ? page_table[b * max_pages + page] : 0;
    const int nf4 = (3 * C) >> 2;
    if (!qkv_out) {   // scatter only: P is the finished qkv buffer (bias applied by the GEMM's own epilogue), K | V of the real tokens -> pages
        if (!cache_ok) return;
        for (int f = (C >> 2) + threadIdx.x; f < nf4; f += 256) {
            const int n = f * 4;
            const int isv = n >= 2 * C, nn = n - (isv ? 2 * C : C);
            kv_store4(pool, layer, phys, isv, nn / pool.dh, slot, nn % pool.dh, ld4(P + m * ldp + n));
        }
        return;
    }
    for (int f = threadIdx.x; f < nf4; f += 256) {
        const int n = f * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (real) {
            v = slab_sum4(P, S, ps, m * ldp + n);
            if (bias) v = add4(v, ld4(bias + n));
        }
        st4(qkv_out + m * 3 * C + n, v);
        if (cache_ok && n >= C) {
            const int isv = n >= 2 * C, nn = n - (isv ? 2 * C : C);
            kv_store4(pool, layer, phys, isv, nn / pool.dh, slot, nn % pool.dh, v);
        }
    }
}

int launch_qkv_scatter(const float* P, int S, int64_t ps, int ldp, const float* bias, float* qkv_out,
                       const KvPool& pool, int layer, const int32_t* page_table, int max_pages,
                       const int32_t* ctx_len, const int32_t* lens, int B, int T, int C, hipStream_t st) {
    hipLaunchKernelGGL(qkv_scatter_kernel, dim3(B * T), dim3(256), 0, st, P, S, ps, ldp, bias, qkv_out, pool, layer,
                       page_table, max_pages, ctx_len, lens, T, C);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

// ------------------------------------------------------------------------------------------
// logits row: v = sum P + bias; optional store; argmax with ties to the lowest index.
__global__ __launch_bounds__(256) void logits_argmax_kernel(const float* __restrict__ P, int S, int64_t ps, int ldp,
                                                           const float* __restrict__ bias,
                                                           float* __restrict__ logits, int V,
                                                           int32_t* __restrict__ argmax_out) {
    __shared__ float sval[4];
    __shared__ int sidx[4];
    const int64_t m = blockIdx.x;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int n = threadIdx.x; n < V; n += 256) {
        float v = P[m * ldp + n];
        for (int s = 1; s < S; ++s) v += P[s * ps + m * ldp + n];
        if (bias) v += bias[n];
        if (logits) logits[m * V + n] = v;
        if (v > best) { best = v; bi = n; }
    }
    if (!argmax_out) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sval[wave] = best; sidx[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (sval[w] > best || (sval[w] == best && sidx[w] < bi)) { best = sval[w]; bi = sidx[w]; }
        argmax_out[m] = bi == 0x7fffffff ? 0 : bi;
    }
}

int launch_logits_argmax(const float* P, int S, int64_t ps, int ldp, const float* bias, float* logits, int M, int V,
                         int32_t* argmax_out, hipStream_t st) {
    hipLaunchKernelGGL(logits_argmax_kernel, dim3(M), dim3(256), 0, st, P, S, ps, ldp, bias, logits, V, argmax_out);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

// ------------------------------------------------------------------------------------------
__global__ void advance_kernel(const int32_t* __restrict__ sampled, StepState s, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int step = s.row_step[b];
    int out = -1;
    if (!s.done[b]) {
        const int tok = sampled[b];
        out = tok;
        s.cur_ids[b] = tok;
        s.ctx_len[b] += 1;  // the token fed this step now sits in the cache
        if (tok == s.eos()) {
            s.done[b] = 1;
            atomicAdd(s.n_done, 1);
        }
    }
    if (s.ids_out && step < s.n_steps) s.ids_out[(int64_t)b * s.n_steps + step] = out;
    s.row_step[b] = step + 1;
}

int launch_advance(const int32_t* sampled, const StepState& s, int B, hipStream_t st) {
    hipLaunchKernelGGL(advance_kernel, dim3(ceil_div(B, 256)), dim3(256), 0, st, sampled, s, B);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

__global__ void add_lens_kernel(int32_t* ctx_len, const int32_t* lens, int T, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) ctx_len[b] += lens ? lens[b] : T;
}
int launch_add_lens(int32_t* ctx_len, const int32_t* lens, int T, int B, hipStream_t st) {
    hipLaunchKernelGGL(add_lens_kernel, dim3(ceil_div(B, 256)), dim3(256), 0, st, ctx_len, lens, T, B);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

__global__ void take_last_kernel(const int32_t* ids, const int32_t* lens, int32_t* cur_ids, int B, int T) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        int n = lens ? lens[b] : T;
        n = n < 1 ? 1 : (n > T ? T : n);
        cur_ids[b] = ids[(int64_t)b * T + n - 1];
    }
}
int launch_take_last(const int32_t* ids, const int32_t* lens, int32_t* cur_ids, int B, int T, hipStream_t st) {
    hipLaunchKernelGGL(take_last_kernel, dim3(ceil_div(B, 256)), dim3(256), 0, st, ids, lens, cur_ids, B, T);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

__global__ void gather_rows_kernel(const float* __restrict__ src, int ld_src, float* __restrict__ dst, int ld_dst,
                                   int row_step, int C, const int32_t* __restrict__ row_idx) {
    const int64_t r = blockIdx.x;
    const int64_t sr = row_idx ? (int64_t)row_idx[r] : r * row_step;
    for (int f = threadIdx.x; f < (C >> 2); f += blockDim.x)
        st4(dst + r * ld_dst + f * 4, ld4(src + sr * (int64_t)ld_src + f * 4));
}
int launch_gather_rows(const float* src, int ld_src, float* dst, int ld_dst, int rows, int row_step, int C,
                       hipStream_t st, const int32_t* row_idx) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3(rows), dim3(256), 0, st, src, ld_src, dst, ld_dst, row_step, C, row_idx);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

__global__ void lora_merge_kernel(float* __restrict__ w, const float* __restrict__ a, const float* __restrict__ b,
                                  int out_dim, int in_dim, int r, float scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)out_dim * in_dim) return;
    const int o = (int)(i / in_dim), k = (int)(i % in_dim);
    float acc = 0.f;
    for (int j = 0; j < r; ++j) acc = fmaf(b[o * r + j], a[j * in_dim + k], acc);
    w[i] = w[i] + scale * acc;
}
int launch_lora_merge(float* w, const float* a, const float* b, int out_dim, int in_dim, int r, float scale,
                      hipStream_t st) {
    const int64_t n = (int64_t)out_dim * in_dim;
    hipLaunchKernelGGL(lora_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, a, b, out_dim,
                       in_dim, r, scale);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

}  // namespace mgea
