// Decode-step projections for very small batches (M <= 2 rows, i.e. the reference's own serving case B = 1,
// api_cache.py:159-184): out[m, n] = f(LN?(x[m]) . W[n] + bias[n]) as wave-level dot products.
//
// The MFMA kernel of gemm_skinny.hip needs 16 x 16 output tiles, so at M <= 16 a GEMM with N = 512 is only 32
// workgroups, each streaming 128-256 KB of weights through one CU (~5 us per launch whatever the batch).  Here a wave
// owns CW output columns and walks the whole K with consecutive lanes on consecutive 16 bytes of the ROW-MAJOR arena
// weights (no tiled copy needed): 128-520 workgroups, a few KB each.  LayerNorm is applied directly (two-pass
// statistics of the row in registers, then (x - mean) * rstd * gamma + beta on the fly): the reference's own operation
// order (api_cache.py:60-62, 73).  Activations stay in the k-tiled layout (common.h) so that the attention kernel and
// the tail kernels are shared with the batched path.  Deterministic: fixed lane-reduction tree, no atomics.
#include "common.h"

namespace mgea {

constexpr int GEMV_MR = 4;   // rows

__device__ __forceinline__ float wave_sum_fast(float v) {   // same total in every lane
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// grid.x = ceil(N / (4 * CW)); 256 threads; wave w of workgroup g owns columns (4 g + w) * CW .. + CW - 1.  MR = rows
// computed (1, 2 or 4 >= M).  LN kernels have K = d_model <= 1024: x, gamma, beta and the wave's W rows are all
// requested up front (one memory round trip), the statistics and the normalisation then run on registers.
template <int EPI, bool LN, int CW, int MR>
__global__ __launch_bounds__(256) void gemv_rows_kernel(SkinnyArgs a) {
    __shared__ float s_best[4][GEMV_MR];
    __shared__ int s_bidx[4][GEMV_MR];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = (blockIdx.x * 4 + wave) * CW;
    const int KS = a.K >> 8;   // 256 floats of k per step: one float4 per lane

    // this lane's output element (lane l < CW * MR: column j = l / MR, row m = l % MR) and everything its epilogue
    // reads from memory, requested before the dot products: the kernel is one dependent chain of memory round trips
    const int ej = lane / MR, m = lane % MR, n = n0 + ej;
    const bool on = lane < CW * MR && m < a.M && n < a.N;
    float e_bias = 0.f, e_res = 0.f;
    int e_pos = 0, e_phys = -1;
    if (on) {
        if (a.bias) e_bias = a.bias[n];
        if (EPI == EPI_RES) e_res = a.out[tiled_off(m, n, a.N)];
        if (EPI == EPI_QKV && n >= a.C) {
            e_pos = a.ctx_len[m];
            if ((e_pos >> 6) < a.max_pages) e_phys = a.page_table[m * a.max_pages + (e_pos >> 6)];
        }
    }

    float acc[CW][MR];
#pragma unroll
    for (int j = 0; j < CW; ++j)
#pragma unroll
        for (int m = 0; m < MR; ++m) acc[j][m] = 0.f;
    const float* wrow[CW];
#pragma unroll
    for (int j = 0; j < CW; ++j) {
        const int n = n0 + j < a.N ? n0 + j : a.N - 1;
        wrow[j] = a.W + (int64_t)n * a.K + lane * 4;
    }
    auto dot = [&](const float4 (&x)[MR], const float4 (&w)[CW]) {
#pragma unroll
        for (int j = 0; j < CW; ++j)
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                acc[j][m] = fmaf(x[m].x, w[j].x, acc[j][m]);
                acc[j][m] = fmaf(x[m].y, w[j].y, acc[j][m]);
                acc[j][m] = fmaf(x[m].z, w[j].z, acc[j][m]);
                acc[j][m] = fmaf(x[m].w, w[j].w, acc[j][m]);
            }
    };
    if (LN) {
        constexpr int KSX = 4;   // K <= 1024 (launcher check)
        float4 xr[KSX][MR], wr[KSX][CW], g[KSX], b[KSX];
#pragma unroll
        for (int st = 0; st < KSX; ++st) {
            const int k = (st < KS ? st : 0) * 256 + lane * 4;   // steps beyond K re-read step 0 and are not used
#pragma unroll
            for (int m = 0; m < MR; ++m) xr[st][m] = ld4(a.A + tiled_off(m < a.M ? m : 0, k, a.K));
#pragma unroll
            for (int j = 0; j < CW; ++j) wr[st][j] = ld4(wrow[j] + (st < KS ? st : 0) * 256);
            g[st] = ld4(a.ln_g + k);
            b[st] = ld4(a.ln_b + k);
        }
        // two-pass statistics of each row on the registers (F.layer_norm's biased variance, api_cache.py:60)
        float mu[MR], rs[MR];
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            float s = 0.f;
#pragma unroll
            for (int st = 0; st < KSX; ++st)
                if (st < KS) s += (xr[st][m].x + xr[st][m].y) + (xr[st][m].z + xr[st][m].w);
            mu[m] = s;
        }
#pragma unroll
        for (int m = 0; m < MR; ++m) mu[m] = wave_sum_fast(mu[m]) / (float)a.K;
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            float q = 0.f;
#pragma unroll
            for (int st = 0; st < KSX; ++st)
                if (st < KS) {
                    const float d0 = xr[st][m].x - mu[m], d1 = xr[st][m].y - mu[m], d2 = xr[st][m].z - mu[m], d3 = xr[st][m].w - mu[m];
                    q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                }
            rs[m] = q;
        }
#pragma unroll
        for (int m = 0; m < MR; ++m) rs[m] = 1.0f / sqrtf(wave_sum_fast(rs[m]) / (float)a.K + a.eps);
#pragma unroll
        for (int st = 0; st < KSX; ++st) {
            if (st < KS) {
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    xr[st][m].x = (xr[st][m].x - mu[m]) * rs[m] * g[st].x + b[st].x;
                    xr[st][m].y = (xr[st][m].y - mu[m]) * rs[m] * g[st].y + b[st].y;
                    xr[st][m].z = (xr[st][m].z - mu[m]) * rs[m] * g[st].z + b[st].z;
                    xr[st][m].w = (xr[st][m].w - mu[m]) * rs[m] * g[st].w + b[st].w;
                }
                dot(xr[st], wr[st]);
            }
        }
    } else {
        // K up to 4096: four steps of loads in flight at a time
        for (int st0 = 0; st0 < KS; st0 += 4) {
            float4 x[4][MR], w[4][CW];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int st = st0 + i < KS ? st0 + i : st0;
#pragma unroll
                for (int m = 0; m < MR; ++m) x[i][m] = ld4(a.A + tiled_off(m < a.M ? m : 0, st * 256 + lane * 4, a.K));
#pragma unroll
                for (int j = 0; j < CW; ++j) w[i][j] = ld4(wrow[j] + st * 256);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (st0 + i < KS) dot(x[i], w[i]);
        }
    }
#pragma unroll
    for (int j = 0; j < CW; ++j)
#pragma unroll
        for (int m = 0; m < MR; ++m) acc[j][m] = wave_sum_fast(acc[j][m]);

    // epilogue: lane l < CW * MR finishes output (column j = l / MR, row m = l % MR)
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < CW; ++j)
#pragma unroll
        for (int m = 0; m < MR; ++m)
            if (lane == j * MR + m) v = acc[j][m];
    v += e_bias;
    if (EPI == EPI_ACT) {
        if (a.act == ACT_GELU) v = gelu_erf(v);
        if (a.act == ACT_RELU) v = fmaxf(v, 0.f);
        if (on) a.out[tiled_off(m, n, a.N)] = v;
    }
    if (EPI == EPI_RES) {
        if (on) a.out[tiled_off(m, n, a.N)] = e_res + v;
    }
    if (EPI == EPI_QKV) {   // decode step only: one new token per row (T = 1, no ragged lengths)
        if (on) {
            a.out[(int64_t)m * a.ldo + n] = v;
            if (n >= a.C) {
                const int slot = e_pos & 63, phys = e_phys;
                if (phys >= 0) {
                    const int64_t pf = a.pool.page_elems();   // fp32 pages only (the launcher refuses an fp16 pool)
                    const int isv = n >= 2 * a.C;
                    const int nn = n - (isv ? 2 * a.C : a.C);
                    const int hh = nn / a.pool.dh, d = nn % a.pool.dh;
                    float* page_p = static_cast<float*>(a.pool.base) + a.layer * a.pool.layer_stride + ((int64_t)(phys * 2 + isv) * a.pool.H + hh) * pf;
                    if (isv) page_p[slot * a.pool.dh + d] = v;
                    else     page_p[((d >> 2) * MGEA_KV_PAGE_TOKENS + slot) * 4 + (d & 3)] = v;
                }
            }
        }
    }
    if (EPI == EPI_LOGITS) {
        if (on && a.out) a.out[(int64_t)m * a.ldo + n] = v;
        // per-row (max, lowest argmax) over the workgroup's 4 * CW columns -> one partial per workgroup
        float best = on ? v : -INFINITY;
        int bi = on ? n : 0x7fffffff;
#pragma unroll
        for (int o = MR; o < 64; o <<= 1) {   // lanes with the same row m are MR apart
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane < MR) { s_best[wave][lane] = best; s_bidx[wave][lane] = bi; }
        __syncthreads();
        if (threadIdx.x < MR && (int)threadIdx.x < a.M && a.pmax_val) {
            float b0 = s_best[0][threadIdx.x];
            int i0 = s_bidx[0][threadIdx.x];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float ov = s_best[w][threadIdx.x];
                const int oi = s_bidx[w][threadIdx.x];
                if (ov > b0 || (ov == b0 && oi < i0)) { b0 = ov; i0 = oi; }
            }
            a.pmax_val[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = b0;
            a.pmax_idx[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = i0;
        }
    }
}

// measured on MI355X (tools/decode_batch_sweep.py): 179 vs 205 us per step at B = 1, 195 vs 205 at B = 2, 235 vs 208 at B = 4
// -> the MFMA path takes over from 3 rows
bool gemv_shape_ok(int M, int N, int K) { return M >= 1 && M <= 2 && K % 256 == 0 && K <= 4096 && N >= 4; }

// columns per wave: enough workgroups for the chip, not more than ~2 per CU
static int gemv_cw(int epi, int N) {
    if (epi == EPI_LOGITS) return 4;   // 16 columns per workgroup = the partial count of skinny_logits_tiles (M <= 32)
    return N >= 2048 ? 2 : 1;
}

template <int EPI, int CW, int MR>
static int launch_gemv_m(const SkinnyArgs& a, hipStream_t st) {
    const dim3 grid(ceil_div(a.N, 4 * CW)), block(256);
    if (a.ln_g) hipLaunchKernelGGL((gemv_rows_kernel<EPI, true, CW, MR>), grid, block, 0, st, a);
    else        hipLaunchKernelGGL((gemv_rows_kernel<EPI, false, CW, MR>), grid, block, 0, st, a);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

template <int EPI, int CW>
static int launch_gemv_t(const SkinnyArgs& a, hipStream_t st) {
    if (a.M == 1) return launch_gemv_m<EPI, CW, 1>(a, st);
    if (a.M == 2) return launch_gemv_m<EPI, CW, 2>(a, st);
    return launch_gemv_m<EPI, CW, 4>(a, st);
}

// a.W = ROW-MAJOR [N, K] weights, a.ln_g / a.ln_b = LayerNorm gamma / beta (or NULL), a.A / a.out k-tiled as in launch_skinny
int launch_gemv(int epi, const SkinnyArgs& a, hipStream_t st) {
    MGEA_REQUIRE(gemv_shape_ok(a.M, a.N, a.K), MGEA_EINVAL, "gemv: M=%d (1..2) N=%d K=%d (multiple of 256)", a.M, a.N, a.K);
    MGEA_REQUIRE(epi == EPI_LOGITS || a.N % 8 == 0, MGEA_EINVAL, "gemv: N=%d must be a multiple of 8", a.N);
    MGEA_REQUIRE(epi != EPI_QKV || (a.T == 1 && !a.lens), MGEA_EINVAL, "gemv: the QKV epilogue handles single-token decode steps only");
    MGEA_REQUIRE(!a.w_f16 && !(epi == EPI_QKV && a.pool.f16), MGEA_EINVAL, "gemv: fp32 weights and KV pages only");
    MGEA_REQUIRE(!a.ln_g || (a.ln_b && a.K <= 1024), MGEA_EINVAL, "gemv: LayerNorm prologue needs beta and K <= 1024 (K=%d)", a.K);
    const int cw = gemv_cw(epi, a.N);
    switch (epi) {
        case EPI_QKV: return cw == 2 ? launch_gemv_t<EPI_QKV, 2>(a, st) : launch_gemv_t<EPI_QKV, 1>(a, st);
        case EPI_RES: return cw == 2 ? launch_gemv_t<EPI_RES, 2>(a, st) : launch_gemv_t<EPI_RES, 1>(a, st);
        case EPI_ACT: return cw == 2 ? launch_gemv_t<EPI_ACT, 2>(a, st) : launch_gemv_t<EPI_ACT, 1>(a, st);
        case EPI_LOGITS: return launch_gemv_t<EPI_LOGITS, 4>(a, st);
    }
    set_error("gemv: unknown epilogue %d", epi);
    return MGEA_EINVAL;
}

}  // namespace mgea
