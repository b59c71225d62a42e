// Single-query attention over the paged KV cache: the HBM-bound core of the decode step.
// Replaces nn.MultiheadAttention(q = ln1(x_new), k = v = all cached ln1(x)) with NO mask
// (api_cache.py:62-68) -- but reads *projected* K/V from pages instead of re-projecting the
// whole past every step (identical numbers).
//
// Mapping (gfx950): one 256-thread workgroup per (row b, head h, query t); its 4 waves take
// pages w, w+4, ... (one page = 64 tokens = one wave tile) and are merged through LDS with the
// usual (max, sum, acc) rescale.  B*H = 512 workgroups at the benchmark shape -> 8 waves per CU.
//   QK^T : lane = token.  The K page is stored [dh/4][64 tokens][4], so each of the dh/4 loads
//          of a wave is one fully coalesced 1 KiB global_load_dwordx4; q is wave-uniform (scalar
//          loads -> SGPR operands of the FMAs); no cross-lane reduction for the dot product.
//   softmax: online, one wave-wide max per 64-token tile (shuffle reduction), per-lane partial
//          sums reduced once at the end.
//   PV   : lane = (token group g, 16-byte d-chunk c).  V page is [64 tokens][dh]: a wave
//          instruction reads 64/(dh/4) whole rows (1 KiB contiguous); p is fetched from the lane
//          that owns the token with one bpermute per load.
// Algorithmic bytes per (b, h): 2 * ctx * dh * 4 (K and V each streamed once).
#include "common.h"

namespace mgea {

template <int DH>
__global__ __launch_bounds__(256) void attn_paged_kernel(const float* __restrict__ qkv, KvPool pool, int layer,
                                                        const int32_t* __restrict__ page_table, int max_pages,
                                                        const int32_t* __restrict__ ctx_len,
                                                        const int32_t* __restrict__ lens, float* __restrict__ out,
                                                        int H, int T, int C, float scale, int tiled_out) {
    constexpr int NCH = DH / 4;        // 16-byte chunks per head row
    constexpr int TPI = 64 / NCH;      // tokens per V wave-instruction (head_dim 96: 2 tokens x 24 chunks, 16 lanes idle)
    constexpr int NVI = 64 / TPI;      // V wave-instructions per page
    constexpr bool POW2 = (NCH & (NCH - 1)) == 0;
    static_assert(NCH % 2 == 0 && NCH <= 64 && 64 % TPI == 0, "head_dim must be a multiple of 8, at most 256, with 64 % (64 / (dh/4)) == 0");
    __shared__ float s_m[4], s_l[4];
    __shared__ float s_acc[4][DH];

    const int bh = blockIdx.x;
    const int b = bh / H, h = bh % H;
    const int t = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t m = (int64_t)b * T + t;
    const int64_t pf = pool.page_floats();
    const float* lbase = pool.base + layer * pool.layer_stride;
    const int g = lane / NCH, c = lane % NCH;
    const bool v_lane = g < TPI;       // lanes past TPI * NCH (head_dim 96) take no part in PV

    // Software pipeline over this wave's pages: while QK^T consumes K(page) the V(page) loads are in
    // flight, and while PV consumes V(page) the K(next page) loads are -- each wave keeps 16 KiB
    // (one operand tile) streaming at all times.  K/V are read once per step: non-temporal loads.
    float4 kk[NCH], vv[NVI];
    auto page_base = [&](int pg, int isv) {
        const int phys = page_table[b * max_pages + pg];
        return lbase + ((int64_t)(phys * 2 + isv) * H + h) * pf;
    };
    // n_tok = tokens of the page that are in the cache: lanes (= tokens) past the end of the last, partial page
    // re-read the last cached token instead of their own slot -- a wave instruction is 1 KiB of consecutive bytes,
    // so the tail of the page is never fetched from HBM (whole-page reads cost 8 % extra traffic on the benchmark
    // run).  Clamped addresses rather than predication: a load inside a branch makes the compiler drain vmcnt to 0
    // at the next use and the K/V pipelining below is lost.
    auto load_k = [&](int pg, int n_tok) {
        const float* kpage = page_base(pg, 0);
        const int tok = lane < n_tok ? lane : n_tok - 1;
#pragma unroll
        for (int i = 0; i < NCH; ++i) kk[i] = ldnt4(kpage + (i * 64 + tok) * 4);
    };
    // The first K tile is requested before anything else is known: its page index depends only on the
    // wave id, so the page-table -> K round trips overlap the ctx_len / q fetches (the kernel's fixed
    // latency is what short contexts pay).  A wave beyond the row's pages reads a reserved (zeroed or
    // stale but mapped) page and drops it.
    load_k(wave < max_pages ? wave : max_pages - 1, 64);   // length not known yet: whole page

    const int n_new = lens ? lens[b] : T;
    auto optr = [&](int d) { return tiled_out ? out + tiled_off((int)m, h * DH + d, C) : out + m * C + h * DH + d; };
    if (t >= n_new) {  // padded query row: defined output, never used
        if (threadIdx.x < DH) *optr(threadIdx.x) = 0.f;
        return;
    }
    const int len = ctx_len[b] + n_new;  // tokens visible to this query (whole cache, no mask)
    const int npages = (len + 63) >> 6;

    const float* qp = qkv + m * 3 * C + h * DH;
    float q[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) q[d] = qp[d] * scale;  // 1/sqrt(64) etc.: power-of-two scales are exact

    float mx = -INFINITY, lsum = 0.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int pg = wave; pg < npages; pg += 4) {
        const float* vpage = page_base(pg, 1);
        const int n_tok = len - pg * 64;   // >= 1; > 64 for a full page
#pragma unroll
        for (int j = 0; j < NVI; ++j) {   // rows beyond the cache end: the last cached row instead (their p is 0)
            const int row = v_lane ? j * TPI + g : 0;
            vv[j] = ldnt4(vpage + (row < n_tok ? row : n_tok - 1) * DH + c * 4);
        }

        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; i += 2) {
            s0 = fmaf(q[4 * i + 0], kk[i].x, s0);
            s0 = fmaf(q[4 * i + 1], kk[i].y, s0);
            s0 = fmaf(q[4 * i + 2], kk[i].z, s0);
            s0 = fmaf(q[4 * i + 3], kk[i].w, s0);
            s1 = fmaf(q[4 * i + 4], kk[i + 1].x, s1);
            s1 = fmaf(q[4 * i + 5], kk[i + 1].y, s1);
            s1 = fmaf(q[4 * i + 6], kk[i + 1].z, s1);
            s1 = fmaf(q[4 * i + 7], kk[i + 1].w, s1);
        }
        const bool valid = (pg * 64 + lane) < len;
        const float s = valid ? (s0 + s1) : -INFINITY;
        const float tmax = wave_max(s);            // finite: every visited page holds >= 1 valid token
        const float mnew = fmaxf(mx, tmax);
        const float alpha = __expf(mx - mnew);     // first tile: exp(-inf) = 0
        const float p = valid ? __expf(s - mnew) : 0.f;
        lsum = lsum * alpha + p;
        acc.x *= alpha; acc.y *= alpha; acc.z *= alpha; acc.w *= alpha;
        mx = mnew;
        if (pg + 4 < npages) load_k(pg + 4, len - (pg + 4) * 64);  // K registers are free again: next page's K under PV
#pragma unroll
        for (int j = 0; j < NVI; ++j) {
            const float pv = __shfl(p, j * TPI + g, 64);
            const float pj = v_lane ? pv : 0.f;
            acc.x = fmaf(pj, vv[j].x, acc.x);
            acc.y = fmaf(pj, vv[j].y, acc.y);
            acc.z = fmaf(pj, vv[j].z, acc.z);
            acc.w = fmaf(pj, vv[j].w, acc.w);
        }
    }
    // reduce the token groups g (lanes c, c+NCH, ...) and the per-lane softmax sums
    if (POW2) {
#pragma unroll
        for (int o = 32; o >= NCH; o >>= 1) {
            acc.x += __shfl_xor(acc.x, o, 64);
            acc.y += __shfl_xor(acc.y, o, 64);
            acc.z += __shfl_xor(acc.z, o, 64);
            acc.w += __shfl_xor(acc.w, o, 64);
        }
    } else {   // lanes c < NCH collect the other token groups in a fixed order
        float4 tot = acc;
#pragma unroll
        for (int gg = 1; gg < TPI; ++gg) {
            tot.x += __shfl(acc.x, lane + gg * NCH, 64);
            tot.y += __shfl(acc.y, lane + gg * NCH, 64);
            tot.z += __shfl(acc.z, lane + gg * NCH, 64);
            tot.w += __shfl(acc.w, lane + gg * NCH, 64);
        }
        acc = tot;
    }
    lsum = wave_sum(lsum);
    if (lane == 0) { s_m[wave] = mx; s_l[wave] = lsum; }
    if (lane < NCH) {
        s_acc[wave][4 * lane + 0] = acc.x;
        s_acc[wave][4 * lane + 1] = acc.y;
        s_acc[wave][4 * lane + 2] = acc.z;
        s_acc[wave][4 * lane + 3] = acc.w;
    }
    __syncthreads();
    if (threadIdx.x < DH) {
        const float M = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float f = (s_m[w] == -INFINITY) ? 0.f : __expf(s_m[w] - M);  // waves without pages
            num = fmaf(f, s_acc[w][threadIdx.x], num);
            den = fmaf(f, s_l[w], den);
        }
        *optr(threadIdx.x) = num / den;
    }
}

int launch_attn_paged(const float* qkv, const KvPool& pool, int layer, const int32_t* page_table, int max_pages,
                      const int32_t* ctx_len, const int32_t* lens, float* out, int B, int T, int C, int tiled_out,
                      hipStream_t st) {
    const int H = pool.H, dh = pool.dh;
    MGEA_REQUIRE(H * dh == C, MGEA_EINVAL, "attention: n_head*head_dim != d_model");
    MGEA_REQUIRE(T <= 65535, MGEA_EINVAL, "attention: too many new tokens per row (%d)", T);
    const float scale = 1.0f / sqrtf((float)dh);
    dim3 grid(B * H, T);
    switch (dh) {
        case 32: hipLaunchKernelGGL(attn_paged_kernel<32>, grid, dim3(256), 0, st, qkv, pool, layer, page_table, max_pages, ctx_len, lens, out, H, T, C, scale, tiled_out); break;
        case 64: hipLaunchKernelGGL(attn_paged_kernel<64>, grid, dim3(256), 0, st, qkv, pool, layer, page_table, max_pages, ctx_len, lens, out, H, T, C, scale, tiled_out); break;
        case 96: hipLaunchKernelGGL(attn_paged_kernel<96>, grid, dim3(256), 0, st, qkv, pool, layer, page_table, max_pages, ctx_len, lens, out, H, T, C, scale, tiled_out); break;
        case 128: hipLaunchKernelGGL(attn_paged_kernel<128>, grid, dim3(256), 0, st, qkv, pool, layer, page_table, max_pages, ctx_len, lens, out, H, T, C, scale, tiled_out); break;
        default:
            MGEA_REQUIRE(false, MGEA_EINVAL, "attention: head_dim %d not supported (32, 64, 96, 128)", dh);
    }
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

}  // namespace mgea
