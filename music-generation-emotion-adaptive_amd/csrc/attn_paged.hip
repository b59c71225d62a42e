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
// Algorithmic bytes per (b, h): 2 * ctx * dh * 4 (K and V each streamed once); half of that with fp16 pages.
//
// F16 (MGEA_DTYPE_F16 engines): the pages hold _Float16 with the same two layouts at 16-byte granularity (a group is 8
// elements instead of 4: K [dh/8][64 tokens][8], V [64 tokens][dh]), so every wave load is still 1 KiB of consecutive
// bytes and carries twice the tokens x dims.  q, the scores, the softmax and the output accumulators stay fp32; K/V
// elements are widened in the FMA (v_fma_mix_f32).
#include "common.h"

namespace mgea {

template <bool F16>
__device__ __forceinline__ float kv_elem(const f32x4& raw, int j) {
    if (F16) return (float)__builtin_bit_cast(h16x8, raw)[j];
    return raw[j];
}

// SPLIT (small batches, round 4): the pages of one (row, head, query) are spread over gridDim.z workgroups -- split z takes pages
// 4z .. 4z+3, then 4 (z + gridDim.z) ..., one per wave -- and the workgroups that found pages leave an unnormalised (max, sum, acc[dh])
// partial in `sp.part`; the one that arrives last at the (row, head, query)'s counter merges them in split order (deterministic) and
// writes the output.  Nobody waits for anybody: no spinning, nothing to hang.  Partials and counter are written and read with
// device-scope (sc1) stores, loads and atomics, so the protocol does not depend on which XCD (which L2) a workgroup ran on.
// At B = 1 the plain kernel is 8 workgroups whose waves walk up to 4 pages one memory round trip after the other (4 -> 16 us from 256 to
// 1024 tokens of context); split, every wave has one page and a launch is one round trip plus the merge whatever the context.
// (Round 4 also built the merge INTO the consumer -- the kernel leaving partials only and the single-stream out-projection GEMV merging
// them while loading its row, no counter at all: attention 6.35 -> 5.24 us and the GEMV 4.8 -> 5.3 in the kernel trace, but the B = 1 step
// 174 us against 167.5 for the last-arriver merge, measured three times interleaved; removed.)
template <int DH, bool F16, bool SPLIT>
__global__ __launch_bounds__(256) void attn_paged_kernel(const float* __restrict__ qkv, KvPool pool, int layer,
                                                        const int32_t* __restrict__ page_table, int max_pages,
                                                        const int32_t* __restrict__ ctx_len,
                                                        const int32_t* __restrict__ lens, float* __restrict__ out,
                                                        int H, int T, int C, float scale, int tiled_out, AttnSplit sp) {
    constexpr int G = F16 ? 8 : 4;     // elements per 16-byte group
    constexpr int EB = F16 ? 2 : 4;    // bytes per element
    constexpr int NCH = DH / G;        // 16-byte chunks per head row
    constexpr int TPI = 64 / NCH;      // tokens per V wave-instruction (head_dim 96: 2 tokens x 24 chunks, 16 lanes idle)
    constexpr int NVI = 64 / TPI;      // V wave-instructions per page
    constexpr bool POW2 = (NCH & (NCH - 1)) == 0;
    static_assert(NCH % 2 == 0 && NCH <= 64 && 64 % TPI == 0, "head_dim must be a multiple of 8, at most 256, with 64 % (64 / (dh/4)) == 0");
    __shared__ float s_m[4], s_l[4];
    __shared__ float s_acc[4][DH];

    const int bh = blockIdx.x;
    const int b = bh / H, h = bh % H;
    const int arith_b = pool.arith_batch;   // > 0: physical page = logical page * arith_b + row (no table load in front of the K | V loads)
    const int t = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t m = (int64_t)b * T + t;
    const int64_t pf = pool.page_elems();
    const char* lbase = static_cast<const char*>(pool.base) + layer * pool.layer_stride * EB;
    const int g = lane / NCH, c = lane % NCH;
    const bool v_lane = g < TPI;       // lanes past TPI * NCH (head_dim 96) take no part in PV

    // Software pipeline over this wave's pages: while QK^T consumes K(page) the V(page) loads are in
    // flight, and while PV consumes V(page) the K(next page) loads are -- each wave keeps 16 KiB
    // (one operand tile) streaming at all times.  K/V are read once per step: non-temporal loads.
    f32x4 kk[NCH], vv[NVI];   // raw 16-byte groups: 4 floats or 8 halves
    auto page_base = [&](int pg, int isv) {
        const int phys = arith_b > 0 ? pg * arith_b + b : page_table[b * max_pages + pg];
        return lbase + ((int64_t)(phys * 2 + isv) * H + h) * pf * EB;
    };
    auto ldraw = [](const char* p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); };
    // n_tok = tokens of the page that are in the cache: lanes (= tokens) past the end of the last, partial page
    // re-read the last cached token instead of their own slot -- a wave instruction is 1 KiB of consecutive bytes,
    // so the tail of the page is never fetched from HBM (whole-page reads cost 8 % extra traffic on the benchmark
    // run).  Clamped addresses rather than predication: a load inside a branch makes the compiler drain vmcnt to 0
    // at the next use and the K/V pipelining below is lost.
    auto load_k = [&](int pg, int n_tok) {
        const char* kpage = page_base(pg, 0);
        const int tok = lane < n_tok ? lane : n_tok - 1;
#pragma unroll
        for (int i = 0; i < NCH; ++i) kk[i] = ldraw(kpage + (i * 64 + tok) * 16);
    };
    // Wave 0's first K tile (page 0 always holds a token) is requested before anything else is known, so that the page-table -> K
    // round trips overlap the ctx_len / q fetches; the length is not known yet: the whole page.  Waves 1..3 wait for the length
    // (one scalar round trip) and then request exactly the tokens their first page holds -- or nothing: in round 2 they too read
    // a whole page up front, 24 MB per launch at B x H = 512 that below 256 tokens of context nobody needs
    // (profiles/r3_attn_early_v_ab_by_ctx.txt: that burst, not a latency chain, was most of the 8.6 us a launch took at short
    // contexts).  Above 256 tokens their first pages are full and they start ~0.7 us later than wave 0, which has as many pages or
    // one more.
    const int zsplit = SPLIT ? (int)blockIdx.z : 0, nsplit = SPLIT ? (int)gridDim.z : 1;
    if (wave == 0 && zsplit == 0) load_k(0, 64);

    const int n_new = lens ? lens[b] : T;
    auto optr = [&](int d) { return tiled_out ? out + tiled_off((int)m, h * DH + d, C) : out + m * C + h * DH + d; };
    if (t >= n_new) {  // padded query row: defined output, never used
        if (threadIdx.x < DH && zsplit == 0) *optr(threadIdx.x) = 0.f;
        return;
    }
    const int len = ctx_len[b] + n_new;  // tokens visible to this query (whole cache, no mask)
    const int npages = (len + 63) >> 6;
    const int active = SPLIT ? min(nsplit, (npages + 3) >> 2) : 1;   // splits that have a page
    if (SPLIT && zsplit >= active) return;
    const int pg0 = zsplit * 4 + wave, pg_step = 4 * nsplit;
    if ((wave != 0 || zsplit != 0) && pg0 < npages) load_k(pg0, len - pg0 * 64);

    const float* qp = qkv + m * 3 * C + h * DH;
    float q[DH];
#pragma unroll
    for (int d = 0; d < DH; ++d) q[d] = qp[d] * scale;  // 1/sqrt(64) etc.: power-of-two scales are exact

    float mx = -INFINITY, lsum = 0.f;
    float acc[G];
#pragma unroll
    for (int e = 0; e < G; ++e) acc[e] = 0.f;

    for (int pg = pg0; pg < npages; pg += pg_step) {
        const char* vpage = page_base(pg, 1);
        const int n_tok = len - pg * 64;   // >= 1; > 64 for a full page
#pragma unroll
        for (int j = 0; j < NVI; ++j) {   // rows beyond the cache end: the last cached row instead (their p is 0)
            const int row = v_lane ? j * TPI + g : 0;
            vv[j] = ldraw(vpage + ((row < n_tok ? row : n_tok - 1) * DH + c * G) * EB);
        }

        float s0 = 0.f, s1 = 0.f;   // two chains; fp32: the same order as before (groups alternate between them)
#pragma unroll
        for (int i = 0; i < NCH; i += 2) {
#pragma unroll
            for (int e = 0; e < G; ++e) s0 = fmaf(q[G * i + e], kv_elem<F16>(kk[i], e), s0);
#pragma unroll
            for (int e = 0; e < G; ++e) s1 = fmaf(q[G * (i + 1) + e], kv_elem<F16>(kk[i + 1], e), s1);
        }
        const bool valid = (pg * 64 + lane) < len;
        const float s = valid ? (s0 + s1) : -INFINITY;
        const float tmax = wave_max(s);            // finite: every visited page holds >= 1 valid token
        const float mnew = fmaxf(mx, tmax);
        const float alpha = __expf(mx - mnew);     // first tile: exp(-inf) = 0
        const float p = valid ? __expf(s - mnew) : 0.f;
        lsum = lsum * alpha + p;
#pragma unroll
        for (int e = 0; e < G; ++e) acc[e] *= alpha;
        mx = mnew;
        if (pg + pg_step < npages) load_k(pg + pg_step, len - (pg + pg_step) * 64);  // K registers are free again: next page's K under PV
#pragma unroll
        for (int j = 0; j < NVI; ++j) {
            const float pv = __shfl(p, j * TPI + g, 64);
            const float pj = v_lane ? pv : 0.f;
#pragma unroll
            for (int e = 0; e < G; ++e) acc[e] = fmaf(pj, kv_elem<F16>(vv[j], e), acc[e]);
        }
    }
    // reduce the token groups g (lanes c, c+NCH, ...) and the per-lane softmax sums
    if (POW2) {
#pragma unroll
        for (int o = 32; o >= NCH; o >>= 1) {
#pragma unroll
            for (int e = 0; e < G; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
        }
    } else {   // lanes c < NCH collect the other token groups in a fixed order
        float tot[G];
#pragma unroll
        for (int e = 0; e < G; ++e) tot[e] = acc[e];
#pragma unroll
        for (int gg = 1; gg < TPI; ++gg) {
#pragma unroll
            for (int e = 0; e < G; ++e) tot[e] += __shfl(acc[e], lane + gg * NCH, 64);
        }
#pragma unroll
        for (int e = 0; e < G; ++e) acc[e] = tot[e];
    }
    lsum = wave_sum(lsum);
    if (lane == 0) { s_m[wave] = mx; s_l[wave] = lsum; }
    if (lane < NCH) {
#pragma unroll
        for (int e = 0; e < G; ++e) s_acc[wave][G * lane + e] = acc[e];
    }
    __syncthreads();
    float M = 0.f, num = 0.f, den = 0.f;
    if (threadIdx.x < DH) {
        M = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float f = (s_m[w] == -INFINITY) ? 0.f : __expf(s_m[w] - M);  // waves without pages
            num = fmaf(f, s_acc[w][threadIdx.x], num);
            den = fmaf(f, s_l[w], den);
        }
    }
    constexpr int PS = attn_part_floats(DH);   // floats per partial: acc[DH], max, sum (+ 2 of padding: 16-byte rows)
    const int64_t item = (int64_t)bh * T + t;
    float* part = sp.part + item * sp.max_split * PS;
    if (!SPLIT || active == 1) {   // the only workgroup of this query: done
        if (threadIdx.x < DH) *optr(threadIdx.x) = num / den;
        return;
    }
    // ---- leave the partial, count the arrival; the last one merges
    auto st_dev = [](float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto ld_dev = [](const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    if (threadIdx.x < DH) {
        st_dev(part + zsplit * PS + threadIdx.x, num);
        if (threadIdx.x == 0) { st_dev(part + zsplit * PS + DH, M); st_dev(part + zsplit * PS + DH + 1, den); }
    }
    // Hand-off without cache-maintenance fences (MI355X_MICROARCH.md, inter-workgroup visibility, "valid forms": a buffer_inv sc1 alone
    // is ~1.7 us, three of them made this form SLOWER than the plain kernel -- 180.0 against 176.9 us per B = 1 step): every store of
    // the partial is an sc1 (write-through, device-scope) store and is drained by its wave (vmcnt 0) before the workgroup's barrier;
    // then ONE lane adds to the counter with a device-scope atomic; the workgroup whose add returns active - 1 loads the partials, after
    // a barrier behind that add, with sc1 loads only (they bypass the L1 and the XCD's L2).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int s_last;
    if (threadIdx.x == 0) {
        const int old = __hip_atomic_fetch_add(sp.count + item, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = old == active - 1;
        if (s_last) __hip_atomic_store(sp.count + item, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x < DH) {
        // every partial requested before the first is used (clamped indices past `active`): one memory round trip, not one per split
        float pm[MGEA_ATTN_MAX_SPLIT], pl[MGEA_ATTN_MAX_SPLIT], pa[MGEA_ATTN_MAX_SPLIT];
#pragma unroll
        for (int z = 0; z < MGEA_ATTN_MAX_SPLIT; ++z) {
            const float* pz = part + (z < active ? z : 0) * PS;
            pm[z] = ld_dev(pz + DH); pl[z] = ld_dev(pz + DH + 1); pa[z] = ld_dev(pz + threadIdx.x);
        }
        float Mall = -INFINITY;
#pragma unroll
        for (int z = 0; z < MGEA_ATTN_MAX_SPLIT; ++z) Mall = z < active ? fmaxf(Mall, pm[z]) : Mall;
        float n = 0.f, d = 0.f;
#pragma unroll
        for (int z = 0; z < MGEA_ATTN_MAX_SPLIT; ++z) {   // split order, whoever arrived last: deterministic
            const float f = z < active ? __expf(pm[z] - Mall) : 0.f;
            n = fmaf(f, pa[z], n);
            d = fmaf(f, pl[z], d);
        }
        *optr(threadIdx.x) = n / d;
    }
}

int attn_split_count(int B, int H, int T, int max_pages) {
    // worth it only while the plain grid leaves the chip short of waves to hide the page round trips behind each other, and only for
    // the decode step's single query per row.  The switch's value is the largest number of (row, head) pairs that are split.
    const int limit = tune(TUNE_ATTN_SPLIT);
    if (limit <= 0 || T != 1 || B * H > limit || B * H > MGEA_ATTN_SPLIT_ITEMS || max_pages <= 4) return 1;
    int s = (max_pages + 3) / 4;
    const int room = 512 / (B * H);   // up to two workgroups per CU
    if (s > room) s = room;
    if (s > MGEA_ATTN_MAX_SPLIT) s = MGEA_ATTN_MAX_SPLIT;
    return s < 1 ? 1 : s;
}

int launch_attn_paged(const float* qkv, const KvPool& pool, int layer, const int32_t* page_table, int max_pages,
                      const int32_t* ctx_len, const int32_t* lens, float* out, int B, int T, int C, int tiled_out,
                      hipStream_t st, const AttnSplit* split) {
    const int H = pool.H, dh = pool.dh;
    MGEA_REQUIRE(H * dh == C, MGEA_EINVAL, "attention: n_head*head_dim != d_model");
    MGEA_REQUIRE(T <= 65535, MGEA_EINVAL, "attention: too many new tokens per row (%d)", T);
    const float scale = 1.0f / sqrtf((float)dh);
    KvPool pool_k = pool;
    if (!tune(TUNE_ATTN_ARITH_PAGES)) pool_k.arith_batch = 0;
    const int ns = split && split->part && split->count ? attn_split_count(B, H, T, max_pages) : 1;
    MGEA_REQUIRE(ns == 1 || (ns <= split->max_split && B * H * T <= split->max_items), MGEA_EINVAL, "attention: split scratch too small");
    dim3 grid(B * H, T, ns);
    const AttnSplit sp = split ? *split : AttnSplit{};
#define MGEA_ATTN(DH, F) do { if (ns > 1) hipLaunchKernelGGL((attn_paged_kernel<DH, F, true>), grid, dim3(256), 0, st, qkv, pool_k, layer, page_table, max_pages, ctx_len, lens, out, H, T, C, scale, tiled_out, sp); \
                              else hipLaunchKernelGGL((attn_paged_kernel<DH, F, false>), grid, dim3(256), 0, st, qkv, pool_k, layer, page_table, max_pages, ctx_len, lens, out, H, T, C, scale, tiled_out, sp); } while (0)
    if (pool.f16) {
        switch (dh) {
            case 32: MGEA_ATTN(32, true); break;
            case 64: MGEA_ATTN(64, true); break;
            case 128: MGEA_ATTN(128, true); break;
            default:
                MGEA_REQUIRE(false, MGEA_EINVAL, "attention over fp16 KV pages: head_dim %d not supported (32, 64, 128)", dh);
        }
    } else {
        switch (dh) {
            case 32: MGEA_ATTN(32, false); break;
            case 64: MGEA_ATTN(64, false); break;
            case 96: MGEA_ATTN(96, false); break;
            case 128: MGEA_ATTN(128, false); break;
            default:
                MGEA_REQUIRE(false, MGEA_EINVAL, "attention: head_dim %d not supported (32, 64, 96, 128)", dh);
        }
    }
#undef MGEA_ATTN
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

}  // namespace mgea
