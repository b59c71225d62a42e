// Token sampler over a logits matrix [B, V]: replaces api_cache.py:169-178
//   logits / temperature -> additive -1e10 outside the top-k -> softmax -> multinomial(1)
// plus a build-defined nucleus (top-p) cut that the reference does not have.
//
// One 256-thread workgroup per row; the row lives in registers (thread t owns logits t, t + 256, ...), nothing is sorted.  (Round 4 measured
// two ways of spending more hardware on a row, both NEUTRAL: 1024 threads per row -- 21.9 / 32.3 us for top-k / top-p at B = 64, V = 8324
// against 20 / 31.9 -- and per-lane counters with one DPP reduction per pass instead of a ballot + s_bcnt1 per logit -- 22.8 us.  What did
// pay is taking the row out of the passes: "top-k without a pass over the row per bit" and "top-p without a top-k" below, 21.0 -> 11.5
// and 29.6 -> 20.5 us.)
//   * top-k : exact k-th largest logit by bit-wise bisection of order-preserving uint keys (integer counts ->
//             deterministic), over a pivot-selected candidate list for top_k <= 64, over the registers otherwise; kept = {logit > k-th} plus as many of the
//             entries EQUAL to the k-th as it takes to keep exactly k, lowest ids first (topk + scatter_ of
//             api_cache.py:172-175 keeps exactly top_k entries; which of several tied ones torch keeps is
//             unspecified, lowest-id is this build's rule).  exp(-1e10) underflows to exactly 0 in fp32, so
//             "mask then softmax" == "softmax over kept".
//   * top-p : the nucleus {i : mass of strictly larger logits < top_p} by the same bisection over
//             fixed-point (2^-40) probability masses (64-bit integer sums -> deterministic), over the candidates of the
//             top-k path or of a mass histogram ("top-p without a top-k" below), over the registers otherwise;
//             kept = {logit >= boundary}.
//   * draw  : u from Philox4x32-10 keyed (seed; row, step); inverse CDF over the kept set in
//             thread-major order (any fixed order gives the same distribution).  torch.multinomial's
//             stream cannot be reproduced on device: equality with the reference is
//             distributional, the pre-draw probabilities are compared exactly (probs_out).
// top_k == 1 is the argmax path (rowops.hip, ties to the lowest id) for the ids.
// The sampler's scalars (temperature, top_k, top_p, seed; eos for the fused tail) are read from a SamplerParams
// record in DEVICE memory when the caller passes one: the captured decode graph then serves every request, whatever
// its seed (the reference's endpoint draws a fresh one per call, api_cache.py:204).
#include "common.h"

namespace mgea {

__device__ __forceinline__ uint32_t fkey(float f) {  // larger float -> larger key
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c[0], p1 = (uint64_t)M1 * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += W0; k1 += W1;
    }
}

constexpr int SAMP_NT = 256, SAMP_NW = SAMP_NT / 64;   // threads / waves per row

__device__ __forceinline__ float block_sum_f(float v, float* red /* [SAMP_NW] */) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < SAMP_NW; ++w) t += red[w];          // fixed order: deterministic
    return t;
}

// Whole-wave reductions on DPP alone (no LDS crossbar: a ds_bpermute round trip is ~100+ cycles, and the bisections below reduce once or
// twice per pass): butterfly inside each row of 16 lanes, row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3 (gfx9 DPP
// controls), the result read from lane 63 -- wave-uniform, in a scalar register.
#define MGEA_DPP(v, ctrl, rows) __builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rows, 0xF, true)
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    v += (uint32_t)MGEA_DPP(v, 0xB1, 0xF);    // quad_perm [1,0,3,2]
    v += (uint32_t)MGEA_DPP(v, 0x4E, 0xF);    // quad_perm [2,3,0,1]
    v += (uint32_t)MGEA_DPP(v, 0x141, 0xF);   // row_half_mirror
    v += (uint32_t)MGEA_DPP(v, 0x140, 0xF);   // row_mirror: every lane holds its row's sum
    v += (uint32_t)MGEA_DPP(v, 0x142, 0xA);   // row_bcast:15 -> rows 1, 3
    v += (uint32_t)MGEA_DPP(v, 0x143, 0xC);   // row_bcast:31 -> rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t umax32(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {   // rows not written by a masked step see 0: neutral for an unsigned maximum
    v = umax32(v, (uint32_t)MGEA_DPP(v, 0xB1, 0xF));
    v = umax32(v, (uint32_t)MGEA_DPP(v, 0x4E, 0xF));
    v = umax32(v, (uint32_t)MGEA_DPP(v, 0x141, 0xF));
    v = umax32(v, (uint32_t)MGEA_DPP(v, 0x140, 0xF));
    v = umax32(v, (uint32_t)MGEA_DPP(v, 0x142, 0xA));
    v = umax32(v, (uint32_t)MGEA_DPP(v, 0x143, 0xC));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Boundary search shared by top-k (weights = 1, target = k) and top-p (weights = fixed-point mass): the largest key t
// with weight({key >= t, key >= floor_key}) >= target, i.e. the k-th largest logit / the logit at which the nucleus
// mass is reached; floor_key if even everything weighs less than the target.  Bit-by-bit bisection of the 32-bit
// order-preserving key over the thread's register-resident keys (0 = not a candidate): 32 counting passes, integer
// arithmetic only -> exact and deterministic.  Counts come from wave ballots (no reduction at all); masses are summed
// as two 20-bit halves in 32-bit lanes.  One barrier per pass (the cross-wave slots alternate with the pass parity).
// (The first version descended a radix tree through LDS histograms; with ~8k logits sharing a handful of exponent
// bytes its 64-bit LDS atomics serialised.)
template <bool MASS, int MAXE>
__device__ __forceinline__ uint32_t bisect_boundary(const uint32_t (&key)[MAXE], const uint32_t (&whi)[MAXE], const uint32_t (&wlo)[MAXE],
                                                    uint32_t floor_key, unsigned long long target, unsigned long long* red /* [2 * SAMP_NW] */,
                                                    unsigned long long* weight_at_boundary = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t prefix = 0u;
    unsigned long long at = ~0ull;   // weight({key >= prefix}): everything while prefix == 0
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t cand = prefix | (1u << bit);
        const uint32_t lim = cand > floor_key ? cand : floor_key;   // only keys >= floor_key are candidates
        unsigned long long mine;
        if (MASS) {
            uint32_t hi = 0u, lo = 0u;
#pragma unroll
            for (int j = 0; j < MAXE; ++j) {
                const bool in = key[j] >= lim;
                hi += in ? whi[j] : 0u;
                lo += in ? wlo[j] : 0u;
            }
            mine = ((unsigned long long)wave_sum_u32(hi) << 20) + wave_sum_u32(lo);
        } else {
            uint32_t cnt = 0u;
#pragma unroll
            for (int j = 0; j < MAXE; ++j) cnt += (uint32_t)__popcll(__ballot(key[j] >= lim));
            mine = cnt;
        }
        unsigned long long* slot = red + SAMP_NW * (bit & 1);
        if (lane == 0) slot[wave] = mine;
        __syncthreads();
        unsigned long long tot = 0;
#pragma unroll
        for (int w = 0; w < SAMP_NW; ++w) tot += slot[w];
        if (tot >= target) { prefix = cand; at = tot; }
    }
    __syncthreads();   // the slots are reused by the caller
    if (weight_at_boundary) *weight_at_boundary = at;
    return prefix > floor_key ? prefix : floor_key;
}

// ---- top-k without a pass over the row per bit (round 4).  A wave issues one instruction per ~5 cycles when it is alone on its SIMD, as
// here, scalar ones included: the block-wide bisection is 32 x (36 compares + 72 scalar count instructions), ~12 us of the kernel's 21.
// (A first attempt kept those passes and only dropped their barrier -- every wave bisecting its own quarter, candidates merged at the end:
// 25 us, SLOWER; the passes themselves are the cost.)  Instead a PIVOT cuts the row down to a few dozen candidates first:
//   1. every lane takes the largest of its keys; every wave bisects ITS 64 lane maxima (one compare per pass) for the j-th largest,
//      j = ceil(k / 4); the pivot P is the smallest of the four.  At least 4 j >= k logits are >= P, so the k-th largest is >= P, and with
//      one logit in ~36 being a lane maximum the number of logits >= P is about k (k = 50: 60-70 of 8324);
//   2. one pass over the row writes the keys >= P into the wave's 64 LDS slots (ballot prefix; most ballots are empty);
//   3. after a barrier every wave bisects the <= 4 x 64 candidates (4 per lane) for the k-th largest of the row, and counts the
//      candidates at or above it -- that is every such logit of the row, so ties are seen exactly as before.
// A wave with more than 64 logits >= P (heavy ties, or fewer than j valid lanes in some wave) sends the block to the block-wide
// bisection.  Integer counts throughout: same boundary key either way.  top_k <= SAMP_KFAST takes this path.
constexpr int SAMP_KFAST = 64;

// p (a probability, <= 1) as the 2^-40 fixed-point integer floor(p * 2^40), split as hi = bits 20.., lo = bits 0..19 -- the integers the
// mass sums are made of.  All float operations below are exact (powers of two, an integer part, a remainder), so this is the same integer
// as (unsigned long long)((double)p * 2^40) at a third of the instructions (no f64, no 64-bit conversion).
__device__ __forceinline__ void mass_fixed(float p, uint32_t& hi, uint32_t& lo) {
    const float t = p * 1048576.0f;
    hi = (uint32_t)t;
    lo = (uint32_t)((t - (float)hi) * 1048576.0f);
}

// The non-zero keys a wave holds lie in [lo, hi]: every boundary the bisections below look for does too, so the bits above the highest
// bit in which lo and hi differ are already known.  Returns that common prefix and the number of low bits left to decide (32: nothing
// known -- no non-zero key at all, or keys on both sides of the top bit).
template <int N>
__device__ __forceinline__ uint32_t wave_common_prefix(const uint32_t (&key)[N], int* nbits) {
    uint32_t hi = 0u, lo_inv = 0u;   // lo via the maximum of the complements
#pragma unroll
    for (int j = 0; j < N; ++j) {
        hi = hi > key[j] ? hi : key[j];
        const uint32_t c = key[j] != 0u ? ~key[j] : 0u;
        lo_inv = lo_inv > c ? lo_inv : c;
    }
    hi = wave_max_u32(hi);
    const uint32_t lo = ~wave_max_u32(lo_inv);   // 0xFFFFFFFF when there is no non-zero key
    const uint32_t diff = hi ^ lo;
    const int nb = (hi == 0u) ? 32 : 32 - __clz((int)diff);   // diff == 0: one distinct key, nothing left to decide (clz(0) = 32)
    *nbits = nb;
    return nb >= 32 ? 0u : (hi >> nb) << nb;
}

template <int N>
__device__ __forceinline__ uint32_t wave_kth_largest(const uint32_t (&key)[N], uint32_t k) {   // 0 when fewer than k keys are non-zero
    // (starting below the keys' common prefix, as wave_mass_boundary does, was measured here too: the two wave reductions cost more than
    // the ~10 passes of N compares they save -- top-k 50 at 12.2 us against 11.5)
    uint32_t prefix = 0u;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t cand = prefix | (1u << bit);
        uint32_t cnt = 0u;
#pragma unroll
        for (int j = 0; j < N; ++j) cnt += (uint32_t)__popcll(__ballot(key[j] >= cand));
        if (cnt >= k) prefix = cand;
    }
    return prefix;
}

// keys >= pivot (> 0) of this wave's registers -> slots[0, 64), zeros behind them; returns how many there were (may exceed 64)
template <int N>
__device__ __forceinline__ int wave_publish_ge(const uint32_t (&key)[N], uint32_t pivot, uint32_t* slots) {
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    int base = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const bool up = key[j] >= pivot;
        const unsigned long long m = __ballot(up);
        if (m != 0ull) {   // wave-uniform, rarely taken
            const int pos = base + __popcll(m & below);
            if (up && pos < SAMP_KFAST) slots[pos] = key[j];
            base += __popcll(m);
        }
    }
    if (lane >= base) slots[lane] = 0u;   // SAMP_KFAST == 64: one lane per slot
    return base;
}

// ---- top-p without a top-k (or behind one larger than SAMP_KFAST): no count bounds the nucleus, so the candidates come from a mass
// histogram instead.  Bucket = floor((max - x) * scale), 256 of them over the row's own span below its maximum -- or over the 28
// units beyond which a probability rounds to 0 in the 2^-40 fixed point, if the row spans more -- non-increasing in the key, so the
// buckets are ordered slices of the sorted row.  (With a fixed 28-unit span a random-weight model's nearly flat rows -- all 8324 logits
// within ~2.5 units -- put hundreds of entries into the boundary bucket and every step fell back: configs[4]'s sampler 32 -> 37.6 us.)
//   1. one pass adds every entry's fixed-point mass to its bucket (64-bit LDS atomics: integer sums, order-free);
//   2. every wave scans the 256 buckets (4 per lane) for the first one at which the running mass reaches the target: the nucleus
//      boundary lies in it, and everything in the buckets before it is kept;
//   3. one pass writes that bucket's keys into the wave's 64 LDS slots; every wave bisects those <= 4 x 64 candidates for the largest
//      key at which (mass before the bucket + mass of the candidates at or above it) reaches the target.
// Same integers as the block-wide bisection -> same boundary key.  More than 64 candidates in a wave: the block-wide bisection.
constexpr float SAMP_HSPAN = 28.0f;

__device__ __forceinline__ int mass_bucket(float mx, float x, float scale) {
    const int b = (int)((mx - x) * scale);   // v_cvt_i32_f32: NaN -> 0, +inf saturates
    return (unsigned)b > 255u ? 255 : b;
}

// first bucket whose inclusive running mass reaches `target`, and the mass before it; -1 when the whole row weighs less
__device__ __forceinline__ int wave_find_bucket(const unsigned long long* hist /* [256] in LDS */, unsigned long long target,
                                                unsigned long long* before) {
    const int lane = threadIdx.x & 63;
    unsigned long long h[4], c[5];
    c[0] = 0ull;
#pragma unroll
    for (int i = 0; i < 4; ++i) { h[i] = hist[4 * lane + i]; c[i + 1] = c[i] + h[i]; }
    unsigned long long inc = c[4];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    const unsigned long long excl = inc - c[4];
    const unsigned long long hit = __ballot(inc >= target);
    if (hit == 0ull) { *before = 0ull; return -1; }
    const int L = __ffsll((long long)hit) - 1;
    int i = 0;
#pragma unroll
    for (int q = 3; q >= 0; --q) i = excl + c[q + 1] >= target ? q : i;   // first of my four that reaches it
    unsigned long long bef = excl + (i == 0 ? c[0] : i == 1 ? c[1] : i == 2 ? c[2] : c[3]);
    const int b = __shfl(4 * lane + i, L, 64);
    *before = __shfl(bef, L, 64);
    return b;
}

// the mass bisection of bisect_boundary<true> over candidates a single wave holds (R per lane): same integers, same boundary
template <int R>
__device__ __forceinline__ uint32_t wave_mass_boundary(const uint32_t (&key)[R], const uint32_t (&whi)[R], const uint32_t (&wlo)[R],
                                                       uint32_t floor_key, unsigned long long target) {
    // candidates that count: key >= floor_key and non-zero (the others carry no mass: whi = wlo = 0)
    uint32_t live[R];
    uint32_t hi = 0u, lo = 0u;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        live[j] = (key[j] >= floor_key && (whi[j] | wlo[j]) != 0u) ? key[j] : 0u;
        hi += live[j] != 0u ? whi[j] : 0u;
        lo += live[j] != 0u ? wlo[j] : 0u;
    }
    const unsigned long long all = ((unsigned long long)wave_sum_u32(hi) << 20) + wave_sum_u32(lo);
    if (all < target) return floor_key;   // even everything weighs less: the bisection would end at prefix 0
    int nbits;
    uint32_t prefix = wave_common_prefix<R>(live, &nbits);   // mass({key >= lowest live key}) = all >= target: the boundary is among them
    for (int bit = nbits - 1; bit >= 0; --bit) {
        const uint32_t cand = prefix | (1u << bit);
        hi = 0u; lo = 0u;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const bool in = live[j] >= cand;
            hi += in ? whi[j] : 0u;
            lo += in ? wlo[j] : 0u;
        }
        const unsigned long long tot = ((unsigned long long)wave_sum_u32(hi) << 20) + wave_sum_u32(lo);
        if (tot >= target) prefix = cand;
    }
    return prefix > floor_key ? prefix : floor_key;
}

__device__ __forceinline__ float funkey(uint32_t k) {   // inverse of fkey
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// One SAMP_NT-thread workgroup per row; thread t owns the logits t, t + SAMP_NT, ... in registers (MAXE of them), so the row is
// read from memory once, in one batch of loads, and never goes through LDS.
template <int MAXE>
__global__ __launch_bounds__(SAMP_NT) void sample_kernel(const float* __restrict__ logits, int V, SamplerParams pv,
                                                    const SamplerParams* __restrict__ pd,
                                                    const int32_t* __restrict__ row_step, int64_t step_host,
                                                    int32_t* __restrict__ ids_out, float* __restrict__ probs_out,
                                                    TailArgs tail, int fuse_tail, int wave_select) {
    constexpr int NT = SAMP_NT, NW = SAMP_NW;
    static_assert(SAMP_KFAST == 64, "wave_publish_ge fills one slot per lane");
    __shared__ unsigned long long red64[2 * NW];
    __shared__ uint32_t s_cand[NW * SAMP_KFAST];
    __shared__ unsigned long long s_hist[NT];
    static_assert(NT == 256, "the mass histogram has one bucket per thread");
    __shared__ int s_tie[NW], s_fit[NW];
    if (pd) pv = *pd;   // device-resident scalars (one 32-byte scalar load) win over the by-value copy
    const float temperature = pv.temperature, top_p = pv.top_p;
    const int top_k = pv.top_k;
    const uint64_t seed = ((uint64_t)pv.seed_hi << 32) | pv.seed_lo;
    __shared__ float redf[NW];
    __shared__ float s_scan[NW], s_low[NW];
    __shared__ int s_thread, s_choice;
    __shared__ float sh_tail[8];

    const int b = blockIdx.x, tid = threadIdx.x;
    const float* lg = logits + (int64_t)b * V;
    int st_step = 0, st_fed = 0, st_len = 0, st_done = 0;   // the row's loop state, for the fused tail
    if (fuse_tail && tid == 0) {
        st_step = tail.s.row_step[b]; st_fed = tail.s.cur_ids[b]; st_len = tail.s.ctx_len[b]; st_done = tail.s.done[b];
    }
    // the whole row is requested at once (a rolled loop pays one ~1 us round trip per pass: the row was just
    // written by the head kernel and sits in another XCD's L2 / the Infinity Cache)
    float x[MAXE];
#pragma unroll
    for (int j = 0; j < MAXE; ++j) {
        const int i = tid + NT * j;
        x[j] = i < V ? lg[i] : -INFINITY;
    }
    float mx = -INFINITY;
    if (temperature != 1.0f) {   // logits / temperature (api_cache.py:170); x / 1 is x
#pragma unroll
        for (int j = 0; j < MAXE; ++j) x[j] = x[j] / temperature;
    }
#pragma unroll
    for (int j = 0; j < MAXE; ++j) mx = fmaxf(mx, x[j]);
    mx = wave_max(mx);
    if ((tid & 63) == 0) redf[tid >> 6] = mx;
    __syncthreads();
    mx = redf[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, redf[w]);

    uint32_t key[MAXE], whi[MAXE], wlo[MAXE];
#pragma unroll
    for (int j = 0; j < MAXE; ++j) {
        key[j] = (tid + NT * j < V) ? fkey(x[j]) : 0u;   // 0 = below every candidate
        whi[j] = wlo[j] = 0u;
    }
    uint32_t keep_key = 0u;  // keep everything
    uint32_t cand[NW] = {};  // wave-select path: this lane's share of the candidates (all waves hold the same list)
    bool cand_cover = false; // ... and they contain every entry >= keep_key (no ties beyond top_k)
    if (top_k > 0 && top_k < V) {
        unsigned long long n_ge = 0;
        bool fast = false;
        if (wave_select && top_k <= SAMP_KFAST) {
            uint32_t lm[1] = {key[0]};
#pragma unroll
            for (int j = 1; j < MAXE; ++j) lm[0] = lm[0] > key[j] ? lm[0] : key[j];
            const uint32_t piv_w = wave_kth_largest<1>(lm, (uint32_t)((top_k + NW - 1) / NW));
            if ((tid & 63) == 0) s_tie[tid >> 6] = (int)piv_w;
            __syncthreads();
            uint32_t pivot = (uint32_t)s_tie[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) pivot = pivot < (uint32_t)s_tie[w] ? pivot : (uint32_t)s_tie[w];
            if (pivot != 0u) {   // block-uniform; 0: some wave has fewer than j valid lanes (a tiny vocabulary)
                const int n_w = wave_publish_ge<MAXE>(key, pivot, s_cand + (tid >> 6) * SAMP_KFAST);
                if ((tid & 63) == 0) s_fit[tid >> 6] = n_w;
            }
            __syncthreads();
            fast = pivot != 0u;
            if (fast) {
#pragma unroll
                for (int w = 0; w < NW; ++w) fast = fast && s_fit[w] <= SAMP_KFAST;
            }
            if (fast) {
#pragma unroll
                for (int r = 0; r < NW; ++r) cand[r] = s_cand[(tid & 63) + 64 * r];
                keep_key = wave_kth_largest<NW>(cand, (uint32_t)top_k);   // >= pivot: at least top_k candidates exist
#pragma unroll
                for (int r = 0; r < NW; ++r) n_ge += (unsigned long long)__popcll(__ballot(cand[r] >= keep_key));
                cand_cover = n_ge == (unsigned long long)top_k;   // more: ties, trimmed below in the registers only
            }
        }
        if (!fast) keep_key = bisect_boundary<false, MAXE>(key, whi, wlo, 0u, (unsigned long long)top_k, red64, &n_ge);
        if (n_ge > (unsigned long long)top_k) {
            // several logits equal the k-th largest: keep exactly top_k entries, the tied ones by ascending id
            // (id = tid + NT j, so the order is j-major, thread-minor).  Rare (exact fp32 ties): block-uniform branch.
            uint32_t n_gt = 0u;
#pragma unroll
            for (int j = 0; j < MAXE; ++j) n_gt += (uint32_t)__popcll(__ballot(key[j] > keep_key));
            if ((tid & 63) == 0) s_tie[tid >> 6] = (int)n_gt;
            __syncthreads();
            int n_gt_all = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) n_gt_all += s_tie[w];
            const int need = top_k - n_gt_all;   // >= 1 tied entries survive
            __syncthreads();
            int seen = 0;   // tied entries with a lower id than this pass's (block-uniform)
#pragma unroll
            for (int j = 0; j < MAXE; ++j) {
                const bool tie = key[j] == keep_key;
                const unsigned long long bal = __ballot(tie);
                if ((tid & 63) == 0) s_tie[tid >> 6] = __popcll(bal);
                __syncthreads();
                int before = seen + __popcll(bal & ((1ull << (tid & 63)) - 1ull));
                for (int w = 0; w < NW; ++w) {
                    if (w < (tid >> 6)) before += s_tie[w];
                    seen += s_tie[w];
                }
                if (tie && before >= need) key[j] = 0u;   // 0 = not a candidate: below every kept key
                __syncthreads();
            }
        }
    }
    if (top_p > 0.f && top_p < 1.f) {
        float z = 0.f, lowest = mx;   // lowest: the smallest kept logit within SAMP_HSPAN of the maximum (the histogram's span)
#pragma unroll
        for (int j = 0; j < MAXE; ++j) z += (key[j] != 0u && key[j] >= keep_key) ? __expf(x[j] - mx) : 0.f;
        if (!cand_cover) {   // block-uniform: only the histogram path needs the span
#pragma unroll
            for (int j = 0; j < MAXE; ++j)
                lowest = (key[j] != 0u && key[j] >= keep_key && x[j] >= mx - SAMP_HSPAN) ? fminf(lowest, x[j]) : lowest;
            lowest = -wave_max(-lowest);
            if ((tid & 63) == 0) s_low[tid >> 6] = lowest;   // rides on block_sum_f's barriers
        }
        const float invZ = 1.0f / block_sum_f(z, redf);
        if (!cand_cover) {
#pragma unroll
            for (int w = 0; w < NW; ++w) lowest = fminf(lowest, s_low[w]);
        }
        const float hscale = 255.99f / (mx - lowest);   // a one-value row: inf, every entry in bucket 0 (NaN -> 0) and the fallback below
        const unsigned long long target = (unsigned long long)((double)top_p * 1099511627776.0);
        if (cand_cover) {
            // the kept set is among the candidates every wave already holds: the nucleus boundary without a barrier per bit.  The
            // masses are the same integers (same x, mx, invZ), so the boundary is the one the block-wide bisection finds.
            uint32_t chi[NW], clo[NW];
#pragma unroll
            for (int r = 0; r < NW; ++r) {
                chi[r] = clo[r] = 0u;
                if (cand[r] != 0u && cand[r] >= keep_key) mass_fixed(__expf(funkey(cand[r]) - mx) * invZ, chi[r], clo[r]);
            }
            keep_key = wave_mass_boundary<NW>(cand, chi, clo, keep_key, target);
        } else {
#pragma unroll
            for (int j = 0; j < MAXE; ++j) {
                if (key[j] != 0u && key[j] >= keep_key) mass_fixed(__expf(x[j] - mx) * invZ, whi[j], wlo[j]);   // hi <= 2^20: 64 lanes x MAXE of them stay below 2^32
            }
            bool done = false;
            if (wave_select) {
                s_hist[tid] = 0ull;   // NT == 256 buckets
                __syncthreads();
#pragma unroll
                for (int j = 0; j < MAXE; ++j) {
                    const unsigned long long w = ((unsigned long long)whi[j] << 20) + wlo[j];
                    if (w != 0ull) atomicAdd(&s_hist[mass_bucket(mx, x[j], hscale)], w);
                }
                __syncthreads();
                unsigned long long before = 0ull;
                const int bstar = wave_find_bucket(s_hist, target, &before);   // the same in every wave
                if (bstar < 0) {
                    done = true;   // even everything weighs less than the target: keep all of it (keep_key stays the floor)
                } else {
                    uint32_t inb[MAXE];   // the keys of that bucket (0 elsewhere)
#pragma unroll
                    for (int j = 0; j < MAXE; ++j) inb[j] = ((whi[j] | wlo[j]) != 0u && mass_bucket(mx, x[j], hscale) == bstar) ? key[j] : 0u;
                    const int n_w = wave_publish_ge<MAXE>(inb, 1u, s_cand + (tid >> 6) * SAMP_KFAST);
                    if ((tid & 63) == 0) s_fit[tid >> 6] = n_w;
                    __syncthreads();
                    bool fits = true;
#pragma unroll
                    for (int w = 0; w < NW; ++w) fits = fits && s_fit[w] <= SAMP_KFAST;
                    if (fits) {
                        uint32_t chi[NW], clo[NW];
#pragma unroll
                        for (int r = 0; r < NW; ++r) {
                            cand[r] = s_cand[(tid & 63) + 64 * r];
                            chi[r] = clo[r] = 0u;
                            if (cand[r] != 0u) mass_fixed(__expf(funkey(cand[r]) - mx) * invZ, chi[r], clo[r]);
                        }
                        keep_key = wave_mass_boundary<NW>(cand, chi, clo, keep_key, target - before);
                        done = true;
                    }
                    __syncthreads();   // s_fit / s_cand / red64 are reused by the fallback
                }
            }
            if (!done) keep_key = bisect_boundary<true, MAXE>(key, whi, wlo, keep_key, target, red64);
        }
    }

    // ---- final distribution over the kept set; CDF order = thread-major (t, then t + NT, ...): any fixed order gives
    // the same distribution
    float e[MAXE];
    float loc = 0.f;
#pragma unroll
    for (int j = 0; j < MAXE; ++j) {
        e[j] = (key[j] != 0u && key[j] >= keep_key) ? __expf(x[j] - mx) : 0.f;
        loc += e[j];
    }
    // exclusive prefix of the per-thread masses in thread order: a fixed tree (wave scan, then the wave totals left
    // to right) -> deterministic
    float inc = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float up = __shfl_up(inc, o, 64);
        if ((tid & 63) >= o) inc += up;
    }
    if ((tid & 63) == 63) s_scan[tid >> 6] = inc;
    if (tid == 0) { s_thread = -1; s_choice = -1; }
    __syncthreads();
    float wpre = 0.f, total = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        if (w == (tid >> 6)) wpre = total;
        total += s_scan[w];
    }
    const float pre = wpre + (inc - loc);
    if (probs_out) {
        const float inv = 1.0f / total;
#pragma unroll
        for (int j = 0; j < MAXE; ++j)
            if (tid + NT * j < V) probs_out[(int64_t)b * V + tid + NT * j] = e[j] * inv;
    }
    if (!ids_out) return;
    uint32_t ctr[4] = {(uint32_t)b, (uint32_t)(row_step ? row_step[b] : (int32_t)step_host),
                       (uint32_t)(step_host >> 32), 0x6d676561u};
    philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), ctr);
    const float u = (float)(ctr[0] >> 8) * (1.0f / 16777216.0f);  // [0, 1)
    const float target = u * total;
    if (loc > 0.f && pre <= target) atomicMax(&s_thread, tid);  // thread 0's pre == 0 <= target
    __syncthreads();
    if (tid == s_thread) {
        float acc = pre;
        int pick = -1, last = -1;
#pragma unroll
        for (int j = 0; j < MAXE; ++j) {
            if (e[j] > 0.f && pick < 0) {
                last = j;
                acc += e[j];
                if (acc > target) pick = j;
            }
        }
        const int jj = pick >= 0 ? pick : last;
        s_choice = jj >= 0 ? tid + NT * jj : -1;
    }
    __syncthreads();
    const int tok = s_choice >= 0 ? s_choice : 0;
    if (fuse_tail) advance_embed_row(b, tok, tail, ids_out, st_step, st_fed, st_len, st_done, sh_tail);   // writes ids_out[b] too
    else if (tid == 0) ids_out[b] = tok;
}

int launch_sample(const float* logits, int B, int V, const mgea_sampler_config& s, const SamplerParams* params_dev,
                  const int32_t* row_step_dev, int64_t step_host, int32_t* ids_out, float* probs_out, hipStream_t st, const TailArgs* tail) {
    MGEA_REQUIRE(params_dev || s.temperature > 0.f, MGEA_EINVAL, "sampler: temperature must be > 0");
    MGEA_REQUIRE(V > 0 && V <= MGEA_SAMPLER_MAX_VOCAB, MGEA_EINVAL, "sampler: vocab %d exceeds the register-resident row (%d)", V,
                 MGEA_SAMPLER_MAX_VOCAB);
    MGEA_REQUIRE(!tail || (ids_out && tail->C % 4 == 0 && tail->C <= 4096), MGEA_EINVAL, "sampler: bad fused-tail arguments");
    const TailArgs t = tail ? *tail : TailArgs{};
    const SamplerParams pv = sampler_params(s);
    static_assert(SAMP_NT * 56 >= MGEA_SAMPLER_MAX_VOCAB, "the register-resident row must hold the largest vocabulary");
    if (V <= SAMP_NT * 36)
        hipLaunchKernelGGL(sample_kernel<36>, dim3(B), dim3(SAMP_NT), 0, st, logits, V, pv, params_dev, row_step_dev, step_host, ids_out,
                           probs_out, t, tail ? 1 : 0, tune(TUNE_SAMPLER_WAVE_SELECT));
    else
        hipLaunchKernelGGL(sample_kernel<56>, dim3(B), dim3(SAMP_NT), 0, st, logits, V, pv, params_dev, row_step_dev, step_host, ids_out,
                           probs_out, t, tail ? 1 : 0, tune(TUNE_SAMPLER_WAVE_SELECT));
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

// params_dev <- s, stream-ordered (a kernel argument, so no host buffer has to outlive the call)
__global__ void set_sampler_params_kernel(SamplerParams* dst, SamplerParams v) { *dst = v; }

int launch_set_sampler_params(SamplerParams* params_dev, const mgea_sampler_config& s, hipStream_t st) {
    hipLaunchKernelGGL(set_sampler_params_kernel, dim3(1), dim3(1), 0, st, params_dev, sampler_params(s));
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

}  // namespace mgea
