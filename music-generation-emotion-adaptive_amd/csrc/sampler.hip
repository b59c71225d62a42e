// Token sampler over a logits matrix [B, V]: replaces api_cache.py:169-178
//   logits / temperature -> additive -1e10 outside the top-k -> softmax -> multinomial(1)
// plus a build-defined nucleus (top-p) cut that the reference does not have.
//
// One 256-thread workgroup per row; the row lives in LDS (V*4 bytes), nothing is sorted:
//   * top-k : exact k-th largest logit by a 4-pass radix select on order-preserving uint keys
//             (integer LDS histograms -> deterministic); kept = {logit >= k-th}.  exp(-1e10)
//             underflows to exactly 0 in fp32, so "mask then softmax" == "softmax over kept".
//   * top-p : the nucleus {i : mass of strictly larger logits < top_p} by the same radix descent
//             over fixed-point (2^-40) probability-mass histograms (64-bit LDS atomics ->
//             deterministic); kept = {logit >= boundary}.
//   * draw  : u from Philox4x32-10 keyed (seed; row, step); inverse CDF over the kept set in
//             index order (any fixed order gives the same distribution).  torch.multinomial's
//             stream cannot be reproduced on device: equality with the reference is
//             distributional, the pre-draw probabilities are compared exactly (probs_out).
// top_k == 1 is the argmax path (rowops.hip, ties to the lowest id) for the ids.
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

__device__ __forceinline__ unsigned long long shfl_down_u64(unsigned long long v, int o) {
    const uint32_t lo = __shfl_down((uint32_t)(v & 0xffffffffull), o, 64);
    const uint32_t hi = __shfl_down((uint32_t)(v >> 32), o, 64);
    return ((unsigned long long)hi << 32) | lo;
}

// inclusive suffix sum over the 256 threads: result(t) = sum of v over threads >= t
__device__ __forceinline__ unsigned long long block_suffix_sum(unsigned long long v, unsigned long long* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long up = shfl_down_u64(v, o);
        if (lane + o < 64) v += up;
    }
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    for (int w = wave + 1; w < 4; ++w) v += red[w];
    return v;
}

__device__ __forceinline__ float block_sum_f(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// Radix descent shared by top-k (weights = 1, target = k) and top-p (weights = fixed-point mass):
// returns the smallest key t* present in {key >= floor_key} such that weight(keys > t*) < target.
template <bool MASS>
__device__ uint32_t radix_boundary(const float* sx, int V, uint32_t floor_key, unsigned long long target, float mx,
                                   float invZ, unsigned long long* hist, unsigned long long* red,
                                   uint32_t* s_prefix, unsigned long long* s_above, uint32_t* s_pick) {
    const int tid = threadIdx.x;
    if (tid == 0) { *s_prefix = 0u; *s_above = 0ull; }
    for (int pass = 3; pass >= 0; --pass) {
        hist[tid] = 0ull;
        if (tid == 0) *s_pick = 0xffffffffu;
        __syncthreads();
        const uint32_t prefix = *s_prefix;
        const unsigned long long above = *s_above;
        const uint32_t himask = pass == 3 ? 0u : (0xffffffffu << ((pass + 1) * 8));
        for (int i = tid; i < V; i += 256) {
            const float x = sx[i];
            const uint32_t k = fkey(x);
            if (k >= floor_key && (k & himask) == prefix) {
                unsigned long long w = 1ull;
                if (MASS) w = (unsigned long long)((double)(__expf(x - mx) * invZ) * 1099511627776.0);
                atomicAdd(&hist[(k >> (pass * 8)) & 255u], w);
            }
        }
        __syncthreads();
        const unsigned long long mine = hist[tid];
        const unsigned long long incl = block_suffix_sum(mine, red);
        const unsigned long long a_bin = above + (incl - mine);  // weight strictly above this bin
        // boundary bin: the lowest non-empty bin whose top element still has weight-above < target
        if (mine > 0ull && a_bin < target && a_bin + mine >= target) *s_pick = (uint32_t)tid;
        __syncthreads();
        const uint32_t first = *s_pick;
        __syncthreads();
        // everything fits under the target: take the lowest non-empty bin
        if (first == 0xffffffffu && mine > 0ull) atomicMin(s_pick, (uint32_t)tid);
        __syncthreads();
        const uint32_t pick = *s_pick;
        __syncthreads();
        if ((uint32_t)tid == pick) {
            *s_prefix = prefix | (pick << (pass * 8));
            *s_above = a_bin;
        }
        __syncthreads();
    }
    return *s_prefix;
}

__global__ __launch_bounds__(256) void sample_kernel(const float* __restrict__ logits, int V, float temperature,
                                                    int top_k, float top_p, uint64_t seed,
                                                    const int32_t* __restrict__ row_step, int64_t step_host,
                                                    int32_t* __restrict__ ids_out, float* __restrict__ probs_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* sx = reinterpret_cast<float*>(smem);  // [V] scaled logits
    __shared__ unsigned long long hist[256];
    __shared__ unsigned long long red64[4];
    __shared__ float redf[4];
    __shared__ float s_scan[256];
    __shared__ uint32_t s_prefix, s_pick;
    __shared__ unsigned long long s_above;
    __shared__ int s_thread, s_choice;

    const int b = blockIdx.x, tid = threadIdx.x;
    const float* lg = logits + (int64_t)b * V;

    float mx = -INFINITY;
    for (int i = tid; i < V; i += 256) {
        const float x = lg[i] / temperature;
        sx[i] = x;
        mx = fmaxf(mx, x);
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) redf[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));

    uint32_t keep_key = 0u;  // keep everything
    if (top_k > 0 && top_k < V)
        keep_key = radix_boundary<false>(sx, V, 0u, (unsigned long long)top_k, mx, 0.f, hist, red64, &s_prefix,
                                         &s_above, &s_pick);
    if (top_p > 0.f && top_p < 1.f) {
        float z = 0.f;
        for (int i = tid; i < V; i += 256) z += (fkey(sx[i]) >= keep_key) ? __expf(sx[i] - mx) : 0.f;
        const float Z = block_sum_f(z, redf);
        const unsigned long long target = (unsigned long long)((double)top_p * 1099511627776.0);
        keep_key = radix_boundary<true>(sx, V, keep_key, target, mx, 1.0f / Z, hist, red64, &s_prefix, &s_above,
                                        &s_pick);
    }

    // ---- final distribution over the kept set, chunked by thread in index order
    const int chunk = (V + 255) / 256;
    const int i0 = tid * chunk, i1 = min(V, i0 + chunk);
    float loc = 0.f;
    for (int i = i0; i < i1; ++i) loc += (fkey(sx[i]) >= keep_key) ? __expf(sx[i] - mx) : 0.f;
    s_scan[tid] = loc;
    if (tid == 0) { s_thread = -1; s_choice = -1; }
    __syncthreads();
    float pre = 0.f, total = 0.f;
    for (int t = 0; t < 256; ++t) {
        if (t == tid) pre = total;
        total += s_scan[t];
    }
    if (probs_out) {
        const float inv = 1.0f / total;
        for (int i = tid; i < V; i += 256)
            probs_out[(int64_t)b * V + i] = (fkey(sx[i]) >= keep_key) ? __expf(sx[i] - mx) * inv : 0.f;
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
        for (int i = i0; i < i1; ++i) {
            if (fkey(sx[i]) >= keep_key) {
                const float e = __expf(sx[i] - mx);
                if (e > 0.f) {
                    last = i;
                    acc += e;
                    if (acc > target) { pick = i; break; }
                }
            }
        }
        s_choice = pick >= 0 ? pick : last;
    }
    __syncthreads();
    if (tid == 0) ids_out[b] = s_choice >= 0 ? s_choice : 0;
}

int launch_sample(const float* logits, int B, int V, const mgea_sampler_config& s, const int32_t* row_step_dev,
                  int64_t step_host, int32_t* ids_out, float* probs_out, hipStream_t st) {
    MGEA_REQUIRE(s.temperature > 0.f, MGEA_EINVAL, "sampler: temperature must be > 0");
    MGEA_REQUIRE(V > 0 && V <= 14336, MGEA_EINVAL, "sampler: vocab %d exceeds the LDS row buffer (14336)", V);
    const size_t shmem = (size_t)round_up(V, 64) * sizeof(float);
    hipLaunchKernelGGL(sample_kernel, dim3(B), dim3(256), shmem, st, logits, V, s.temperature, s.top_k, s.top_p,
                       (uint64_t)s.seed, row_step_dev, step_host, ids_out, probs_out);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

}  // namespace mgea
