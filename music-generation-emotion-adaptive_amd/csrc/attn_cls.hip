// Attention of ONE query per sequence -- the [CLS] position -- against all keys of its sequence: what the last DistilBERT layer needs
// once nothing but hidden_state[:, 0] of its output is read (emotion_analysis/modeling.py:14-21 -> transformers
// DistilBertForSequenceClassification: pooled_output = hidden_state[:, 0]).  The keys and values of every position are still
// projected; the query, the attention, the out-projection and the FFN of the last layer run for the B [CLS] rows only (bert.hip).
// fp32 q and output; K / V rows are read from the packed qkv buffer of the engine (fp32 or bf16), key mask as in attn_dense.hip.
#include "common.h"

namespace mgea {

// One wave per (sequence, head).  Lane = (key group g, 16-byte-or-32-byte chunk c of the head dimension): C = DH / 8 chunks of 8
// dimensions, G = 64 / C key groups; iteration i of a block of 16 handles keys 16-iteration-block base + i * G + g.  Scores: every lane
// dots its 8 dimensions, three (two) xor-shuffles sum the chunks of a key, so that all C lanes of a group hold the key's score; the 16
// scores of a lane stay in registers and are exactly the probabilities its PV pass needs (no transposition): every lane accumulates its 8
// output dimensions over its keys, and xor-shuffles over the groups finish the sum.  All K loads of a block are requested before the
// first use, then all V loads (the first version read V two bytes per lane, one key per iteration: 64 us per launch on [256, 128, 12 x 64]
// against ~20 us for streaming the 100 MB of K | V once).
template <typename T> struct Row8;                                   // 8 consecutive elements of a K / V row as fp32
template <> struct Row8<float> {
    static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
        const float4 a = ld4(p), b = ld4(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
};
template <> struct Row8<__bf16> {
    static __device__ __forceinline__ void load(const __bf16* p, float (&v)[8]) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 u = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(u[i] << 16); v[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u); }
    }
};

template <typename T, int DH>
__global__ __launch_bounds__(64) void attn_cls_kernel(const float* __restrict__ q, const T* __restrict__ qkv, const int32_t* __restrict__ mask,
                                                     float* __restrict__ out, int S_pad, int H, float scale, const int32_t* __restrict__ cu) {
    constexpr int C = DH / 8, G = 64 / C, NI = 16, BLK = NI * G;    // keys per block of 16 iterations: 128 (DH 64) / 256 (DH 32)
    const int lane = threadIdx.x, c = lane % C, g = lane / C;
    const int b = blockIdx.x / H, hd = blockIdx.x % H, D = H * DH;
    // packed input (cu != NULL): the sequence's rows are cu[b] .. cu[b + 1] - 1, all of them real tokens (no mask)
    const int64_t row0 = cu ? (int64_t)cu[b] : (int64_t)b * S_pad;
    const int S = cu ? cu[b + 1] - cu[b] : S_pad;
    const T* kbase = qkv + row0 * 3 * D + D + hd * DH + c * 8;                // K of key t: kbase + t * 3 D; V: + D
    float qv[8];
    {
        const float4 q0 = ld4(q + (int64_t)b * D + hd * DH + c * 8), q1 = ld4(q + (int64_t)b * D + hd * DH + c * 8 + 4);
        qv[0] = q0.x * scale; qv[1] = q0.y * scale; qv[2] = q0.z * scale; qv[3] = q0.w * scale;
        qv[4] = q1.x * scale; qv[5] = q1.y * scale; qv[6] = q1.z * scale; qv[7] = q1.w * scale;
    }
    float m = -INFINITY, l = 0.f, acc[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) acc[d] = 0.f;
    for (int t0 = 0; t0 < S; t0 += BLK) {
        float sc[NI], kv[NI][8];
        bool ok[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int t = t0 + i * G + g;
            ok[i] = t < S;
            if (ok[i] && mask) ok[i] = mask[(int64_t)b * S_pad + t] != 0;
            Row8<T>::load(kbase + (int64_t)(t < S ? t : S - 1) * 3 * D, kv[i]);     // clamped: loaded, not used
        }
        float cm = -INFINITY;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < 8; ++d) a = fmaf(qv[d], kv[i][d], a);
#pragma unroll
            for (int o = 1; o < C; o <<= 1) a += __shfl_xor(a, o, 64);
            sc[i] = ok[i] ? a : -INFINITY;
            cm = fmaxf(cm, sc[i]);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i)                                    // the V rows of the block, requested before the reductions below
            Row8<T>::load(kbase + D + (int64_t)(t0 + i * G + g < S ? t0 + i * G + g : S - 1) * 3 * D, kv[i]);
#pragma unroll
        for (int o = C; o < 64; o <<= 1) cm = fmaxf(cm, __shfl_xor(cm, o, 64));
        const float mn = fmaxf(m, cm);
        if (mn == -INFINITY) continue;                               // no valid key so far (wave-uniform)
        const float alpha = __expf(m - mn);                          // exp(-inf) = 0 on the first block with a valid key
        float ps = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) acc[d] *= alpha;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const float p = ok[i] ? __expf(sc[i] - mn) : 0.f;
            ps += p;
#pragma unroll
            for (int d = 0; d < 8; ++d) acc[d] = fmaf(p, kv[i][d], acc[d]);
        }
        l = l * alpha + ps;                                          // per key group; summed over the groups at the end
        m = mn;
    }
#pragma unroll
    for (int o = C; o < 64; o <<= 1) {
        l += __shfl_xor(l, o, 64);
#pragma unroll
        for (int d = 0; d < 8; ++d) acc[d] += __shfl_xor(acc[d], o, 64);
    }
    if (g == 0) {
        const float inv = l > 0.f ? 1.0f / l : 0.f;                  // a sequence without any valid key gives zeros
        float* o = out + (int64_t)b * D + hd * DH + c * 8;
        *reinterpret_cast<float4*>(o) = make_float4(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
        *reinterpret_cast<float4*>(o + 4) = make_float4(acc[4] * inv, acc[5] * inv, acc[6] * inv, acc[7] * inv);
    }
}

int launch_attn_cls(const float* q, const void* qkv, int qkv_bf16, const int32_t* mask, float* out, int B, int S, int H, int dh, hipStream_t st,
                    const int32_t* cu) {
    MGEA_REQUIRE(!(cu && mask), MGEA_EINVAL, "attn_cls: packed rows carry no key mask");
    const float scale = 1.0f / sqrtf((float)dh);
    const dim3 grid(B * H), block(64);
    if (qkv_bf16) {
        MGEA_REQUIRE(dh == 64, MGEA_EINVAL, "attn_cls: bf16 keys need head_dim 64");
        hipLaunchKernelGGL((attn_cls_kernel<__bf16, 64>), grid, block, 0, st, q, (const __bf16*)qkv, mask, out, S, H, scale, cu);
    } else if (dh == 64) {
        hipLaunchKernelGGL((attn_cls_kernel<float, 64>), grid, block, 0, st, q, (const float*)qkv, mask, out, S, H, scale, cu);
    } else {
        MGEA_REQUIRE(dh == 32, MGEA_EINVAL, "attn_cls: head_dim %d not built (32 or 64)", dh);
        hipLaunchKernelGGL((attn_cls_kernel<float, 32>), grid, block, 0, st, q, (const float*)qkv, mask, out, S, H, scale, cu);
    }
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

}  // namespace mgea
