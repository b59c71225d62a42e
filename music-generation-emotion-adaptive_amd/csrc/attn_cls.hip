// Attention of ONE query per sequence -- the [CLS] position -- against all keys of its sequence: what the last DistilBERT layer needs
// once nothing but hidden_state[:, 0] of its output is read (emotion_analysis/modeling.py:14-21 -> transformers
// DistilBertForSequenceClassification: pooled_output = hidden_state[:, 0]).  The keys and values of every position are still
// projected; the query, the attention, the out-projection and the FFN of the last layer run for the B [CLS] rows only (bert.hip).
// One wave per (sequence, head): keys in chunks of 64 (lane = key), online softmax across chunks, then lane = output dimension.
// fp32 q and output; K / V rows are read from the packed qkv buffer of the engine (fp32 or bf16), key mask as in attn_dense.hip.
#include "common.h"

namespace mgea {

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }

template <typename T, int DH>
__global__ __launch_bounds__(64) void attn_cls_kernel(const float* __restrict__ q, const T* __restrict__ qkv, const int32_t* __restrict__ mask,
                                                     float* __restrict__ out, int S, int H, float scale) {
    const int lane = threadIdx.x;
    const int b = blockIdx.x / H, hd = blockIdx.x % H, D = H * DH;
    const T* kbase = qkv + (int64_t)b * S * 3 * D + D + hd * DH;     // K of key t: kbase + t * 3 D; V: + D
    float qv[DH];
#pragma unroll
    for (int d = 0; d < DH; d += 4) {
        const float4 v = ld4(q + (int64_t)b * D + hd * DH + d);
        qv[d] = v.x * scale; qv[d + 1] = v.y * scale; qv[d + 2] = v.z * scale; qv[d + 3] = v.w * scale;
    }
    float m = -INFINITY, l = 0.f, acc = 0.f;                         // running maximum, sum, and output dimension `lane` (lane < DH)
    for (int t0 = 0; t0 < S; t0 += 64) {
        const int t = t0 + lane;
        bool ok = t < S;
        if (ok && mask) ok = mask[(int64_t)b * S + t] != 0;
        float s = -INFINITY;
        if (ok) {
            const T* kr = kbase + (int64_t)t * 3 * D;
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < DH; ++d) a = fmaf(qv[d], to_f32(kr[d]), a);
            s = a;
        }
        float cm = s;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cm = fmaxf(cm, __shfl_xor(cm, o, 64));
        const float mn = fmaxf(m, cm);
        if (mn == -INFINITY) continue;                               // no valid key so far (wave-uniform)
        const float alpha = __expf(m - mn);                          // exp(-inf) = 0 on the first chunk with a valid key
        const float p = ok ? __expf(s - mn) : 0.f;
        float ps = p;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ps += __shfl_xor(ps, o, 64);
        l = l * alpha + ps;
        acc *= alpha;
        const int nk = S - t0 < 64 ? S - t0 : 64;
        for (int j = 0; j < nk; ++j) {
            const float pj = __shfl(p, j, 64);
            if (lane < DH) acc = fmaf(pj, to_f32(kbase[(int64_t)(t0 + j) * 3 * D + D + lane]), acc);
        }
        m = mn;
    }
    if (lane < DH) out[(int64_t)b * D + hd * DH + lane] = l > 0.f ? acc / l : 0.f;   // a sequence without any valid key gives zeros
}

int launch_attn_cls(const float* q, const void* qkv, int qkv_bf16, const int32_t* mask, float* out, int B, int S, int H, int dh, hipStream_t st) {
    const float scale = 1.0f / sqrtf((float)dh);
    const dim3 grid(B * H), block(64);
    if (qkv_bf16) {
        MGEA_REQUIRE(dh == 64, MGEA_EINVAL, "attn_cls: bf16 keys need head_dim 64");
        hipLaunchKernelGGL((attn_cls_kernel<__bf16, 64>), grid, block, 0, st, q, (const __bf16*)qkv, mask, out, S, H, scale);
    } else if (dh == 64) {
        hipLaunchKernelGGL((attn_cls_kernel<float, 64>), grid, block, 0, st, q, (const float*)qkv, mask, out, S, H, scale);
    } else {
        MGEA_REQUIRE(dh == 32, MGEA_EINVAL, "attn_cls: head_dim %d not built (32 or 64)", dh);
        hipLaunchKernelGGL((attn_cls_kernel<float, 32>), grid, block, 0, st, q, (const float*)qkv, mask, out, S, H, scale);
    }
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

}  // namespace mgea
