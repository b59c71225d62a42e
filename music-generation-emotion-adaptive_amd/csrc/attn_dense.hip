// Dense non-causal attention over a packed qkv buffer [B*T, 3C] (q | k | v), exact-fp32 MFMA.
// Used for the decoder prefill without past (the reference never masks: api_cache.py:68, so the
// prompt is attended bidirectionally) and for DistilBERT's self-attention with its additive
// padding mask (emotion_analysis/inference.py:16-17 -> transformers MultiHeadSelfAttention).
//
// Flash-style: one workgroup = 64 queries of one (b, h) (4 waves x 16 queries), looping over
// 64-key tiles staged in LDS (K and V, 16-B chunks XOR-swizzled by key&7 -> conflict-free
// ds_read_b128 fragments and ds_read_b32 V^T reads); the S x S score matrix never exists.
//   S^T = K Q^T   ("swapped" product): the accumulator of v_mfma_f32_16x16x4_f32 then has the
//         QUERY on lane&15 and 4 consecutive KEYS in its 4 registers, so (a) the row softmax is
//         in-lane + two xor-shuffles, and (b) P^T is already the B operand of the next product;
//   O^T = V^T P^T: k-step s of the MFMA takes key 4g+s from register s of the S^T accumulator and
//         V[key 4g+s][d] from LDS; O^T leaves d in registers -> 16-byte output stores.
// Keys are valid when (t < lens[b] if lens) && (mask[b,t] != 0 if mask); invalid keys get p = 0
// exactly (HF adds finfo.min before softmax: same result unless a row has no valid key).
#include "common.h"

namespace mgea {

template <int DH>
__global__ __launch_bounds__(256) void attn_dense_kernel(const float* __restrict__ qkv, const int32_t* __restrict__ lens,
                                                        const int32_t* __restrict__ mask, float* __restrict__ out,
                                                        int T_pad, int H, float scale, int tiled_out, const int32_t* __restrict__ cu) {
    constexpr int NCH = DH / 4;   // 16-B chunks per row
    constexpr int DC = DH / 16;   // 16-wide d tiles
    constexpr int F4 = 64 * NCH / 256;
    __shared__ float4 sK[64 * NCH];
    __shared__ float4 sV[64 * NCH];
    __shared__ int sValid[64];

    const int C = H * DH;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    // packed rows (cu != NULL, DistilBERT on unpadded batches): sequence b = rows cu[b] .. cu[b + 1] - 1, all real; the grid is sized
    // for the longest sequence, blocks past this one's end leave at once and its key loop stops at its own length
    const int64_t row0 = cu ? (int64_t)cu[b] : (int64_t)b * T_pad;
    const int T = cu ? cu[b + 1] - cu[b] : T_pad;
    if (q0 >= T) return;
    const int nvalid = lens ? lens[b] : T;

    // Q fragments (pre-scaled): query q0 + wave*16 + c, d = dc*16 + 4g .. +3
    int qrow = q0 + wave * 16 + c;
    const bool q_in = qrow < T;
    qrow = q_in ? qrow : T - 1;
    float4 qf[DC];
#pragma unroll
    for (int dc = 0; dc < DC; ++dc) {
        float4 v = ld4(qkv + (row0 + qrow) * 3 * C + h * DH + dc * 16 + 4 * g);
        qf[dc] = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
    }

    f32x4 oacc[DC];
#pragma unroll
    for (int dt = 0; dt < DC; ++dt) oacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float mx = -INFINITY, lsum = 0.f;

    const int ntiles = (T + 63) >> 6;
    for (int kt0 = 0; kt0 < ntiles; ++kt0) {
        const int k0 = kt0 * 64;
        __syncthreads();  // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < F4; ++i) {
            const int idx = tid + i * 256, key = idx / NCH, ch = idx % NCH;
            int kr = k0 + key;
            kr = kr < T ? kr : T - 1;
            const float* src = qkv + (row0 + kr) * 3 * C + h * DH + ch * 4;
            const int dst = key * NCH + ((ch & ~7) | ((ch ^ key) & 7));
            sK[dst] = ld4(src + C);
            sV[dst] = ld4(src + 2 * C);
        }
        if (tid < 64) {
            const int kidx = k0 + tid;
            bool ok = kidx < T && kidx < nvalid;
            if (ok && mask) ok = mask[row0 + kidx] != 0;   // (padded rows only: row0 = b * T_pad)
            sValid[tid] = ok ? 1 : 0;
        }
        __syncthreads();

        // ---- S^T tiles: 4 x (16 keys x 16 queries)
        f32x4 sc[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            sc[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            const int key = kt * 16 + c;
#pragma unroll
            for (int dc = 0; dc < DC; ++dc) {
                const int ch = dc * 4 + g;
                const float4 kf = sK[key * NCH + ((ch & ~7) | ((ch ^ key) & 7))];
                sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[dc].x, sc[kt], 0, 0, 0);
                sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[dc].y, sc[kt], 0, 0, 0);
                sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[dc].z, sc[kt], 0, 0, 0);
                sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[dc].w, sc[kt], 0, 0, 0);
            }
        }
        // ---- mask + online softmax for this lane's query (keys kt*16 + 4g + r)
        float tmax = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int4 vl = *reinterpret_cast<const int4*>(&sValid[kt * 16 + 4 * g]);
            sc[kt][0] = vl.x ? sc[kt][0] : -INFINITY;
            sc[kt][1] = vl.y ? sc[kt][1] : -INFINITY;
            sc[kt][2] = vl.z ? sc[kt][2] : -INFINITY;
            sc[kt][3] = vl.w ? sc[kt][3] : -INFINITY;
            tmax = fmaxf(tmax, fmaxf(fmaxf(sc[kt][0], sc[kt][1]), fmaxf(sc[kt][2], sc[kt][3])));
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mnew = fmaxf(mx, tmax);
        const float msafe = (mnew == -INFINITY) ? 0.f : mnew;   // no valid key seen yet
        const float alpha = (mx == -INFINITY) ? 0.f : __expf(mx - msafe);
        float psum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(sc[kt][r] - msafe);   // exp(-inf) = 0 for masked keys
                sc[kt][r] = p;
                psum += p;
            }
        lsum = lsum * alpha + psum;
        mx = mnew;
#pragma unroll
        for (int dt = 0; dt < DC; ++dt) {
            oacc[dt][0] *= alpha; oacc[dt][1] *= alpha; oacc[dt][2] *= alpha; oacc[dt][3] *= alpha;
        }
        // ---- O^T += V^T P^T
        const float* sVf = reinterpret_cast<const float*>(sV);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int key = kt * 16 + 4 * g + s;
#pragma unroll
                for (int dt = 0; dt < DC; ++dt) {
                    const int ch = dt * 4 + (c >> 2);
                    const float vv = sVf[(key * NCH + ((ch & ~7) | ((ch ^ key) & 7))) * 4 + (c & 3)];
                    oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv, sc[kt][s], oacc[dt], 0, 0, 0);
                }
            }
    }
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    if (q_in) {
        const float inv = 1.0f / lsum;
#pragma unroll
        for (int dt = 0; dt < DC; ++dt) {
            const int n = h * DH + dt * 16 + 4 * g;
            float* dst = tiled_out ? out + tiled_off((int)(row0 + qrow), n, C) : out + (row0 + qrow) * C + n;
            st4(dst, make_float4(oacc[dt][0] * inv, oacc[dt][1] * inv, oacc[dt][2] * inv, oacc[dt][3] * inv));
        }
    }
}

int launch_attn_dense(const float* qkv, const int32_t* lens, const int32_t* mask, float* out, int B, int T, int H,
                      int dh, int tiled_out, hipStream_t st, const int32_t* cu) {
    MGEA_REQUIRE(!(cu && (lens || mask || tiled_out)), MGEA_EINVAL, "attention: packed rows carry no lengths / mask");
    MGEA_REQUIRE(B > 0 && T > 0 && B <= 65535 && H <= 65535, MGEA_EINVAL, "attention: bad shape B=%d T=%d H=%d", B, T, H);
    const float scale = 1.0f / sqrtf((float)dh);
    dim3 grid(ceil_div(T, 64), H, B);
    switch (dh) {
        case 32: hipLaunchKernelGGL(attn_dense_kernel<32>, grid, dim3(256), 0, st, qkv, lens, mask, out, T, H, scale, tiled_out, cu); break;
        case 64: hipLaunchKernelGGL(attn_dense_kernel<64>, grid, dim3(256), 0, st, qkv, lens, mask, out, T, H, scale, tiled_out, cu); break;
        case 96: hipLaunchKernelGGL(attn_dense_kernel<96>, grid, dim3(256), 0, st, qkv, lens, mask, out, T, H, scale, tiled_out, cu); break;
        default:
            MGEA_REQUIRE(false, MGEA_EINVAL, "attention: head_dim %d not supported (32, 64, 96)", dh);
    }
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

}  // namespace mgea
