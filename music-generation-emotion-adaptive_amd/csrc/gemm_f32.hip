// Exact-fp32 MFMA GEMM for gfx950:  P[z] = A[M, Kz] @ W[N, Kz]^T   (Linear layers, torch [out,in] weights)
//
// Used for every dense contraction on the path (fused QKV in_proj, out_proj, MLP, LM head;
// reference: nn.MultiheadAttention in_proj/out_proj api_cache.py:43,68, mlp api_cache.py:45-49,
// head api_cache.py:85,105; DistilBERT q/k/v/out_lin, ffn.lin1/lin2, pre_classifier, classifier).
//
// * v_mfma_f32_16x16x4_f32: bit-for-bit an fp32 fma chain (no xf32 on gfx950), which is what the
//   1e-3-logit / bit-exact-greedy parity bar needs.  The MFMA is issued "swapped"
//   (D = Wfrag x Afrag) so each lane ends up with 4 CONSECUTIVE output columns -> 16-byte stores.
// * BK = 32 floats = one full 128-B line per row per k-tile; tiles live in LDS as [row][8 x 16 B]
//   with the 16-B chunk index XOR-swizzled by (row & 7): the staging ds_write_b128 (8 lanes = one
//   row) and the fragment ds_read_b128 (16 rows x 4 k-chunks) are both bank-conflict free.
// * One float4 LDS read feeds four MFMA k-steps: lane (c = lane&15, g = lane>>4) reads
//   row c, k = kk + 4g .. 4g+3, and step s of the MFMA consumes k = kk + 4g + s from BOTH
//   operands -- any k permutation is legal as long as A and W agree.
// * Register-staged double buffering: global loads for tile t+1 are issued before the MFMAs of
//   tile t and written to the other LDS buffer after them; one barrier per k-tile.
// * (Round 4, measured and removed: a persistent form -- two workgroups per CU walking tile lists, the next tile's first k-tile requested
//   under the last MFMAs and the epilogue, bitwise equal -- ran 2-9 % SLOWER than one workgroup per tile on every big shape (K = 512:
//   0.66 vs 0.67 of the fp32 matrix peak, K = 2048: 0.75 vs 0.83), after two rounds of fighting hipcc's copies of in-flight staging
//   registers; a second set of staging registers (two k-tiles of loads in flight) spilled at two waves per SIMD; a single LDS buffer
//   with two barriers per k-tile (32 KB: three workgroups per CU) made the [64, 1024] cache fill 26.39 ms against 25.95; both k-halves'
//   fragment reads requested before the first MFMA (a scheduling fence keeps them there) 26.77 ms.  The library
//   reaches 0.83-0.94 on these shapes: profiles/r4_gemm_vendor_compare.txt.)
// * split-K over gridDim.z writes raw partial slabs; the row epilogues (rowops.hip) sum the slabs
//   in a fixed order, so results are run-to-run deterministic (no float atomics).
#include "common.h"

namespace mgea {

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void gemm_f32_nt_kernel(const float* __restrict__ A, int lda,
                                                         const float* __restrict__ W, int ldw,
                                                         float* __restrict__ P, int M, int N, int K, int ldp,
                                                         int kt_per_split, int tiles_n, const float* __restrict__ bias,
                                                         int act) {
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int MT = WM / 16, NT = WN / 16;
    constexpr int A_F4 = BM * 8 / 256, W_F4 = BN * 8 / 256;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per block");
    static_assert(A_F4 >= 1 && W_F4 >= 1, "tile too small for 256 threads");
    __shared__ float4 lds[2][(BM + BN) * 8];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // XCD-aware tile order: consecutive block ids round-robin over the 8 XCDs, so give each XCD
    // a contiguous run of tiles that share W column panels (speed only, never correctness).
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int KT = K / 32;
    const int kt0 = blockIdx.z * kt_per_split;
    int kt1 = kt0 + kt_per_split;
    if (kt1 > KT) kt1 = KT;
    const int nt = kt1 - kt0;

    f32x4 acc[NT][MT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 ra[A_F4], rw[W_F4];
    auto load_tile = [&](int kt) {
        const int k0 = kt * 32;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int idx = tid + i * 256, row = idx >> 3, ch = idx & 7;
            int r = m0 + row;
            r = r < M ? r : M - 1;
            ra[i] = ld4(A + (int64_t)r * lda + k0 + ch * 4);
        }
#pragma unroll
        for (int i = 0; i < W_F4; ++i) {
            const int idx = tid + i * 256, row = idx >> 3, ch = idx & 7;
            int r = n0 + row;
            r = r < N ? r : N - 1;
            rw[i] = ld4(W + (int64_t)r * ldw + k0 + ch * 4);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int idx = tid + i * 256, row = idx >> 3, ch = idx & 7;
            lds[buf][row * 8 + (ch ^ (row & 7))] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < W_F4; ++i) {
            const int idx = tid + i * 256, row = idx >> 3, ch = idx & 7;
            lds[buf][(BM + row) * 8 + (ch ^ (row & 7))] = rw[i];
        }
    };

    if (nt > 0) {
        load_tile(kt0);
        store_tile(0);
    }
    __syncthreads();
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) load_tile(kt0 + t + 1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            float4 af[MT], wf[NT];
#pragma unroll
            for (int m = 0; m < MT; ++m) af[m] = lds[buf][(wm * WM + m * 16 + c) * 8 + ((kk * 4 + g) ^ (c & 7))];
#pragma unroll
            for (int n = 0; n < NT; ++n)
                wf[n] = lds[buf][(BM + wn * WN + n * 16 + c) * 8 + ((kk * 4 + g) ^ (c & 7))];
            // k-step outermost: consecutive MFMAs write DIFFERENT accumulators (a dependent 16x16x4 pair issues 36 clocks apart, an
            // independent one 32: tools/micro/mfma_rate.hip).  Every accumulator still sees its k-steps in the order x, y, z, w.
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const float wv = ks == 0 ? wf[n].x : ks == 1 ? wf[n].y : ks == 2 ? wf[n].z : wf[n].w;
                        const float av = ks == 0 ? af[m].x : ks == 1 ? af[m].y : ks == 2 ? af[m].z : af[m].w;
                        acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, av, acc[n][m], 0, 0, 0);
                    }
        }
        if (t + 1 < nt) store_tile(buf ^ 1);
        __syncthreads();
    }

    // D[i][j]: i = W row (output column) = 4g + reg, j = A row (output row) = c
    float* Pz = P + (int64_t)blockIdx.z * M * ldp;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row = m0 + wm * WM + m * 16 + c;
        if (row < M) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int col = n0 + wn * WN + n * 16 + 4 * g;
                if (col < ldp) {  // ldp = round_up(N, 64) for slabs: whole 16-B groups are in or out
                    float4 v = make_float4(acc[n][m][0], acc[n][m][1], acc[n][m][2], acc[n][m][3]);
                    if (bias) {   // direct epilogue (no split-K): P is the final [M, ldp] output, N % 4 == 0
                        if (col >= N) continue;
                        v = add4(v, ld4(bias + col));
                        if (act == ACT_GELU) v = make_float4(gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w));
                        if (act == ACT_RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                    }
                    st4(Pz + (int64_t)row * ldp + col, v);
                }
            }
        }
    }
}

int pick_split_k(int M, int N, int K, int64_t cap_floats) {
    const int KT = K / 32;
    int s = 1;
    if (M > 64) {
        // 128 x 128 tiles: a mid-size M (a single prompt of 128 tokens, a short prefill) is only a handful of tiles --
        // split K until the grid covers the chip (a [128, 768] x [768, 768] GEMM is 6 workgroups otherwise)
        const int tiles = ceil_div(M, 128) * ceil_div(N, 128);
        while (tiles * s < 256 && s * 2 <= KT && s < 16) s *= 2;
    } else {
        // Skinny (decode) GEMMs: one 64-row tile; split K until ~2 blocks per CU are available.
        const int tiles = ceil_div(N, 64);
        while (tiles * s < 384 && s * 2 <= KT && s < 32) s *= 2;
    }
    if (cap_floats >= 0)
        while (s > 1 && (int64_t)s * slab_floats(M, N) > cap_floats) s /= 2;   // the caller's slab workspace
    return s;
}

// enough 128 x 128 tiles for the chip without split-K: the bias / activation epilogue can go inside the GEMM
bool gemm_direct_epilogue_ok(int M, int N) { return M > 64 && ceil_div(M, 128) * ceil_div(N, 128) >= 128; }

// Large-M GEMM with the bias / activation epilogue inside the kernel: out[M, ldo] = act(A W^T + bias).
// Saves the slab write + read of the two-kernel form (one HBM round trip of M x N floats).
int launch_gemm_f32_bias_act(const float* A, int lda, const float* W, int ldw, const float* bias, float* out, int ldo,
                             int M, int N, int K, int act, hipStream_t st) {
    MGEA_REQUIRE(M > 64 && N % 4 == 0 && ldo % 4 == 0 && ldo >= N && K % 32 == 0 && bias, MGEA_EINVAL,
                 "gemm_bias_act: bad shape M=%d N=%d K=%d ldo=%d", M, N, K, ldo);
    const int tm = ceil_div(M, 128), tn = ceil_div(N, 128);
    hipLaunchKernelGGL((gemm_f32_nt_kernel<128, 128, 2, 2>), dim3(tm * tn, 1, 1), dim3(256), 0, st, A, lda, W, ldw, out, M, N, K,
                       ldo, K / 32, tn, bias, act);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

int launch_gemm_f32(const float* A, int lda, const float* W, int ldw, float* P, int M, int N, int K,
                    int split_k, hipStream_t st) {
    MGEA_REQUIRE(M > 0 && N > 0 && K > 0 && (K % 32) == 0, MGEA_EINVAL, "gemm: bad shape M=%d N=%d K=%d (K %% 32 != 0?)", M, N, K);
    MGEA_REQUIRE((lda % 4) == 0 && (ldw % 4) == 0, MGEA_EINVAL, "gemm: leading dims must be multiples of 4");
    const int KT = K / 32;
    if (split_k < 1) split_k = 1;
    if (split_k > KT) split_k = KT;
    const int kt_per = ceil_div(KT, split_k);
    split_k = ceil_div(KT, kt_per);  // no empty slabs
    const int ldp = (int)slab_ld(N);
    if (M > 64) {
        const int tm = ceil_div(M, 128), tn = ceil_div(N, 128);
        dim3 grid(tm * tn, 1, split_k);
        hipLaunchKernelGGL((gemm_f32_nt_kernel<128, 128, 2, 2>), grid, dim3(256), 0, st, A, lda, W, ldw, P, M, N, K,
                           ldp, kt_per, tn, (const float*)nullptr, 0);
    } else {
        const int tn = ceil_div(N, 64);
        dim3 grid(tn, 1, split_k);
        hipLaunchKernelGGL((gemm_f32_nt_kernel<64, 64, 1, 4>), grid, dim3(256), 0, st, A, lda, W, ldw, P, M, N, K,
                           ldp, kt_per, tn, (const float*)nullptr, 0);
    }
    MGEA_CHECK_HIP(hipGetLastError());
    return split_k;  // > 0: number of slabs written
}

}  // namespace mgea
