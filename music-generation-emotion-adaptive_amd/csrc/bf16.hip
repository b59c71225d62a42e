// bf16 "perf mode" kernels for the prefill-only DistilBERT path (BASELINE.json configs[1]: bf16,
// B=256, S=128): bf16 storage for weights and activations, fp32 accumulation in the MFMAs, fp32
// LayerNorm / softmax / GELU math.  Parity mode stays fp32 (gemm_f32.hip, attn_dense.hip); these
// kernels are validated against the oracle run on bf16-rounded weights (tests/test_gpu_bf16.py).
//
// * gemm_bf16_nt_kernel: C[M,N] = epi(A[M,K] @ W[N,K]^T + bias), v_mfma_f32_16x16x32_bf16,
//   128x128x64 tiles, 4 waves (2x2, 64x64 per wave), LDS tiles [row][8 x 16 B] with the 16-B chunk
//   XOR-swizzled by row&7 (conflict-free ds_write_b128 staging and ds_read_b128 fragments),
//   register-staged double buffering (next tile's global loads fly under this tile's MFMAs), XCD-aware
//   tile order, "swapped" MFMA so a lane owns 4 consecutive output columns (8-byte bf16 stores).
// * attn_bf16_kernel: persistent, LDS-DMA double-buffered flash attention over a packed bf16 qkv buffer with an optional key
//   mask; S^T = K Q^T keeps the query on lane&15, so the softmax is in-lane + two permlane swaps and the fp32 accumulators,
//   converted to bf16, ARE the B operand of O^T = V^T P^T (k-permutation shared by V^T, which is read with ds_read_b64_tr_b16).
// * layernorm / embedding kernels: bf16 in/out, fp32 statistics.
#include <stdlib.h>

#include "common.h"

namespace mgea {

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void f32_to_bf16_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t n) {
    typedef T t4 __attribute__((ext_vector_type(4)));
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 v = ld4(src + i);
        t4 o = {(T)v.x, (T)v.y, (T)v.z, (T)v.w};
        *reinterpret_cast<t4*>(dst + i) = o;
    } else {
        for (int64_t j = i; j < n; ++j) dst[j] = (T)src[j];
    }
}
int launch_f32_to_bf16(const float* src, void* dst, int64_t n, hipStream_t st, int f16) {
    if (f16) hipLaunchKernelGGL(f32_to_bf16_kernel<_Float16>, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, st, src, (_Float16*)dst, n);
    else     hipLaunchKernelGGL(f32_to_bf16_kernel<bf16_t>, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, st, src, (bf16_t*)dst, n);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

// ------------------------------------------------------------------------------------------
enum { BEPI_BIAS = 0, BEPI_BIAS_GELU = 1, BEPI_BIAS_RES = 2,
       // LayerNorm folded around the GEMMs (persistent 256 x 256 kernel only; see BfEpiLn in common.h):
       BEPI_LNFOLD = 3,        // out = rstd_row (A W'^T - mean_row c1) + c2         (A = raw rows, W' = W diag(gamma), c2 in the bias slot)
       BEPI_LNFOLD_GELU = 4,   // ... then GELU
       BEPI_RES_LN = 5,        // out = A W^T + bias + LN(res rows) (LayerNorm of the residual applied on the way in), + row statistics of out
       BEPI_BIAS_F32 = 6 };    // out = A W^T + bias written as fp32 [M, ldc floats] (the decoder's LM head: N = vocab need only be a multiple of 4)

// erf-GELU for bf16 outputs with ONE transcendental.  With a = |x|: gelu(x) = max(x, 0) - 0.5 a erfc(a / sqrt 2), and
// erfc(a / sqrt 2) = 2^q(a) where q = log2(erfcx) - (a^2 / 2) log2 e is smooth: a degree-5 polynomial (weighted least squares on
// [0, 6.5], fitted and checked in fp32 against scipy's erf: |gelu error| < 1.5e-5 on [-12, 12], far below the 2^-9 rounding of the
// bf16 result; beyond 6.5 the clamped tail is < 2^-30).  11 instructions per element, v_exp_f32 the only quarter-rate one.  History:
// A&S 7.1.26 with a full-precision division and libm-style exp cost ~40 (14 us of a 40 us FC1 tile round); A&S 7.1.25 on
// v_rcp_f32 + v_exp_f32 12 with two transcendentals, still 28 of FC1's 169 us (measured by switching GELU off).
#define MGEA_GELU_D0 -1.585257735e-05f
#define MGEA_GELU_D1 -1.150631181e+00f
#define MGEA_GELU_D2 -4.612153958e-01f
#define MGEA_GELU_D3 -4.992833406e-02f
#define MGEA_GELU_D4 6.105210696e-03f
#define MGEA_GELU_D5 -3.249367875e-04f
#define MGEA_GELU_AMAX 6.505382387f
__device__ __forceinline__ float gelu_fast(float x) {
    const float a = fminf(fabsf(x), MGEA_GELU_AMAX);
    float q = fmaf(MGEA_GELU_D5, a, MGEA_GELU_D4);
    q = fmaf(q, a, MGEA_GELU_D3);
    q = fmaf(q, a, MGEA_GELU_D2);
    q = fmaf(q, a, MGEA_GELU_D1);
    q = fmaf(q, a, MGEA_GELU_D0);
    return fmaf(-0.5f * a, __builtin_amdgcn_exp2f(q), fmaxf(x, 0.f));
}
// Two elements with packed fp32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32: two results per instruction).  For epilogues only --
// beside MFMAs packed fp32 is slower than scalar (MI355X_MICROARCH.md).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
    const f32x2 a = {fminf(fabsf(x[0]), MGEA_GELU_AMAX), fminf(fabsf(x[1]), MGEA_GELU_AMAX)};
    f32x2 q = __builtin_elementwise_fma((f32x2){MGEA_GELU_D5, MGEA_GELU_D5}, a, (f32x2){MGEA_GELU_D4, MGEA_GELU_D4});
    q = __builtin_elementwise_fma(q, a, (f32x2){MGEA_GELU_D3, MGEA_GELU_D3});
    q = __builtin_elementwise_fma(q, a, (f32x2){MGEA_GELU_D2, MGEA_GELU_D2});
    q = __builtin_elementwise_fma(q, a, (f32x2){MGEA_GELU_D1, MGEA_GELU_D1});
    q = __builtin_elementwise_fma(q, a, (f32x2){MGEA_GELU_D0, MGEA_GELU_D0});
    const f32x2 e = {__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
    const f32x2 pos = {fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
    return __builtin_elementwise_fma(a * -0.5f, e, pos);
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_nt_kernel(const bf16_t* __restrict__ A, int lda,
                                                          const bf16_t* __restrict__ W, int ldw,
                                                          const float* __restrict__ bias,
                                                          const bf16_t* __restrict__ res, bf16_t* __restrict__ C,
                                                          int ldc, int M, int N, int K, int tiles_n) {
    constexpr int BM = 128, BN = 128;
    __shared__ float4 lds[2][(BM + BN) * 8];  // 16-B chunks: [buf][row*8 + (chunk ^ (row&7))]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);  // XCD-contiguous tile runs
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
    const int KT = K >> 6;

    f32x4 acc[4][4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 ra[4], rw[4];
    auto load_tile = [&](int kt) {
        const int k0 = kt * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * 256, row = idx >> 3, ch = idx & 7;
            int r = m0 + row; r = r < M ? r : M - 1;
            ra[i] = *reinterpret_cast<const float4*>(A + (int64_t)r * lda + k0 + ch * 8);
            int q = n0 + row; q = q < N ? q : N - 1;
            rw[i] = *reinterpret_cast<const float4*>(W + (int64_t)q * ldw + k0 + ch * 8);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * 256, row = idx >> 3, ch = idx & 7;
            lds[buf][row * 8 + (ch ^ (row & 7))] = ra[i];
            lds[buf][(BM + row) * 8 + (ch ^ (row & 7))] = rw[i];
        }
    };
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < KT; ++t) {
        const int buf = t & 1;
        if (t + 1 < KT) load_tile(t + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[4], wf[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float4 v = lds[buf][(wm * 64 + m * 16 + c) * 8 + ((ks * 4 + g) ^ (c & 7))];
                af[m] = *reinterpret_cast<const bf16x8*>(&v);
            }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const float4 v = lds[buf][(BM + wn * 64 + n * 16 + c) * 8 + ((ks * 4 + g) ^ (c & 7))];
                wf[n] = *reinterpret_cast<const bf16x8*>(&v);
            }
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n], af[m], acc[n][m], 0, 0, 0);
        }
        if (t + 1 < KT) store_tile(buf ^ 1);
        __syncthreads();
    }
    // D[i = output column 4g + r][j = output row c]
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int row = m0 + wm * 64 + m * 16 + c;
        if (row >= M) continue;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int col = n0 + wn * 64 + n * 16 + 4 * g;
            if (col >= N) continue;
            float4 v = make_float4(acc[n][m][0], acc[n][m][1], acc[n][m][2], acc[n][m][3]);
            if (bias) v = add4(v, ld4(bias + col));
            if (EPI == BEPI_BIAS_GELU) v = make_float4(gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w));
            if (EPI == BEPI_BIAS_RES) {
                const bf16x4 r4 = *reinterpret_cast<const bf16x4*>(res + (int64_t)row * ldc + col);
                v = add4(v, make_float4((float)r4[0], (float)r4[1], (float)r4[2], (float)r4[3]));
            }
            bf16x4 o = {(bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
            *reinterpret_cast<bf16x4*>(C + (int64_t)row * ldc + col) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Large-M variant: 256 x (64*NT) tile, 8 waves (2 x 4, 128 x 16*NT per wave), BK = 64, operands go
// HBM/L2 -> LDS directly with global_load_lds_dwordx4 (no staging registers, no ds_write pass).
// The LDS image is lane-linear per wave instruction (8 rows x 128 B), so the XOR swizzle is applied
// on the SOURCE address: the lane that fills LDS chunk (row, ch') fetches global chunk ch' ^ (row&7),
// and the fragment reads use the same involution.  Two LDS buffers: the loads of tile t+1 are issued
// before the MFMAs of tile t and drained (vmcnt(0)) just before the barrier that ends it.
// WMW = waves along M (each 128 rows): WMW = 2 -> 256-row tile, 8 waves, one workgroup per CU;
// WMW = 1 -> 128-row tile, 4 waves, 64 KB of LDS, two workgroups per CU whose barriers interleave.
template <int EPI, int NT, int WMW>
__global__ __launch_bounds__(256 * WMW) void gemm_bf16_glds_kernel(const bf16_t* __restrict__ A, int lda,
                                                            const bf16_t* __restrict__ W, int ldw,
                                                            const float* __restrict__ bias,
                                                            const bf16_t* __restrict__ res, bf16_t* __restrict__ C,
                                                            int ldc, int M, int N, int K, int tiles_n) {
    constexpr int BM = 128 * WMW, BN = 64 * NT, NTHR = 256 * WMW;
    constexpr int WPW = (BN / 8) / (4 * WMW);           // W pieces (8 rows x 128 B) per wave per tile
    constexpr int SA = BM * 8, STAGE = (BM + BN) * 8;   // in 16-byte chunks
    extern __shared__ __attribute__((aligned(16))) float4 lds[];  // [2][STAGE]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int wm = wave >> 2, wn = wave & 3;
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
    const int KT = K >> 6;

    // per-lane source of the glds pieces: piece p covers rows 8p .. 8p+7; lane -> (row 8p + lane/8, chunk)
    const int lr = lane >> 3, lch = (lane & 7) ^ lr;
    const bf16_t* asrc[4];
    const bf16_t* wsrc[WPW];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int r = m0 + (wave * 4 + i) * 8 + lr;
        r = r < M ? r : M - 1;
        asrc[i] = A + (int64_t)r * lda + lch * 8;
    }
#pragma unroll
    for (int i = 0; i < WPW; ++i) {
        int r = n0 + (wave * WPW + i) * 8 + lr;
        r = r < N ? r : N - 1;
        wsrc[i] = W + (int64_t)r * ldw + lch * 8;
    }
    auto issue = [&](int buf, int kt) {
        const int k0 = kt * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + k0),
                                             (__attribute__((address_space(3))) void*)(lds + buf * STAGE + (wave * 4 + i) * 64),
                                             16, 0, 0);
#pragma unroll
        for (int i = 0; i < WPW; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + k0),
                                             (__attribute__((address_space(3))) void*)(lds + buf * STAGE + SA + (wave * WPW + i) * 64),
                                             16, 0, 0);
    };

    f32x4 acc[NT][8];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // NS-deep LDS ring (3 stages of 48 KB for the 256x128 tile, 2 of 64 KB for 256x256): NS-1 tiles are
    // in flight while one is consumed.  Counted vmcnt + raw s_barrier: a __syncthreads() would drain
    // the LDS-DMA queue (vmcnt(0)) at every tile.  RAW: each wave waits for ITS pieces of tile t+1,
    // then the barrier makes all pieces visible before iteration t+1 reads them.  WAR: the buffer
    // re-filled in iteration t was last read in iteration t-1, behind that iteration's barrier.
    constexpr int NS = (NT == 2 && WMW == 2) ? 3 : 2;
    constexpr int PER_TILE = 4 + WPW;             // glds instructions per wave per tile
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < KT) issue(s, s);
    if (KT > NS - 2 && NS > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PER_TILE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int cur = 0, nxt = NS - 1;                     // ring slots of tile t and of tile t + NS - 1
    for (int t = 0; t < KT; ++t) {
        const float4* sb = lds + cur * STAGE;
        const bool more = t + NS - 1 < KT;
        if (more) issue(nxt, t + NS - 1);
        cur = cur + 1 == NS ? 0 : cur + 1;
        nxt = nxt + 1 == NS ? 0 : nxt + 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 wf[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const float4 v = sb[SA + (wn * 16 * NT + n * 16 + c) * 8 + ((ks * 4 + g) ^ (c & 7))];
                wf[n] = *reinterpret_cast<const bf16x8*>(&v);
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const float4 v = sb[(wm * 128 + m * 16 + c) * 8 + ((ks * 4 + g) ^ (c & 7))];
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(&v);
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n], af, acc[n][m], 0, 0, 0);
            }
        }
        // tile t+1 must have landed; the tiles issued after it (NS-2 of them) may stay in flight
        if (more && NS > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PER_TILE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    // Epilogue through LDS (the operand ring is free now): the MFMA layout gives each lane 4 columns
    // of one row (8-byte pieces, 16 rows per instruction); staged as a [256][BN] bf16 tile and
    // streamed out as whole rows, 16 bytes per lane -> full-line stores and residual reads.
    constexpr int PITCH = BN * 2 + 16;            // bytes; +16 keeps 16-B alignment and staggers banks
    unsigned char* sC = reinterpret_cast<unsigned char*>(lds);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                 // every wave is done reading the last operand tile
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int lrow = wm * 128 + m * 16 + c;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int lcol = wn * 16 * NT + n * 16 + 4 * g;
            float4 v = make_float4(acc[n][m][0], acc[n][m][1], acc[n][m][2], acc[n][m][3]);
            if (bias && n0 + lcol < N) v = add4(v, ld4(bias + n0 + lcol));
            if (EPI == BEPI_BIAS_GELU) v = make_float4(gelu_fast(v.x), gelu_fast(v.y), gelu_fast(v.z), gelu_fast(v.w));
            bf16x4 o = {(bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
            *reinterpret_cast<bf16x4*>(sC + lrow * PITCH + lcol * 2) = o;
        }
    }
    __syncthreads();
    constexpr int CPR = BN / 8;                   // 16-byte chunks per tile row
#pragma unroll
    for (int i = 0; i < BM * CPR / NTHR; ++i) {
        const int id = tid + i * NTHR, lrow = id / CPR, ch = id % CPR;
        const int row = m0 + lrow, col = n0 + ch * 8;
        if (row < M && col < N) {
            bf16x8 v = *reinterpret_cast<const bf16x8*>(sC + lrow * PITCH + ch * 16);
            if (EPI == BEPI_BIAS_RES) {
                const bf16x8 r8 = *reinterpret_cast<const bf16x8*>(res + (int64_t)row * ldc + col);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (bf16_t)((float)v[j] + (float)r8[j]);
            }
            *reinterpret_cast<bf16x8*>(C + (int64_t)row * ldc + col) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------
// 256 x 256 x 64 tile, 8 waves (2 along M x 4 along N, 128 x 64 of C per wave), the phase-interleaved structure of the
// CDNA4 GEMM playbook (cdna_hip_programming.md T3+T4, "The 256^2 8-phase template"), written out from its rules:
//
//   * a K-tile is consumed in 4 PHASES of 16 MFMAs (one 64 x 32 quadrant of the wave's C over the whole BK = 64).  A phase is
//     [L-segment: ds_read the fragments this phase's MFMAs need, issue 2 global_load_lds (this wave's share of one
//      128-row half-tile of a LATER K-tile), wait for the reads] s_barrier [C-segment: 16 MFMAs] s_barrier.
//   * waves 4-7 (the second wave of every SIMD) run ONE BARRIER LATER than waves 0-3: in every barrier-to-barrier interval
//     one wave of each SIMD is in its C-segment (matrix pipe) and the other in its L-segment (LDS + address VALU + DMA
//     issue) -- instead of both reading, then both fighting for the matrix pipe.
//   * operands go HBM/L2 -> LDS by LDS-DMA (no staging registers, no ds_write), two 64 KB stages; the loads of a tile are
//     issued 4-6 phases before its first read and are waited for with a COUNTED vmcnt once per K-tile (6 younger DMAs stay
//     in flight across it); raw s_barrier only (a __syncthreads() would drain the DMA queue).
//
// Hazards, by construction (u = K-tile, stage = u & 1, phases q = 4u + p; L(q) / C(q) of waves 0-3 are intervals 2q / 2q+1,
// of waves 4-7 intervals 2q+1 / 2q+2):
//   reads  : p0 reads B(j=0) + A(i=0), p1 B(j=1), p2 A(i=1), p3 nothing (B(j=0) stays in registers); every ds_read is
//            complete (lgkmcnt(0)) before the barrier that ends its L-segment.  So tile u's B halves are last read in
//            interval 2(4u+1)+1 and its A halves in interval 2(4u+2)+1.
//   WAR    : stage u & 1 is refilled with tile u+2: B-half 0 issued in phase 4u+2 (first interval 2(4u+2) > 2(4u+1)+1),
//            BOTH A halves in phase 4u+3 (2(4u+3) > 2(4u+2)+1), B-half 1 in phase 4(u+1).
//            (Round 2 issued A-half 1 in phase 4(u+1)+1, three phases before its first read: A is the operand that misses L2 -- an
//            A panel is shared by N / 256 tiles only, the W panel by every row tile of the XCD's run -- and in-kernel stamps
//            (tools/gemm_bf16_stamps.py) showed the K loop of the N = 768 GEMMs at 2.0 us per K-tile against 1.45 at N >= 2304.
//            The A halves now have the longest lookahead, 5 phases, the L2-resident B-half 1 the shortest, 4.)
//   RAW    : tile u+1 is first read in interval 2*4(u+1) = 8u+8.  Its last DMA (B-half 1) is issued in phase 4u; every wave
//            waits until ITS DMAs of tile u+1 have landed -- vmcnt(6): only the three half-tiles issued in phases 4u+2, 4u+3
//            may still be in flight -- at the end of interval 8u+7 (waves 0-3: end of C(4u+3); waves 4-7: end of L(4u+3)),
//            and the barrier that ends that interval publishes them to every wave.
// One 1 KiB LDS-DMA piece (64 lanes x 16 B, LDS destination = wave-uniform byte address + 16 * lane) issued from inline
// asm, so that hipcc does not see it: seen through the builtin, ROCm 7.2 puts an s_waitcnt vmcnt(0) in front of the next
// ds_read of every phase (it cannot tell which LDS bytes a pending DMA writes) and the pipeline drains.  Hidden, none of the
// compiler's waits refer to it; every wait for these pieces is one of the hand-counted vmcnt below (this kernel has no other
// vector-memory instruction between its prologue and its epilogue).  M0 is compiler-reserved: saved and restored in the same
// statement (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void glds16_hidden(const void* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}

// Same, with a wave-uniform 64-bit base in SGPRs and a 32-bit byte offset per lane (no 64-bit vector address arithmetic).
__device__ __forceinline__ void glds16_hidden_s(const void* sbase, unsigned voff, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}

typedef unsigned long long u64x1;

// The two 16-bit storage types of the perf modes (bf16: DistilBERT, MGEA_DTYPE_BF16; fp16: the decoder's big-batch prefill,
// MGEA_DTYPE_F16) behind one set of kernels: vector types, the 16 x 16 x 32 MFMA, and a packed pair (one dword) <-> two fp32.
template <typename T> struct X16;
template <> struct X16<__bf16> {
    static constexpr bool is_bf16 = true;
    typedef bf16x8 v8; typedef bf16x4 v4; typedef bf16x2 v2;
    static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x2 unpack(unsigned u) { return (f32x2){__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; }
};
template <> struct X16<_Float16> {
    static constexpr bool is_bf16 = false;
    typedef h16x8 v8; typedef h16x4 v4; typedef _Float16 v2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ f32x2 unpack(unsigned u) {
        const v2 h = __builtin_bit_cast(v2, u);
        return (f32x2){(float)h[0], (float)h[1]};
    }
};

// Sum over the 32 lanes of a half wave (lanes 32 h .. 32 h + 31), result in all of them: four DPP row rotations (VALU speed) give
// the 16-lane sums, one ds_bpermute adds the neighbouring row.  (Five __shfl_xor steps = five trips through the LDS pipeline per
// value: 160 of them per tile made the statistics cost more than the LayerNorm kernel they replace.)
__device__ __forceinline__ float half_wave_sum(float v) {
#define MGEA_ROR_ADD(n) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + (n), 0xf, 0xf, false))
    MGEA_ROR_ADD(1); MGEA_ROR_ADD(2); MGEA_ROR_ADD(4); MGEA_ROR_ADD(8);
#undef MGEA_ROR_ADD
    return v + __shfl_xor(v, 16, 64);
}

// In-kernel time stamps of the persistent GEMM (cdna_hip_programming.md section 7), compiled ONLY into the tools build
// (tools/build_stamps.sh: -DMGEA_PH_STAMPS, a separate .so that tools/gemm_bf16_stamps.py loads); the product library has none of it.
#ifdef MGEA_PH_STAMPS
__device__ unsigned long long* g_ph_stamps = nullptr;      // [workgroup][64] 100 MHz ticks, written by thread 0
__device__ unsigned long long* g_ph_cycles = nullptr;      // same slots, shader-clock cycles (s_memtime): cycles / ticks = the clock the loop held
#define PH_STAMP(i) do { if (g_ph_stamps && tid == 0 && (i) < 64) { g_ph_stamps[(size_t)blockIdx.x * 64 + (i)] = __builtin_amdgcn_s_memrealtime(); \
                                                                    if (g_ph_cycles) g_ph_cycles[(size_t)blockIdx.x * 64 + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)
extern "C" int mgea_dbg_set_ph_stamps(unsigned long long* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ph_stamps), &p, sizeof(p)); }
extern "C" int mgea_dbg_set_ph_cycles(unsigned long long* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ph_cycles), &p, sizeof(p)); }
// ablation (wrong results, tools-only build): every unit loads tile 0's operands -- 192 KB of A and W that never leave L2 -- to
// tell a K loop bound by operand DELIVERY (it speeds up) from one bound by instruction issue (it does not)
__device__ int g_ph_same_tile = 0;
extern "C" int mgea_dbg_set_ph_same_tile(int v) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ph_same_tile), &v, sizeof(v)); }
#define PH_TILE_FOR_LOADS(t) (g_ph_same_tile ? 0 : (t))
// more timing ablations of the whole-tile K loop (wrong results), chosen at COMPILE time (-DMGEA_PH_ABLATE=bits: a run-time test around
// the reads or the MFMAs changes the loop it is meant to measure -- it tripled it): bit 0 no LDS-DMA after the prologue, bit 1 no fragment
// reads, bit 2 no MFMAs
#ifndef MGEA_PH_ABLATE
#define MGEA_PH_ABLATE 0
#endif
#define PH_ABLATE_INIT
#define PH_ABLATE(bit) (((MGEA_PH_ABLATE) >> (bit)) & 1)
#else
#define PH_ABLATE_INIT
#define PH_ABLATE(bit) 0
#define PH_STAMP(i) do { } while (0)
#define PH_TILE_FOR_LOADS(t) (t)
#endif

template <int EPI, typename T, int PH>
__global__ __launch_bounds__(512) void gemm_bf16_ph_kernel(const T* __restrict__ A, int lda, const T* __restrict__ W, int ldw,
                                                          const float* __restrict__ bias, const T* __restrict__ res,
                                                          T* __restrict__ C, int ldc, int M, int N, int K, int tiles_n, int n_tiles,
                                                          int tail_mode, BfEpiLn ln) {
    typedef typename X16<T>::v8 v8;
    typedef typename X16<T>::v4 v4;
    typedef typename X16<T>::v2 v2;
    constexpr int BM = 256, BN = 256;
    constexpr bool LNF = EPI == BEPI_LNFOLD || EPI == BEPI_LNFOLD_GELU;
    constexpr bool RESV = EPI == BEPI_BIAS_RES || EPI == BEPI_RES_LN;
    constexpr int SA = BM * 8, STAGE = (BM + BN) * 8;     // in 16-byte chunks: A tile, then the W tile
    constexpr int CST = 2 * STAGE;                         // the epilogue's own 32 KB behind the two operand stages
    extern __shared__ __attribute__((aligned(16))) float4 lds[];   // [2][STAGE] + [2048] = 160 KB; all LDS in this one array
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int wm = wave >> 2, wn = wave & 3;               // wm = 1: the late group
    const int KT = K >> 6;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float4*)lds;   // LDS byte address of the ring
    PH_ABLATE_INIT

    // PERSISTENT: one workgroup per CU walks its output tiles.  Workgroups b and b + 8 share an XCD (round-robin dispatch, speed
    // only): XCD r = b % 8 gets the contiguous run of tiles [r * n_tiles / 8, (r + 1) * n_tiles / 8) so that the tiles in flight on
    // one L2 share A and W panels; the run is walked by the XCD's workgroups with stride gridDim.x / 8.
    const int wg_x = gridDim.x >> 3;                         // host: gridDim.x % 8 == 0
    const int slot = (int)blockIdx.x >> 3;
    const int run0 = ((int)blockIdx.x & 7) * ((n_tiles + 7) >> 3);
    const int per_x = (run0 + ((n_tiles + 7) >> 3) < n_tiles ? ((n_tiles + 7) >> 3) : n_tiles - run0);   // tiles in this XCD's run (<= 0: none)
    // HALF-TILE TAIL: 384 tiles on 256 workgroups are one full round and a half-empty one.  When the tiles left over after the full
    // rounds number at most half the XCD's workgroups, each of them is cut into two 128-row halves computed by two workgroups
    // INDEPENDENTLY (no exchange, no scratch, no co-residency assumption -- round 2 cut K instead and swapped fp32 partial sums
    // through a coherent scratch, which two such launches sharing the GPU could deadlock on): a half unit keeps the 256-wide W tile
    // and the whole K loop, its 8 waves own 64 x 64 of C each, in a K loop of its own (3 LDS stages of 48 KB, 2 phases per K-tile:
    // "HALF UNITS" below).  Every output element sees the same MFMA sequence over K as in a whole tile: the two schedules are
    // bitwise identical.  tail_mode 2 additionally lets the odd slots run their half unit FIRST, so that the two halves of the chip
    // reach their store bursts half a tile apart.
    const int full_rounds = per_x > 0 ? per_x / wg_x : 0, rem = per_x > 0 ? per_x - full_rounds * wg_x : 0;
    const bool split = (tail_mode & 3) != 0 && rem > 0 && 2 * rem <= wg_x;
    int u_tile = 0, u_mode = 0;                              // mode 0: whole tile, 1 / 2: rows 0..127 / 128..255 of a shared tile
    // (tail_mode bit 2 = walk the XCD's run of tiles from its END: a GEMM whose A operand is the previous kernel's output and larger
    // than what the 256 MB Infinity Cache keeps of it -- DistilBERT's FC2 reads FC1's 201 MB -- otherwise reads that operand in the
    // order it was written, the one order in which an LRU cache that has just lost the head of the stream misses on every line)
    const bool rev = (tail_mode & 4) != 0;
    auto run_tile = [&](int idx) { return run0 + (rev ? per_x - 1 - idx : idx); };
    // A lower half (mode 2: rows 128..255 of its tile) that starts at or beyond row M does not exist: with M % 256 in (0, 128] the
    // last row tile has no second half, and its set_tile() clamp `M - 1 - lm0` would be negative -- as an unsigned LDS-DMA offset
    // 4 GB past the end of A (found by reading; e.g. DistilBERT at B = 255, S = 128).  Such a workgroup has no tail unit.
    const bool my_half = split && slot < 2 * rem &&
                         !((slot & 1) && (run_tile(full_rounds * wg_x + (slot >> 1)) / tiles_n) * BM + 128 >= M);
    const bool half_first = my_half && (tail_mode & 3) == 2 && (slot & 1);
    auto get_unit = [&](int ui) -> bool {                   // the ui-th unit of this workgroup
        if (half_first) {                                    // the tail unit comes first, the whole tiles after it
            if (ui == 0) { u_tile = run_tile(full_rounds * wg_x + (slot >> 1)); u_mode = 1 + (slot & 1); return true; }
            ui -= 1;
            if (ui < full_rounds) { u_tile = run_tile(slot + ui * wg_x); u_mode = 0; return true; }
            return false;
        }
        if (ui < full_rounds) { u_tile = run_tile(slot + ui * wg_x); u_mode = 0; return true; }
        if (ui > full_rounds) return false;
        if (split) {
            if (!my_half) return false;
            u_tile = run_tile(full_rounds * wg_x + (slot >> 1));
            u_mode = 1 + (slot & 1);
            return true;
        }
        if (slot >= rem) return false;
        u_tile = run_tile(full_rounds * wg_x + slot); u_mode = 0;
        return true;
    };

    // LDS-DMA sources: a half-tile = 128 rows x 128 B = 16 pieces of 8 rows; wave w moves pieces 2w, 2w+1 of every half-tile.
    // The LDS image is lane-linear, so the XOR swizzle sits on the SOURCE chunk (lane -> row lane/8, chunk (lane%8) ^ (row%8)).
    const int lr = lane >> 3, lch = (lane & 7) ^ lr;
    // A piece's source = a wave-uniform 64-bit base in SGPRs (the unit's first A / W row, advanced by 128 bytes per K-tile with scalar
    // adds) + a 32-bit byte offset per lane (row inside the unit, clamped to the matrix, and the swizzled chunk): no vector
    // arithmetic per piece.  (Round 2 kept 8 full pointers per lane and added kt * 128 to one for every piece: two 64-bit VALU adds
    // per piece in the L-segments, whose instruction count is what the K loop is bound by -- beside the partner wave's MFMAs an
    // L-segment gets about one issue slot per MFMA.)
    unsigned voff[4][2];     // [half-tile: A0, A1, W0, W1][piece]
    const char *abase = nullptr, *wbase = nullptr;
    int m0 = 0, n0 = 0;
    auto set_tile = [&](int t, int md) {                    // md != 0: the 128 rows of that half sit in rows 0..127 of the A stage
        m0 = (t / tiles_n) * BM + (md == 2 ? 128 : 0); n0 = (t % tiles_n) * BN;
        const int lt = PH_TILE_FOR_LOADS(t), lm0 = (lt / tiles_n) * BM + (md == 2 ? 128 : 0), ln0 = (lt % tiles_n) * BN;   // (= m0, n0 in the product build)
        abase = reinterpret_cast<const char*>(A + (int64_t)lm0 * lda);
        wbase = reinterpret_cast<const char*>(W + (int64_t)ln0 * ldw);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int ra = hf * 128 + (wave * 2 + i) * 8 + lr;          // row inside the unit; the unit's first row is inside the matrix
                ra = lm0 + ra < M ? ra : M - 1 - lm0;
                voff[hf][i] = (unsigned)(ra * lda + lch * 8) * 2u;
                int rw = hf * 128 + (wave * 2 + i) * 8 + lr;
                rw = ln0 + rw < N ? rw : N - 1 - ln0;
                voff[2 + hf][i] = (unsigned)(rw * ldw + lch * 8) * 2u;
            }
    };
    auto issue_half = [&](int hid, int kt, int stage) {   // hid: 0 A rows 0..127, 1 A rows 128..255, 2 W rows 0..127, 3 W rows 128..255
        const char* base = (hid >= 2 ? wbase : abase) + kt * 128;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row0 = (hid & 1) * 128 + (wave * 2 + i) * 8;
            const unsigned dst = lds_base + (unsigned)(stage * STAGE + (hid >= 2 ? SA : 0) + row0 * 8) * 16u;
            glds16_hidden_s(base, voff[hid][i], dst);
        }
    };
    // HALF UNITS (128 x 256 of C, mode 1 / 2) have their own K loop: a K-tile is A 128 rows (16 KB) + W 256 rows (32 KB) = 48 KB, so
    // THREE stages fit where the whole tiles have two (stage 2 = bytes 96 K .. 144 K overlaps the epilogue's C stage, which is only
    // live between two K loops), and a K-tile is consumed in TWO phases of 16 MFMAs per wave (each wave owns 64 x 64 of C: quadrants
    // (0, 0) and (0, 1) of the whole-tile wave tile) -- half the barriers of the 4-phase loop, which with two of its phases empty
    // took 0.82 of a whole tile's time for half the work.  Schedule (phase q = 2u + p of K-tile u, stage u % 3; early / late wave
    // groups as above: L(q) / C(q) = intervals 2q / 2q+1 resp. 2q+1 / 2q+2):
    //   reads : p0 reads W(j=0) + A, p1 reads W(j=1); every ds_read complete (lgkmcnt(0)) before the barrier that ends its L-segment,
    //           so stage u % 3 is last read in interval 2(2u+1)+1 = 4u+3.
    //   WAR   : stage (u+3) % 3 = u % 3 is refilled with K-tile u+3 in phases 2(u+1) (A0) and 2(u+1)+1 (W0, W1), first interval
    //           2(2u+2) = 4u+4 > 4u+3.
    //   RAW   : K-tile u+1 is first read in interval 2*2(u+1) = 4u+4; all its DMAs were issued in phases 2u-2, 2u-1 (or the
    //           prologue); every wave waits for ITS pieces at the end of interval 4u+3 (early: end of C(2u+1), late: end of
    //           L(2u+1)) with vmcnt(6) -- only the 6 DMAs of K-tile u+2, issued in phases 2u and 2u+1, may still be in flight --
    //           and the barrier that ends that interval publishes them.
    constexpr int HSA = 128 * 8, HSTAGE = (128 + 256) * 8;   // half units: 16-byte chunks of the A part / of a stage
    auto issue_h = [&](int hid, int kt, int st3) {          // hid 0: A rows 0..127, 2 / 3: W rows 0..127 / 128..255
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row0 = (hid & 1) * 128 + (wave * 2 + i) * 8;
            const unsigned dst = lds_base + (unsigned)(st3 * HSTAGE + (hid >= 2 ? HSA : 0) + row0 * 8) * 16u;
            glds16_hidden_s((hid >= 2 ? wbase : abase) + kt * 128, voff[hid][i], dst);
        }
    };
    // whole tiles: tile 0 whole, then W0 / A0 / A1 of tile 1 (the steady state issues W0 and A0 + A1 of K-tile u+2 in phases 4u+2
    // and 4u+3, W1 of K-tile u+1 in phase 4u); half units: K-tiles 0 and 1 whole (6 + 6 DMAs per wave)
    auto issue_prologue = [&](int nkt, int md) {
        if (md == 0) {
            issue_half(2, 0, 0); issue_half(0, 0, 0); issue_half(3, 0, 0); issue_half(1, 0, 0);
            if (nkt > 1) {
                if (PH == 2)      { issue_half(2, 1, 1); issue_half(3, 1, 1); }                   // two-phase loop: W of K-tile 1
                else if (PH == 1) { issue_half(0, 1, 1); issue_half(1, 1, 1); }                   // pipelined loop: A of K-tile 1
                else              { issue_half(2, 1, 1); issue_half(0, 1, 1); issue_half(1, 1, 1); }
            }
        } else {
            issue_h(2, 0, 0); issue_h(0, 0, 0); issue_h(3, 0, 0);
            if (nkt > 1) { issue_h(2, 1, 1); issue_h(0, 1, 1); issue_h(3, 1, 1); }
        }
    };

    // STREAM ACROSS WHOLE TILES: when this unit and the next are both whole tiles, the LDS-DMA schedule simply continues over the
    // boundary -- the "K-tile u+1 / u+2" of the last two K-tiles are the next unit's K-tiles 0 / 1 (their source addresses replace this
    // unit's in phase 1 of K-tile nkt-2, after this unit's last DMA and before the next unit's first) -- instead of a burst of 14 DMAs
    // per wave between the K loop and the epilogue (stamps: 1.7 us per tile during which all 8 waves only issue, and K-tile 0 / 1 get
    // half the lookahead).  Stages keep alternating: the next unit reads K-tile v from stage (v + sb) & 1, sb = (sb + nkt) & 1.
    // The hazard analysis above is unchanged: it is the same schedule with tile indices taken modulo the unit.  Units next to a half
    // unit (other stage geometry) and single-K-tile units start from the burst as before, with sb = 0.
    int ui = 0, sb = 0;
    bool streamed_in = false;                               // this unit's first K-tiles were issued inside the previous unit's K loop
    bool have = get_unit(0);
    int cur_tile = u_tile, cur_mode = u_mode;
    if (have) { set_tile(cur_tile, cur_mode); issue_prologue(KT, cur_mode); }
    while (have) {
        const int cm0 = m0, cn0 = n0;                       // this unit's origin (set_tile moves on to the next one below)
        const int nkt = KT, mode = cur_mode;
        const bool nx_have = get_unit(ui + 1);              // peek: u_tile / u_mode now describe the NEXT unit
        const int nx_tile = u_tile, nx_mode = u_mode;
        const bool stream = nx_have && nx_mode == 0 && mode == 0 && nkt >= 2;
        const int sbase = ui * 12;
        PH_STAMP(sbase + 0);                                // unit start
        const bool halfu = mode != 0;                       // workgroup-uniform
        const int wrows = halfu ? 64 : 128;                 // rows of the unit per wave row wm
        f32x4 acc[4][8];             // [n-tile of 16 columns][m-tile of 16 rows]
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // K-tile 0 has landed when at most the 6 DMAs of K-tile 1 issued by the prologue (W0 / A0 / A1 of a whole tile, all of a half
        // unit's) are still in flight.  (After the first unit the previous epilogue's stores are younger than these DMAs and count too: the
        // wait is then stronger, never weaker.)
        // (pipelined loop, unit streamed into: the previous unit's last barrier has already published this unit's K-tile 0)
        if (!(PH == 1 && streamed_in)) {
        if (nkt > 1) {
            if (PH != 4 && !halfu) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else               asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (wm == 1 && (PH != 1 || halfu)) __builtin_amdgcn_s_barrier();   // the late group starts one interval later (phase loops)
        }
        PH_STAMP(sbase + 1);                                // K-tile 0 landed, K loop starts

        v8 af[4][2], bf0[2][2], bf1[2][2];                // A: 4 m-tiles x 2 k-steps; W: 2 n-tiles x 2 k-steps, for j = 0 and j = 1
        if (!halfu && PH == 1) {
        // SOFTWARE-PIPELINED loop, ONE barrier per K-tile (switch bf16_gemm_phases = 1).  The phase loops separate a wave's LDS reads /
        // DMA issue (L-segment) from its MFMAs (C-segment) with a barrier and run the two waves of a SIMD in opposite roles; the
        // stamps and ablations of round 3 say what that costs: a K-tile's 64 MFMAs per wave are 1.0 us of matrix pipe for the SIMD's
        // two waves, the barrier skeleton alone (8 intervals) 0.69 us, the loop 1.5-1.6 us -- an interval is as long as the LONGER of
        // one wave's L-segment and the other's C-segment plus a barrier turn-around.  Here every wave runs ONE stream per K-tile: four
        // quarters of 16 MFMAs (quarter (ks, i): k-step ks of the K-tile, 64-row half i of the wave's rows, all 4 n-tiles), and among
        // the MFMAs of a quarter it reads the NEXT quarter's fragments into the other half of a double buffer (fa0 / fa1, fw0 / fw1:
        // the same 64 fragment registers as the phase loops) and issues its LDS-DMA pieces; hipcc counts lgkmcnt for the fragment
        // reads itself (it sees them), the DMAs stay hidden and hand-counted.  Nothing separates the two waves of a SIMD: the hardware
        // issues one wave's MFMAs under the other's reads and DMA issue.
        //   q0 (ks0, i0): reads A(ks0, i1);              issues W0, W1 of K-tile u+1 (4 pieces per wave)
        //   q1 (ks0, i1): reads A(ks1, i0) and W(ks1)
        //   q2 (ks1, i0): reads A(ks1, i1);              then lgkmcnt(0), vmcnt(0), BARRIER
        //   q3 (ks1, i1): reads A(ks0, i0), W(ks0) of K-tile u+1 (other stage);   issues A0, A1 of K-tile u+2 (4 pieces)
        //   WAR : the stage of K-tile u is last read in q2 (complete before the barrier: lgkmcnt(0)); its refill (A of K-tile u+2 in
        //         q3 of K-tile u, W of K-tile u+2 in q0 of K-tile u+1) is issued after that barrier.
        //   RAW : K-tile u+1 (A issued in q3 of K-tile u-1, W in q0 of K-tile u; nothing younger) is first read in q3 of K-tile u; every
        //         wave waits for ITS pieces with vmcnt(0) at the end of q2 and the barrier publishes them.  A has a whole K-tile of
        //         lookahead, the L2-resident W three quarters.
        // Unit start: K-tile 0 landed and published, A of K-tile 1 in flight (prologue: K-tile 0 + A of K-tile 1; streamed into: the
        // same state, left by the previous unit's last two K-tiles); the unit's first fragments are read outside the pipeline.
        v8 fa0[4], fa1[4], fw0[4], fw1[4];
        auto rdA = [&](const float4* sa_, int i, int ks, int m) -> v8 {
            const float4 v = sa_[((i * 4 + m) * 16 + c) * 8 + ((ks * 4 + g) ^ (c & 7))];
            return *reinterpret_cast<const v8*>(&v);
        };
        auto rdW = [&](const float4* sw_, int ks, int n) -> v8 {
            const float4 v = sw_[(n * 16 + c) * 8 + ((ks * 4 + g) ^ (c & 7))];
            return *reinterpret_cast<const v8*>(&v);
        };
        auto issue_piece = [&](int hid, int i, int kt, int stage) {   // one 1 KiB piece of half-tile hid (issue_half = pieces 0 and 1)
            const int row0 = (hid & 1) * 128 + (wave * 2 + i) * 8;
            const unsigned dst = lds_base + (unsigned)(stage * STAGE + (hid >= 2 ? SA : 0) + row0 * 8) * 16u;
            glds16_hidden_s((hid >= 2 ? wbase : abase) + kt * 128, voff[hid][i], dst);
        };
        {
            const float4* sbuf = lds + (sb & 1) * STAGE;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                fa0[m] = rdA(sbuf + (wm * 128) * 8, 0, 0, m);
                fw0[m] = rdW(sbuf + SA + (wn * 64) * 8, 0, m);
            }
        }
        if (PH_ABLATE(1)) {                                   // (tools-only ablation without fragment reads: defined, opaque operands)
#pragma unroll
            for (int m = 0; m < 4; ++m) { fa1[m] = fa0[m]; fw1[m] = fw0[m]; asm volatile("" : "+v"(fa1[m]), "+v"(fw1[m])); }
        }
        for (int u = 0; u < nkt; ++u) {
            const float4* sbuf = lds + ((u + sb) & 1) * STAGE;
            const float4* sa = sbuf + (wm * 128) * 8;
            const float4* sw = sbuf + SA + (wn * 64) * 8;
            const float4* nbuf = lds + ((u + 1 + sb) & 1) * STAGE;
            const float4* nsa = nbuf + (wm * 128) * 8;
            const float4* nsw = nbuf + SA + (wn * 64) * 8;
            const int g1 = u + 1, g2 = u + 2;               // K-tiles the DMAs of this K-tile belong to (>= nkt: the next unit's)
            const bool i1 = g1 < nkt || stream, i2 = g2 < nkt || stream;
            const int k1 = g1 < nkt ? g1 : g1 - nkt, k2 = g2 < nkt ? g2 : g2 - nkt;
            const bool nxt = g1 < nkt;
            // (reads come BEFORE the MFMAs of their group and in the first groups of a quarter, W fragments first: the next quarter's
            // first group needs all four W fragments and A fragment 0, and nothing is waited for right after it was requested)
            // ---------------- q0
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (!PH_ABLATE(1)) fa1[m] = rdA(sa, 1, 0, m);
                if (i1 && !PH_ABLATE(0)) issue_piece(2 + (m >> 1), m & 1, k1, (g1 + sb) & 1);
#pragma unroll
                for (int n = 0; n < 4; ++n) if (!PH_ABLATE(2)) acc[n][m] = X16<T>::mfma(fw0[n], fa0[m], acc[n][m]);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---------------- q1
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (!PH_ABLATE(1)) {
                if (m == 0) { fw1[0] = rdW(sw, 1, 0); fw1[1] = rdW(sw, 1, 1); fw1[2] = rdW(sw, 1, 2); }
                if (m == 1) { fw1[3] = rdW(sw, 1, 3); fa0[0] = rdA(sa, 0, 1, 0); fa0[1] = rdA(sa, 0, 1, 1); }
                if (m == 2) { fa0[2] = rdA(sa, 0, 1, 2); fa0[3] = rdA(sa, 0, 1, 3); }
                }
#pragma unroll
                for (int n = 0; n < 4; ++n) if (!PH_ABLATE(2)) acc[n][4 + m] = X16<T>::mfma(fw0[n], fa1[m], acc[n][4 + m]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (stream && u == nkt - 2) {                     // this unit's last DMA is out (q0): the next unit's addresses
                int t = nx_tile;
                asm volatile("" : "+s"(t));                   // (opaque: hipcc otherwise computes the next unit's 8 offsets at the unit start and keeps them live)
                set_tile(t, 0);
            }
            // ---------------- q2
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (m < 2 && !PH_ABLATE(1)) { fa1[2 * m] = rdA(sa, 1, 1, 2 * m); fa1[2 * m + 1] = rdA(sa, 1, 1, 2 * m + 1); }
#pragma unroll
                for (int n = 0; n < 4; ++n) if (!PH_ABLATE(2)) acc[n][m] = X16<T>::mfma(fw1[n], fa0[m], acc[n][m]);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every read of this stage is in registers before anyone may refill it
            if (i1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of K-tile u+1 have landed
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---------------- q3
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (nxt && !PH_ABLATE(1)) {
                    if (m == 0) { fw0[0] = rdW(nsw, 0, 0); fw0[1] = rdW(nsw, 0, 1); fw0[2] = rdW(nsw, 0, 2); }
                    if (m == 1) { fw0[3] = rdW(nsw, 0, 3); fa0[0] = rdA(nsa, 0, 0, 0); fa0[1] = rdA(nsa, 0, 0, 1); }
                    if (m == 2) { fa0[2] = rdA(nsa, 0, 0, 2); fa0[3] = rdA(nsa, 0, 0, 3); }
                }
                if (i2 && !PH_ABLATE(0)) issue_piece(m >> 1, m & 1, k2, (g2 + sb) & 1);
#pragma unroll
                for (int n = 0; n < 4; ++n) if (!PH_ABLATE(2)) acc[n][4 + m] = X16<T>::mfma(fw1[n], fa1[m], acc[n][4 + m]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        } else if (!halfu && PH == 2) {
        // TWO phases of 32 MFMAs per K-tile (switch bf16_gemm_phases = 2).  Compile-time ablations of the four-phase loop
        // (tools/gemm_bf16_ablate.sh) put the skeleton of a K-tile -- 8 barrier-to-barrier intervals with their reads and DMAs but
        // no MFMA -- at 0.89 us and its 64 MFMAs per wave at 1.0 us of matrix pipe for the SIMD's two waves, yet the loop takes 1.6 us:
        // every interval pays ~130 cycles of barrier turn-around on top of its 256 MFMA cycles.  Here a K-tile has 4 intervals of 512:
        //   reads : P0 reads W(j=0), W(j=1) and A(i=0) (16 ds_read_b128), P1 reads A(i=1) (8); the W fragments stay in registers.
        //           W of stage u & 1 is last read in interval 2(2u)+1 = 4u+1, A in interval 2(2u+1)+1 = 4u+3.
        //   WAR   : W0 / W1 of K-tile u+2 (same stage) are issued in phase 2u+1 (first interval 4u+2 > 4u+1), A0 / A1 of K-tile u+1
        //           (stage (u+1) & 1, last read in interval 4(u-1)+3 = 4u-1) in phase 2u (first interval 4u).
        //   RAW   : K-tile u+1 is first read in interval 4u+4; its W was issued in phase 2u-1, its A in phase 2u; every wave waits
        //           for ITS pieces at the end of interval 4u+3 (early: end of C(2u+1), late: end of L(2u+1)) with vmcnt(4) -- only
        //           the two W halves of K-tile u+2, issued in phase 2u+1, may still be in flight.
        // A gets 4 intervals (~2500 cycles) of lookahead instead of 10 x 390; W, the L2-resident operand, 6.
        for (int u = 0; u < nkt; ++u) {
            const float4* sbuf = lds + ((u + sb) & 1) * STAGE;
            const float4* sa = sbuf + (wm * 128) * 8;
            const float4* sw = sbuf + SA + (wn * 64) * 8;
            const int g1 = u + 1, g2 = u + 2;               // K-tiles the DMAs of this K-tile belong to (>= nkt: the next unit's)
            const bool i1 = g1 < nkt || stream, i2 = g2 < nkt || stream;
            const int k1 = g1 < nkt ? g1 : g1 - nkt, k2 = g2 < nkt ? g2 : g2 - nkt;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                // ---------------- L-segment
                if (p == 0) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
#pragma unroll
                            for (int ks = 0; ks < 2; ++ks) {
                                const float4 v = sw[((j * 2 + n) * 16 + c) * 8 + ((ks * 4 + g) ^ (c & 7))];
                                if (j == 0) bf0[n][ks] = *reinterpret_cast<const v8*>(&v);
                                else        bf1[n][ks] = *reinterpret_cast<const v8*>(&v);
                            }
                }
#pragma unroll
                for (int m = 0; m < 4; ++m)      // A fragments of the 64-row half i = p
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const float4 v = sa[((p * 4 + m) * 16 + c) * 8 + ((ks * 4 + g) ^ (c & 7))];
                        af[m][ks] = *reinterpret_cast<const v8*>(&v);
                    }
                __builtin_amdgcn_sched_barrier(0);
                if (p == 0 && i1) { issue_half(0, k1, (g1 + sb) & 1); issue_half(1, k1, (g1 + sb) & 1); }
                if (p == 1 && stream && u == nkt - 2) set_tile(nx_tile, 0);   // this unit's last DMA is out: the next unit's addresses
                if (p == 1 && i2) { issue_half(2, k2, (g2 + sb) & 1); issue_half(3, k2, (g2 + sb) & 1); }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this phase's reads are in registers before anyone may refill
                if (p == 1 && wm == 1) {     // the late group confirms K-tile u+1 at the end of its L(2u+1) ...
                    if (i2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                // ---------------- C-segment: quadrants (i = p, j = 0) and (i = p, j = 1)
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n) {
                            acc[n][p * 4 + m] = X16<T>::mfma(bf0[n][ks], af[m][ks], acc[n][p * 4 + m]);
                            acc[2 + n][p * 4 + m] = X16<T>::mfma(bf1[n][ks], af[m][ks], acc[2 + n][p * 4 + m]);
                        }
                __builtin_amdgcn_s_setprio(0);
                if (p == 1 && wm == 0) {     // ... the early group at the end of its C(2u+1): the same interval
                    if (i2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            }
        }
        } else if (!halfu) {
        for (int u = 0; u < nkt; ++u) {
            const float4* sbuf = lds + ((u + sb) & 1) * STAGE;
            const float4* sa = sbuf + (wm * 128) * 8;
            const float4* sw = sbuf + SA + (wn * 64) * 8;
            const int g1 = u + 1, g2 = u + 2;               // K-tiles the DMAs of this K-tile belong to (>= nkt: the next unit's)
            const bool i1 = g1 < nkt || stream, i2 = g2 < nkt || stream;
            const int k1 = g1 < nkt ? g1 : g1 - nkt, k2 = g2 < nkt ? g2 : g2 - nkt;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                // ---------------- L-segment
                if ((p == 0 || p == 1) && !PH_ABLATE(1)) {      // W fragments of the 32-column half j = p
#pragma unroll
                    for (int n = 0; n < 2; ++n)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            const float4 v = sw[((p * 2 + n) * 16 + c) * 8 + ((ks * 4 + g) ^ (c & 7))];
                            if (p == 0) bf0[n][ks] = *reinterpret_cast<const v8*>(&v);
                            else        bf1[n][ks] = *reinterpret_cast<const v8*>(&v);
                        }
                }
                if ((p == 0 || p == 2) && !PH_ABLATE(1)) {      // A fragments of the 64-row half i = p / 2
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            const float4 v = sa[(((p >> 1) * 4 + m) * 16 + c) * 8 + ((ks * 4 + g) ^ (c & 7))];
                            af[m][ks] = *reinterpret_cast<const v8*>(&v);
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!PH_ABLATE(0)) {
                if (p == 0 && i1) issue_half(3, k1, (g1 + sb) & 1);
                if (p == 1 && stream && u == nkt - 2) set_tile(nx_tile, 0);   // this unit's last DMA is out: the next unit's addresses
                if (p == 2 && i2) issue_half(2, k2, (g2 + sb) & 1);
                if (p == 3 && i2) { issue_half(0, k2, (g2 + sb) & 1); issue_half(1, k2, (g2 + sb) & 1); }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this phase's reads are in registers before anyone may refill
                if (p == 3 && wm == 1) {     // the late group confirms K-tile u+1 at the end of its L(4u+3) ...
                    if (i2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                // ---------------- C-segment: quadrant (i, j) = (0,0) (0,1) (1,1) (1,0)
                __builtin_amdgcn_s_setprio(1);
                if (!PH_ABLATE(2))
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n) {
                            const int i = p >> 1, j = (p == 1 || p == 2) ? 1 : 0;
                            const v8 wv = j ? bf1[n][ks] : bf0[n][ks];
                            acc[j * 2 + n][i * 4 + m] = X16<T>::mfma(wv, af[m][ks], acc[j * 2 + n][i * 4 + m]);
                        }
                __builtin_amdgcn_s_setprio(0);
                if (p == 3 && wm == 0) {     // ... the early group at the end of its C(4u+3): the same interval
                    if (i2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            }
        }
        } else {
        int st3 = 0, st3n = 2;                                // ring slots of K-tile u and of K-tile u + 2
        for (int u = 0; u < nkt; ++u) {
            const float4* sb = lds + st3 * HSTAGE;
            const float4* sa = sb + (wm * 64) * 8;
            const float4* sw = sb + HSA + (wn * 64) * 8;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                // ---------------- L-segment
#pragma unroll
                for (int n = 0; n < 2; ++n)      // W fragments of the 32-column half j = p
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const float4 v = sw[((p * 2 + n) * 16 + c) * 8 + ((ks * 4 + g) ^ (c & 7))];
                        if (p == 0) bf0[n][ks] = *reinterpret_cast<const v8*>(&v);
                        else        bf1[n][ks] = *reinterpret_cast<const v8*>(&v);
                    }
                if (p == 0) {                    // A fragments: the wave's 64 rows
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) {
                            const float4 v = sa[(m * 16 + c) * 8 + ((ks * 4 + g) ^ (c & 7))];
                            af[m][ks] = *reinterpret_cast<const v8*>(&v);
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (p == 0 && u + 2 < nkt) issue_h(0, u + 2, st3n);                                 // A first: the operand that misses L2
                if (p == 1 && u + 2 < nkt) { issue_h(2, u + 2, st3n); issue_h(3, u + 2, st3n); }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (p == 1 && wm == 1) {     // the late group confirms K-tile u+1 at the end of its L(2u+1) ...
                    if (u + 2 < nkt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                // ---------------- C-segment: quadrant (0, j = p)
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n) {
                            const v8 wv = p ? bf1[n][ks] : bf0[n][ks];
                            acc[p * 2 + n][m] = X16<T>::mfma(wv, af[m][ks], acc[p * 2 + n][m]);
                        }
                __builtin_amdgcn_s_setprio(0);
                if (p == 1 && wm == 0) {     // ... the early group at the end of its C(2u+1): the same interval
                    if (u + 2 < nkt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            }
            st3 = st3 == 2 ? 0 : st3 + 1;
            st3n = st3n == 2 ? 0 : st3n + 1;
        }
        }
        if (wm == 0 && (PH != 1 || halfu)) __builtin_amdgcn_s_barrier();   // the early group waits out the late group's last C-segment
        PH_STAMP(sbase + 2);                                // K loop done

        // Both operand stages are free now (every read of the last K-tiles completed before the barriers above).  Epilogues without a
        // residual put the NEXT unit's first DMAs in flight right here, so their latency runs under the whole epilogue.  The residual
        // epilogues (RESV) first request ALL residual rows of the unit and issue those DMAs one LDS pass later, behind an explicit
        // wait: hipcc counts only the loads it can see, so with hidden DMAs in flight every wait it inserts for a residual row is a
        // wait for (most of) the DMAs too, and vmcnt retires in order anyway -- round 2 prefetched the residual one pass ahead and
        // in-kernel stamps (tools/gemm_bf16_stamps.py) showed every pass of an epilogue-5 tile sitting through a 2-3 us round trip
        // (16.5 us outside the K loop against 6.5 for epilogue 0).  The residual rows of passes 0 and 1 are requested up front (older
        // than the DMAs); those of pass k + 2 at the end of pass k, into the registers it has freed -- younger than the DMAs, which
        // have had two passes to land by the time these are waited for.  (All four passes up front: 96 more live registers, 40 spilled.)
        float4 bv[4];                                          // !RESV: this unit's bias for the lane's 4 x 4 columns, before the DMAs
        if (!RESV) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int col = cn0 + wn * 64 + n * 16 + 4 * g;
                bv[n] = (bias && col < N) ? ld4(bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        ++ui;
        have = nx_have;
        cur_tile = nx_tile; cur_mode = nx_mode;
        if (have && !stream) set_tile(nx_tile, nx_mode);   // (stream: done inside the K loop, and the next unit's K-tile 0 has landed)
        if (!RESV && have && !stream) issue_prologue(KT, nx_mode);
        sb = stream ? (sb + nkt) & 1 : 0;
        streamed_in = stream;

        const int kpass = halfu ? 2 : 4;                       // epilogue passes of 2 m-tiles per wave row (a half unit has 4 m-tiles)
        PH_STAMP(sbase + 3);                                // next unit's prologue issued (RESV: only its addresses are ready)
        // Epilogue through the 32 KB C stage in 4 passes of 64 rows (m-tiles 2k, 2k+1 of both wave rows): bias / GELU in registers,
        // bf16 rows staged with the 16-byte chunk XOR-swizzled by the row (the 16 rows a ds_write touches would otherwise share
        // their banks: the row pitch is 512 B), streamed out as whole rows, 16 bytes per lane; the residual (and, with it, the bias)
        // is added on the way out.  Raw barriers + lgkmcnt only: a __syncthreads() would wait for the DMAs in flight.
        unsigned char* sC = reinterpret_cast<unsigned char*>(lds + CST);
        // LayerNorm folded into this GEMM (LNF): this lane's rows' (mean, rstd) and its columns' c1 = sum_k W'[n, k]
        float mu[8], rs[8];
        float4 c1v[4];
        if (LNF) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                int row = cm0 + wm * wrows + m * 16 + c;
                row = row < M ? row : M - 1;
                const float2 st = *reinterpret_cast<const float2*>(ln.rowstat + (int64_t)row * 2);
                mu[m] = st.x; rs[m] = st.y;
            }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int col = cn0 + wn * 64 + n * 16 + 4 * g;
                c1v[n] = col < N ? ld4(ln.c1 + col) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        // RES_LN: the residual is LayerNorm(res row) -- the raw row comes with its (mean, rstd); gamma / beta of this thread's 8
        // columns are loaded once per unit (a thread keeps its 16-byte chunk index through all passes), the bias joins beta.
        // (An already-normalised residual comes with identity tables -- mean 0, rstd 1, gamma 1, beta 0 -- not with a null pointer.
        // N % 256 == 0 for the LayerNorm epilogues, so the 8 columns are always in range; the plain residual epilogue clamps.)
        v8 rr[2][4];                                          // ring of two passes
        f32x2 ss[2][4];
        auto load_res = [&](int k, v8 (&rd)[4], f32x2 (&sd)[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int id = tid + i * 512, prow = id >> 5, ch = id & 31;
                const int lrow = (prow >> 5) * wrows + (2 * k + ((prow >> 4) & 1)) * 16 + (prow & 15);
                int row = cm0 + lrow, col = cn0 + ch * 8;
                row = row < M ? row : M - 1; col = col < N ? col : 0;         // clamped: loaded, not used
                rd[i] = *reinterpret_cast<const v8*>(res + (int64_t)row * ldc + col);
                if (EPI == BEPI_RES_LN) sd[i] = *reinterpret_cast<const f32x2*>(ln.rowstat + (int64_t)row * 2);
            }
        };
        const int my_col = cn0 + (tid & 31) * 8;
        f32x4 gam0, gam1, bet0, bet1;
        if (RESV) {
            auto ldv4 = [](const float* p) { return *reinterpret_cast<const f32x4*>(p); };
            const int bc = my_col < N ? my_col : 0;
            bet0 = bias ? ldv4(bias + bc) : (f32x4){0.f, 0.f, 0.f, 0.f};
            bet1 = bias ? ldv4(bias + bc + 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
            if (EPI == BEPI_RES_LN) {
                gam0 = ldv4(ln.ln_g + my_col); gam1 = ldv4(ln.ln_g + my_col + 4);
                bet0 += ldv4(ln.ln_b + my_col); bet1 += ldv4(ln.ln_b + my_col + 4);
            }
            load_res(0, rr[0], ss[0]);
            load_res(1, rr[1], ss[1]);                         // (kpass >= 2 always)
        }
        if constexpr (EPI != BEPI_BIAS_F32) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k >= kpass) break;                            // workgroup-uniform
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
                const int m = 2 * k + mm;
                const int prow = (wm * 2 + mm) * 16 + c;      // row inside the pass
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const int lcol = wn * 64 + n * 16 + 4 * g;
                    float4 v = make_float4(acc[n][m][0], acc[n][m][1], acc[n][m][2], acc[n][m][3]);
                    if (LNF)   // rstd (A W'^T - mean c1) + c2, c2 in the bias slot
                        v = make_float4(rs[m] * fmaf(-mu[m], c1v[n].x, v.x), rs[m] * fmaf(-mu[m], c1v[n].y, v.y),
                                        rs[m] * fmaf(-mu[m], c1v[n].z, v.z), rs[m] * fmaf(-mu[m], c1v[n].w, v.w));
                    if (!RESV) v = add4(v, bv[n]);
                    if (EPI == BEPI_BIAS_GELU || EPI == BEPI_LNFOLD_GELU) {
                        const f32x2 g0 = gelu_fast2((f32x2){v.x, v.y}), g1 = gelu_fast2((f32x2){v.z, v.w});
                        v = make_float4(g0[0], g0[1], g1[0], g1[1]);
                    }
                    v4 o = {(T)v.x, (T)v.y, (T)v.z, (T)v.w};
                    const int ch = (lcol >> 3) ^ (prow & 31);
                    *reinterpret_cast<v4*>(sC + prow * 512 + ch * 16 + (lcol & 7) * 2) = o;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            PH_STAMP(sbase + 4 + 2 * k);                        // pass k staged
            if (RESV && k == 0) {
                // every residual chunk of the unit has landed (and the compiler is told so: the empty statements "write" the registers,
                // so no wait of its own can follow once the hidden DMAs are in flight); only now the next unit's first DMAs
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        asm volatile("" : "+v"(rr[kk][i]));
                        if (EPI == BEPI_RES_LN) asm volatile("" : "+v"(ss[kk][i]));
                    }
                asm volatile("" : "+v"(bet0), "+v"(bet1));
                if (EPI == BEPI_RES_LN) asm volatile("" : "+v"(gam0), "+v"(gam1));
                if (have && !stream) issue_prologue(KT, nx_mode);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {                      // 64 rows x 32 chunks = 2048 chunks / 512 threads
                const int id = tid + i * 512, prow = id >> 5, ch = id & 31;
                const int lrow = (prow >> 5) * wrows + (2 * k + ((prow >> 4) & 1)) * 16 + (prow & 15);
                const int row = cm0 + lrow, col = cn0 + ch * 8;
                const bool inb = row < M && col < N;
                // The read-out side works on PAIRS with packed fp32 arithmetic (v_pk_fma_f32 / v_pk_add_f32: fine in an epilogue, an
                // anti-lever beside MFMAs): epilogue 5 was ~2200 VALU instructions per wave and tile more than epilogue 0, ~9 us of the
                // 17 us an out-proj tile spent outside its K loop.  A bf16 pair sits in one dword: lo << 16 and hi & 0xffff0000 are
                // the two fp32 values.
                u32x4 vw = *reinterpret_cast<const u32x4*>(sC + prow * 512 + ((ch ^ (prow & 31)) * 16));
                if (RESV) {
                    const u32x4 rw = __builtin_bit_cast(u32x4, rr[k & 1][i]);
                    // LayerNorm of the residual row: (r - mean) rstd gamma + beta = (r a + b) gamma + beta with a = rstd, b = -mean rstd
                    // (beta carries the GEMM's bias too)
                    const float ra = ss[k & 1][i][1], rb = -ss[k & 1][i][0] * ss[k & 1][i][1];
                    const f32x2 gam2[4] = {{gam0[0], gam0[1]}, {gam0[2], gam0[3]}, {gam1[0], gam1[1]}, {gam1[2], gam1[3]}};
                    const f32x2 bet2[4] = {{bet0[0], bet0[1]}, {bet0[2], bet0[3]}, {bet1[0], bet1[1]}, {bet1[2], bet1[3]}};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x2 x = X16<T>::unpack(vw[q]);
                        f32x2 r = X16<T>::unpack(rw[q]);
                        if (EPI == BEPI_RES_LN) {
                            r = __builtin_elementwise_fma(r, (f32x2){ra, ra}, (f32x2){rb, rb});
                            r = __builtin_elementwise_fma(r, gam2[q], bet2[q]);
                        } else {
                            r = r + bet2[q];                           // the bias
                        }
                        x = x + r;
                        if constexpr (!X16<T>::is_bf16) {   // fp16 residual stream of the decoder's big prefill: saturate instead of inf -> NaN
                            x[0] = __builtin_amdgcn_fmed3f(x[0], -65504.f, 65504.f);   // (bf16 has fp32's range)
                            x[1] = __builtin_amdgcn_fmed3f(x[1], -65504.f, 65504.f);
                        }
                        const v2 o = {(T)x[0], (T)x[1]};
                        vw[q] = __builtin_bit_cast(unsigned, o);
                    }
                }
                if (inb) *reinterpret_cast<u32x4*>(C + (int64_t)row * ldc + col) = vw;
                if (EPI == BEPI_RES_LN && ln.stats_out) {
                    // (sum, M2 = sum of squared deviations from the TILE's mean) of the row's 256 bf16 outputs in this tile (N % 256 == 0
                    // for this epilogue: every tile has 256 valid columns); the row's 32 chunks sit in 32 consecutive lanes.  Deviations
                    // from the tile mean, not raw squares: E[x^2] - mean^2 in fp32 loses the variance of a row whose mean is large
                    // against its spread (outlier hidden dimensions of trained checkpoints); ln_rowstat merges the tiles exactly (Chan).
                    // (rows beyond M compute garbage that is never stored)
                    f32x2 s1p = {0.f, 0.f}, s2p = {0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 4; ++q) s1p += X16<T>::unpack(vw[q]);
                    const float s1 = half_wave_sum(s1p[0] + s1p[1]);
                    const float tmean = s1 * (1.0f / 256.0f);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x2 d = X16<T>::unpack(vw[q]) - (f32x2){tmean, tmean};
                        s2p = __builtin_elementwise_fma(d, d, s2p);
                    }
                    const float s2 = half_wave_sum(s2p[0] + s2p[1]);
                    if (ch == 0 && row < M)
                        *reinterpret_cast<float2*>(ln.stats_out + ((int64_t)row * tiles_n + (cn0 >> 8)) * 2) = make_float2(s1, s2);
                }
            }
            if (RESV && k + 2 < kpass) load_res(k + 2, rr[k & 1], ss[k & 1]);   // the slot this pass has just consumed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                     // the C stage is rewritten by the next pass
            PH_STAMP(sbase + 5 + 2 * k);                        // pass k read out, stores issued
        }
        } else {
            // fp32 rows: 8 passes (4 for a half unit) of 32 rows -- m-tile k of both wave rows -- through the same 32 KB C stage:
            // 32 rows x 1 KB, the 16-byte chunk XOR-swizzled by the row (the 16 rows a ds_write_b128 touches would share their
            // banks: the row pitch is a multiple of the 256-byte bank row), streamed out as whole rows, 16 bytes per lane.
            float* Cf = reinterpret_cast<float*>(C);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k < 2 * kpass) {                              // workgroup-uniform (no `break`: hipcc then leaves the loop rolled and acc[n][k] goes to scratch)
                const int prow = wm * 16 + c;
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const int lcol = wn * 64 + n * 16 + 4 * g;
                    const float4 v = add4(make_float4(acc[n][k][0], acc[n][k][1], acc[n][k][2], acc[n][k][3]), bv[n]);
                    *reinterpret_cast<float4*>(sC + prow * 1024 + (((lcol >> 2) ^ (prow & 15)) * 16)) = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int i = 0; i < 4; ++i) {                      // 32 rows x 64 chunks = 2048 chunks / 512 threads
                    const int id = tid + i * 512, pr = id >> 6, ch = id & 63;
                    const int row = cm0 + (pr >> 4) * wrows + k * 16 + (pr & 15), col = cn0 + ch * 4;
                    const float4 v = *reinterpret_cast<const float4*>(sC + pr * 1024 + ((ch ^ (pr & 15)) * 16));
                    // 2.2 GB of logits at the decoder's [64, 1024] prefill that no kernel of the forward reads back: non-temporal stores
                    if (row < M && col < N) __builtin_nontemporal_store(*reinterpret_cast<const f32x4*>(&v), reinterpret_cast<f32x4*>(Cf + (int64_t)row * ldc + col));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                     // the C stage is rewritten by the next pass
                }
            }
        }
    }
}

template <int EPI, typename T, int PH>
static int launch_ph2(const T* a, int lda, const T* w, int ldw, const float* bias, const T* r, T* c, int ldc, int M,
                      int N, int K, hipStream_t st, GemmBf16Info* info, const BfEpiLn* lnp) {
    const int shmem = 2 * (256 + 256) * 128 + 32768;   // two operand stages + the C stage: all 160 KB of the CU's LDS
    DeviceInfo di;
    MGEA_TRY(device_info(&di));
    static uint64_t attr_done = 0;                      // per instantiation, one bit per device
    MGEA_TRY(set_max_dynamic_lds(reinterpret_cast<const void*>(&gemm_bf16_ph_kernel<EPI, T, PH>), shmem, di.dev, &attr_done));
    const int tm = ceil_div(M, 256), tn = ceil_div(N, 256), n_tiles = tm * tn;
    const int n_cu = di.n_cu / 8 * 8;
    const int grid = (int)round_up(n_tiles < n_cu ? n_tiles : n_cu, 8);   // one persistent workgroup per CU (160 KB of LDS each)
    const int tail = (tune(TUNE_BF16_GEMM_TAIL) & 3) | ((info && info->reverse && tune(TUNE_BF16_GEMM_REVERSE)) ? 4 : 0);
    if (info) {   // what the kernel will do with the tiles left after the full rounds (the same arithmetic as in the kernel, XCD run 0)
        const int wg_x = grid / 8, per_x = (n_tiles + 7) / 8 < n_tiles ? (n_tiles + 7) / 8 : n_tiles;
        const int rem = per_x - per_x / wg_x * wg_x;
        info->kernel = 2;
        info->half_tiles = ((tail & 3) != 0 && rem > 0 && 2 * rem <= wg_x) ? 1 : 0;
    }
    hipLaunchKernelGGL((gemm_bf16_ph_kernel<EPI, T, PH>), dim3(grid), dim3(512), shmem, st, a, lda, w, ldw, bias, r, c, ldc, M, N, K, tn, n_tiles,
                       tail, lnp ? *lnp : BfEpiLn{nullptr, nullptr, nullptr, nullptr, nullptr});
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}
// ------------------------------------------------------------------------------------------
// TWO WORKGROUPS PER CU ("duo"): 256 x 128 of C per workgroup of 4 waves (2 x 2, the same 128 x 64 per wave and the same fragment
// layout as the 256 x 256 kernel), 80 KB of LDS each, so that a CU holds two of them and they run INDEPENDENTLY: while one is in
// its epilogue (VALU, LDS, global stores; matrix pipe idle), waiting at a barrier or for its fragments, the other one's waves have the
// matrix pipe of every SIMD to themselves.  The 8-wave kernel above is one workgroup whose waves all reach the epilogue together:
// stamps put 24-42 % of a tile's time outside the K loop at K = 768 (VERDICT r2 #2 asks for exactly this overlap).  Price: the tile
// has 1.5 x the operand bytes per FLOP.
//   LDS  : K-tiles of 64 (whole 128-byte lines per row: a first version with K-steps of 32 and a three-stage ring made 64-byte requests,
//          3 x the L2 requests of the 256 x 256 kernel per FLOP, and ran at 0.77 of its rate -- profiles/r3_gemm_duo_bk32_*).  80 KB hold
//          TWO A stages (256 rows x 128 B = 32 KB each) and ONE W buffer (128 rows, 16 KB): W is the operand that sits in L2, its K-tile
//          is read into registers (8 fragments, both k-steps) at the top of a K-tile and refilled behind a second barrier.
//   loop : K-tile u: vmcnt(0) [A(u), W(u): nothing younger], barrier B1 (publishes them; A stage (u+1) & 1 is free: every wave has left
//          K-tile u-1); read the 8 W fragments and the 8 A fragments of k-step 0; lgkmcnt(0), barrier B2 (the W buffer is free);
//          then 64 MFMAs, m-tile by m-tile, with the A fragments of k-step 1 refilled in place behind their last use, and among them
//          this wave's 8 pieces of A(u+1) and 4 of W(u+1).  The exposed reads at the top are what the OTHER workgroup covers.
//   tile : the whole 256 x 128 bf16 tile (64 KB) is staged through the two A stages after the K loop, one barrier, whole-row stores.
// MEASURED (round 3, in-process A/B against the 256 x 256 kernel, outputs bitwise equal; profiles/r3_gemm_duo_*): 0.92 of its rate on QKV
// (K = 768), 0.91 on FC1 with GELU -- the GELU costs this kernel what it costs the other one, nothing hides it --, 0.99 on the out-proj shape,
// 0.93 at K = 3072; 1.25 x at K = 192 and 1.05 x at 8192^3.  Delaying one workgroup of every CU by 4-16 us changes nothing.  NOT the engine's
// choice: switch bf16_gemm_tile = 5, plain epilogues (bias, bias + GELU) only.
template <int EPI, typename T>
__global__ __launch_bounds__(256, 2) void gemm16_duo_kernel(const T* __restrict__ A, int lda, const T* __restrict__ W, int ldw,
                                                           const float* __restrict__ bias, const T* __restrict__ res,
                                                           T* __restrict__ C, int ldc, int M, int N, int K, int tiles_n, int n_tiles) {
    typedef typename X16<T>::v8 v8;
    typedef typename X16<T>::v4 v4;
    constexpr int BM = 256, BN = 128;
    constexpr int SA = BM * 8, WOFF = 2 * SA;               // in 16-byte chunks: two A stages, then the W buffer
    extern __shared__ __attribute__((aligned(16))) float4 lds[];   // 80 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int KT = K >> 6;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float4*)lds;
    // the XCD's contiguous run of tiles, walked with stride gridDim.x / 8 (speed only)
    const int wg_x = gridDim.x >> 3, slot = (int)blockIdx.x >> 3;
    const int run_len = (n_tiles + 7) >> 3, run0 = ((int)blockIdx.x & 7) * run_len;
    const int per_x = run0 + run_len < n_tiles ? run_len : n_tiles - run0;
    // fragment reads: row c of a 16-row tile, chunk (ks * 4 + g) ^ (c & 7); DMA pieces of 8 rows: lane -> row lane / 8, chunk (lane % 8) ^ row
    const int fr0 = c * 8 + (g ^ (c & 7)), fr1 = c * 8 + ((4 + g) ^ (c & 7));
    const int lr = lane >> 3, lch = (lane & 7) ^ lr;
    for (int it = slot; it < per_x; it += wg_x) {
        int t = run0 + it;
        asm volatile("" : "+s"(t));                           // (opaque: keeps the per-tile address arithmetic inside the tile loop)
        const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;
        unsigned voa[8], vow[4];                              // this wave's pieces: A rows (8 w + i) * 8 .., W rows (4 w + i) * 8 ..
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int r = (wave * 8 + i) * 8 + lr;
            r = m0 + r < M ? r : M - 1 - m0;
            voa[i] = (unsigned)(r * lda + lch * 8) * 2u;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int r = (wave * 4 + i) * 8 + lr;
            r = n0 + r < N ? r : N - 1 - n0;
            vow[i] = (unsigned)(r * ldw + lch * 8) * 2u;
        }
        const char* abase = reinterpret_cast<const char*>(A + (int64_t)m0 * lda);
        const char* wbase = reinterpret_cast<const char*>(W + (int64_t)n0 * ldw);
        auto issue_a = [&](int i, int kt, int st) {
            glds16_hidden_s(abase + kt * 128, voa[i], lds_base + (unsigned)(st * SA + (wave * 8 + i) * 64) * 16u);
        };
        auto issue_w = [&](int i, int kt) {
            glds16_hidden_s(wbase + kt * 128, vow[i], lds_base + (unsigned)(WOFF + (wave * 4 + i) * 64) * 16u);
        };
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_w(i, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) issue_a(i, 0, 0);
        f32x4 acc[4][8];
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[n][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto rd = [&](const float4* p_) -> v8 { const float4 v = *p_; return *reinterpret_cast<const v8*>(&v); };
        for (int u = 0; u < KT; ++u) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of A(u), W(u) (and, at u = 0, the previous tile's stores)
            __builtin_amdgcn_s_barrier();                        // B1
            const float4* sa = lds + (u & 1) * SA + (wm * 128) * 8;
            const float4* sw = lds + WOFF + (wn * 64) * 8;
            v8 fw[2][4], fa[8];
#pragma unroll
            for (int n = 0; n < 4; ++n) { fw[0][n] = rd(sw + n * 128 + fr0); fw[1][n] = rd(sw + n * 128 + fr1); }
#pragma unroll
            for (int m = 0; m < 8; ++m) fa[m] = rd(sa + m * 128 + fr0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();                        // B2: every wave holds W(u) in registers
            const bool nx = u + 1 < KT;
            const int st1 = (u + 1) & 1;
#pragma unroll
            for (int m = 0; m < 8; ++m) {                        // k-step 0
                if (nx) issue_a(m, u + 1, st1);
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[n][m] = X16<T>::mfma(fw[0][n], fa[m], acc[n][m]);
                fa[m] = rd(sa + m * 128 + fr1);                   // k-step 1's fragment, in place behind the last use
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) {                        // k-step 1
                if (nx && m < 4) issue_w(m, u + 1);
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[n][m] = X16<T>::mfma(fw[1][n], fa[m], acc[n][m]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // epilogue: the tile as bf16 through the two A stages (row pitch 256 B, 16-byte chunk XOR-swizzled by the row), whole-row stores
        __builtin_amdgcn_s_barrier();                         // every wave has read its last fragments
        unsigned char* sC = reinterpret_cast<unsigned char*>(lds);
        int tid_o = tid, c_o = c, g_o = g;                    // opaque per-tile copies: hipcc otherwise computes every epilogue address
        asm volatile("" : "+v"(tid_o), "+v"(c_o), "+v"(g_o));  // (thread-index-only arithmetic) before the K loop and keeps it live through it
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int lcol = wn * 64 + n * 16 + 4 * g_o;
            const int col = n0 + lcol;
            const float4 bv = (bias && col < N) ? ld4(bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int row = wm * 128 + m * 16 + c_o;
                float4 v = add4(make_float4(acc[n][m][0], acc[n][m][1], acc[n][m][2], acc[n][m][3]), bv);
                if (EPI == BEPI_BIAS_GELU) {
                    const f32x2 g0 = gelu_fast2((f32x2){v.x, v.y}), g1 = gelu_fast2((f32x2){v.z, v.w});
                    v = make_float4(g0[0], g0[1], g1[0], g1[1]);
                }
                v4 o = {(T)v.x, (T)v.y, (T)v.z, (T)v.w};
                *reinterpret_cast<v4*>(sC + row * 256 + (((lcol >> 3) ^ (row & 15)) * 16) + (lcol & 7) * 2) = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 16; ++i) {                        // 256 rows x 16 chunks / 256 threads
            const int id = tid_o + i * 256, row = id >> 4, ch = id & 15;
            const u32x4 vw = *reinterpret_cast<const u32x4*>(sC + row * 256 + ((ch ^ (row & 15)) * 16));
            const int grow = m0 + row, gcol = n0 + ch * 8;
            if (grow < M && gcol < N) *reinterpret_cast<u32x4*>(C + (int64_t)grow * ldc + gcol) = vw;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                         // the stages are free for the next tile's first pieces
    }
}

template <int EPI, typename T>
static int launch_duo(const T* a, int lda, const T* w, int ldw, const float* bias, const T* r, T* c, int ldc, int M,
                      int N, int K, hipStream_t st) {
    const int shmem = 2 * 256 * 128 + 128 * 128;        // 80 KB: two workgroups per CU
    DeviceInfo di;
    MGEA_TRY(device_info(&di));
    static uint64_t attr_done = 0;
    MGEA_TRY(set_max_dynamic_lds(reinterpret_cast<const void*>(&gemm16_duo_kernel<EPI, T>), shmem, di.dev, &attr_done));
    const int tm = ceil_div(M, 256), tn = ceil_div(N, 128), n_tiles = tm * tn;
    const int slots = 2 * (di.n_cu / 8 * 8);
    const int grid = (int)round_up(n_tiles < slots ? n_tiles : slots, 8);
    hipLaunchKernelGGL((gemm16_duo_kernel<EPI, T>), dim3(grid), dim3(256), shmem, st, a, lda, w, ldw, bias, r, c, ldc, M, N, K, tn, n_tiles);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

template <int EPI, typename T = bf16_t>
static int launch_ph(const T* a, int lda, const T* w, int ldw, const float* bias, const T* r, T* c, int ldc, int M,
                     int N, int K, hipStream_t st, GemmBf16Info* info, const BfEpiLn* lnp = nullptr) {
    const int ph = tune(TUNE_BF16_GEMM_PHASES);
    if (ph == 1) return launch_ph2<EPI, T, 1>(a, lda, w, ldw, bias, r, c, ldc, M, N, K, st, info, lnp);
    if (ph == 2) return launch_ph2<EPI, T, 2>(a, lda, w, ldw, bias, r, c, ldc, M, N, K, st, info, lnp);
    return launch_ph2<EPI, T, 4>(a, lda, w, ldw, bias, r, c, ldc, M, N, K, st, info, lnp);
}

template <int EPI, int NT, int WMW>
static int launch_glds(const bf16_t* a, int lda, const bf16_t* w, int ldw, const float* bias, const bf16_t* r, bf16_t* c,
                       int ldc, int M, int N, int K, hipStream_t st) {
    constexpr int BM = 128 * WMW, BN = 64 * NT;
    constexpr int NS = (NT == 2 && WMW == 2) ? 3 : 2;
    constexpr size_t ring = (size_t)NS * (BM + BN) * 128, stage_c = (size_t)BM * (BN * 2 + 16);
    const int shmem = (int)(ring > stage_c ? ring : stage_c);      // 64 KB .. 144 KB of the 160 KB LDS
    DeviceInfo di;
    MGEA_TRY(device_info(&di));
    static uint64_t attr_done = 0;   // one per instantiation
    MGEA_TRY(set_max_dynamic_lds(reinterpret_cast<const void*>(&gemm_bf16_glds_kernel<EPI, NT, WMW>), shmem, di.dev, &attr_done));
    const int tm = ceil_div(M, BM), tn = ceil_div(N, BN);
    hipLaunchKernelGGL((gemm_bf16_glds_kernel<EPI, NT, WMW>), dim3(tm * tn), dim3(256 * WMW), shmem, st, a, lda, w, ldw, bias,
                       r, c, ldc, M, N, K, tn);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

// which kernel runs a shape: 0 the register-staged 128 x 128 kernel, 1 a ring kernel, 2 the persistent phase-interleaved 256 x 256 kernel
static int pick_bf16_kernel(int M, int N, int K, int ldc) {
    if (!(M >= 512 && N >= 128 && N % 8 == 0 && ldc % 8 == 0 && K % 64 == 0) || tune(TUNE_BF16_GEMM_SMALL)) return 0;
    const int force = tune(TUNE_BF16_GEMM_TILE);
    const int64_t blocks256 = (int64_t)ceil_div(M, 256) * ceil_div(N, 256);
    // 256-wide N tiles unless that leaves too few workgroups for 256 CUs.  (From 192 tiles on, round 4: the packed DistilBERT batch -- 18.8 k
    // real tokens of [256, 128] -- is 74 x 3 = 222 tiles at N = 768, and only the persistent kernel has the folded-LayerNorm epilogues.)
    return (force == 4 || (force == 0 && N % 256 == 0 && blocks256 >= 192)) ? 2 : 1;
}

template <int EPI>
static int launch_glds_pick(const bf16_t* a, int lda, const bf16_t* w, int ldw, const float* bias, const bf16_t* r, bf16_t* c,
                            int ldc, int M, int N, int K, hipStream_t st, GemmBf16Info* info) {
    const int force = tune(TUNE_BF16_GEMM_TILE);   // 1: 128x128 / 2: 256x128 / 3: 256x256 ring kernels (tools/gemm_bf16_bench.py)
    if (force == 5 && EPI != BEPI_BIAS_RES && K % 64 == 0 && N % 8 == 0) {   // (prototype switch: the two-workgroups-per-CU kernel)
        if (info) { info->kernel = 3; info->half_tiles = 0; }
        return launch_duo<EPI, bf16_t>(a, lda, w, ldw, bias, r, c, ldc, M, N, K, st);
    }
    if (pick_bf16_kernel(M, N, K, ldc) == 2) return launch_ph<EPI>(a, lda, w, ldw, bias, r, c, ldc, M, N, K, st, info);
    if (info) { info->kernel = 1; info->half_tiles = 0; }
    if (force == 1) return launch_glds<EPI, 2, 1>(a, lda, w, ldw, bias, r, c, ldc, M, N, K, st);
    if (force == 2) return launch_glds<EPI, 2, 2>(a, lda, w, ldw, bias, r, c, ldc, M, N, K, st);
    if (force == 3) return launch_glds<EPI, 4, 2>(a, lda, w, ldw, bias, r, c, ldc, M, N, K, st);
    const int64_t blocks256 = (int64_t)ceil_div(M, 256) * ceil_div(N, 256);
    if (N % 256 == 0 && blocks256 >= 512) return launch_glds<EPI, 4, 2>(a, lda, w, ldw, bias, r, c, ldc, M, N, K, st);
    return launch_glds<EPI, 2, 2>(a, lda, w, ldw, bias, r, c, ldc, M, N, K, st);
}

bool gemm_bf16_is_persistent(int M, int N, int K) { return pick_bf16_kernel(M, N, K, 8) == 2; }

// fp16 operands (the decoder's big-batch prefill): the persistent kernel only, epilogues 3 / 4 / 5 and the fp32-output head (6)
static int launch_gemm_f16(const void* A, int lda, const void* W, int ldw, const float* bias, const void* res, void* C, int ldc, int M, int N,
                           int K, int epi, hipStream_t st, GemmBf16Info* info, const BfEpiLn* lnp) {
    typedef _Float16 H;
    const H *a = (const H*)A, *w = (const H*)W, *r = (const H*)res;
    H* c = (H*)C;
    MGEA_REQUIRE(M >= 512 && N >= 256 && (int64_t)ceil_div(M, 256) * ceil_div(N, 256) >= 8, MGEA_EINVAL,
                 "fp16 gemm: M=%d N=%d below what the persistent kernel takes", M, N);
    if (epi == BEPI_BIAS_F32) {
        MGEA_REQUIRE(bias && N % 4 == 0 && ldc % 4 == 0, MGEA_EINVAL, "fp16 gemm: fp32-output epilogue needs a bias and N, ldc multiples of 4");
        return launch_ph<BEPI_BIAS_F32, H>(a, lda, w, ldw, bias, r, c, ldc, M, N, K, st, info, lnp);
    }
    MGEA_REQUIRE(epi >= BEPI_LNFOLD && epi <= BEPI_RES_LN && lnp && bias && N % 256 == 0 && ldc % 8 == 0, MGEA_EINVAL,
                 "fp16 gemm: epilogue %d / shape not built (3, 4, 5 with N %% 256 == 0, or 6)", epi);
    if (epi == BEPI_RES_LN) {
        MGEA_REQUIRE(res && lnp->rowstat && lnp->ln_g && lnp->ln_b, MGEA_EINVAL, "fp16 gemm: RES_LN without residual / row statistics / gamma / beta");
        return launch_ph<BEPI_RES_LN, H>(a, lda, w, ldw, bias, r, c, ldc, M, N, K, st, info, lnp);
    }
    MGEA_REQUIRE(lnp->rowstat && lnp->c1, MGEA_EINVAL, "fp16 gemm: LNFOLD without row statistics / c1");
    if (epi == BEPI_LNFOLD) return launch_ph<BEPI_LNFOLD, H>(a, lda, w, ldw, bias, r, c, ldc, M, N, K, st, info, lnp);
    return launch_ph<BEPI_LNFOLD_GELU, H>(a, lda, w, ldw, bias, r, c, ldc, M, N, K, st, info, lnp);
}

int launch_gemm_bf16(const void* A, int lda, const void* W, int ldw, const float* bias, const void* res, void* C,
                     int ldc, int M, int N, int K, int epi, hipStream_t st, GemmBf16Info* info, const BfEpiLn* lnp, int f16) {
    MGEA_REQUIRE(M > 0 && N > 0 && K > 0 && K % 64 == 0 && N % 4 == 0 && lda % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0,
                 MGEA_EINVAL, "bf16 gemm: bad shape M=%d N=%d K=%d (K %% 64, N %% 4)", M, N, K);
    if (f16) return launch_gemm_f16(A, lda, W, ldw, bias, res, C, ldc, M, N, K, epi, st, info, lnp);
    if (epi >= BEPI_LNFOLD) {   // LayerNorm folded around the GEMM: the persistent kernel only
        MGEA_REQUIRE(epi <= BEPI_RES_LN && lnp && bias, MGEA_EINVAL, "bf16 gemm: epilogue %d needs its LayerNorm operands and a bias / c2 vector", epi);
        MGEA_REQUIRE(gemm_bf16_is_persistent(M, N, K) && ldc % 8 == 0 && N % 256 == 0, MGEA_EINVAL,
                     "bf16 gemm: epilogue %d exists on the persistent 256 x 256 kernel only (M=%d N=%d K=%d)", epi, M, N, K);
        const bf16_t *a = (const bf16_t*)A, *w = (const bf16_t*)W, *r = (const bf16_t*)res;
        bf16_t* c = (bf16_t*)C;
        if (epi == BEPI_RES_LN) {
            MGEA_REQUIRE(res && lnp->rowstat && lnp->ln_g && lnp->ln_b, MGEA_EINVAL,
                         "bf16 gemm: RES_LN without residual / row statistics / gamma / beta (identity tables for a normalised residual)");
            return launch_ph<BEPI_RES_LN>(a, lda, w, ldw, bias, r, c, ldc, M, N, K, st, info, lnp);
        }
        MGEA_REQUIRE(lnp->rowstat && lnp->c1, MGEA_EINVAL, "bf16 gemm: LNFOLD without row statistics / c1");
        if (epi == BEPI_LNFOLD) return launch_ph<BEPI_LNFOLD>(a, lda, w, ldw, bias, r, c, ldc, M, N, K, st, info, lnp);
        return launch_ph<BEPI_LNFOLD_GELU>(a, lda, w, ldw, bias, r, c, ldc, M, N, K, st, info, lnp);
    }
    if (pick_bf16_kernel(M, N, K, ldc) != 0) {
        const bf16_t *a = (const bf16_t*)A, *w = (const bf16_t*)W, *r = (const bf16_t*)res;
        bf16_t* c = (bf16_t*)C;
        if (epi == BEPI_BIAS) return launch_glds_pick<BEPI_BIAS>(a, lda, w, ldw, bias, r, c, ldc, M, N, K, st, info);
        if (epi == BEPI_BIAS_GELU) return launch_glds_pick<BEPI_BIAS_GELU>(a, lda, w, ldw, bias, r, c, ldc, M, N, K, st, info);
        if (epi == BEPI_BIAS_RES && res) return launch_glds_pick<BEPI_BIAS_RES>(a, lda, w, ldw, bias, r, c, ldc, M, N, K, st, info);
    }
    if (info) { info->kernel = 0; info->half_tiles = 0; }
    const int tm = ceil_div(M, 128), tn = ceil_div(N, 128);
    dim3 grid(tm * tn), block(256);
    const bf16_t *a = (const bf16_t*)A, *w = (const bf16_t*)W, *r = (const bf16_t*)res;
    bf16_t* c = (bf16_t*)C;
    switch (epi) {
        case BEPI_BIAS: hipLaunchKernelGGL(gemm_bf16_nt_kernel<BEPI_BIAS>, grid, block, 0, st, a, lda, w, ldw, bias, r, c, ldc, M, N, K, tn); break;
        case BEPI_BIAS_GELU: hipLaunchKernelGGL(gemm_bf16_nt_kernel<BEPI_BIAS_GELU>, grid, block, 0, st, a, lda, w, ldw, bias, r, c, ldc, M, N, K, tn); break;
        case BEPI_BIAS_RES:
            MGEA_REQUIRE(res, MGEA_EINVAL, "bf16 gemm: residual epilogue without residual");
            hipLaunchKernelGGL(gemm_bf16_nt_kernel<BEPI_BIAS_RES>, grid, block, 0, st, a, lda, w, ldw, bias, r, c, ldc, M, N, K, tn);
            break;
        default: MGEA_REQUIRE(false, MGEA_EINVAL, "bf16 gemm: bad epilogue %d", epi);
    }
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

// ------------------------------------------------------------------------------------------
// Helpers of the folded-LayerNorm pipeline (BfEpiLn).
// (mean, rstd) per row from the per-tile (sum, M2 about the tile's own mean) pairs the RES_LN epilogue left, each over C / n_part
// columns: mean = sum of sums / C, M2 = sum_i M2_i + cnt sum_i (mean_i - mean)^2 (the exact pairwise merge), in double.
// A launch of its own (5 us): deriving the statistics inside the consumers instead (8-16 more loaded values live per lane in their
// epilogues) pushed 20-50 registers of the persistent kernel to scratch and cost 0.45 ms per forward.
__global__ void ln_rowstat_kernel(const float* __restrict__ part, float* __restrict__ rowstat, int M, int n_part, int C, float eps) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= M) return;
    const double cnt = (double)C / n_part;
    double s1 = 0.0;
    for (int i = 0; i < n_part; ++i) s1 += part[((int64_t)row * n_part + i) * 2];
    const double mean = s1 / C;
    double m2 = 0.0;
    for (int i = 0; i < n_part; ++i) {
        const double d = part[((int64_t)row * n_part + i) * 2] / cnt - mean;
        m2 += part[((int64_t)row * n_part + i) * 2 + 1] + cnt * d * d;
    }
    rowstat[(int64_t)row * 2] = (float)mean;
    rowstat[(int64_t)row * 2 + 1] = (float)(1.0 / sqrt(m2 / C + (double)eps));
}
int launch_ln_rowstat(const float* part, float* rowstat, int M, int n_part, int C, float eps, hipStream_t st) {
    hipLaunchKernelGGL(ln_rowstat_kernel, dim3(ceil_div(M, 256)), dim3(256), 0, st, part, rowstat, M, n_part, C, eps);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

// W' = bf16(W diag(gamma)), c1[n] = sum_k W'[n, k] (of the ROUNDED folded weights: the epilogue subtracts exactly what the MFMAs
// added), c2[n] = b[n] + sum_k W[n, k] beta[k].  One workgroup per output row n.
template <typename T>
__global__ __launch_bounds__(256) void fold_ln_weights_bf16_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, const float* __restrict__ b,
                                                                  T* __restrict__ Wf, float* __restrict__ c1,
                                                                  float* __restrict__ c2, int K) {
    const int64_t n = blockIdx.x;
    float s1 = 0.f, s2 = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float w = W[n * K + k];
        const T wf = (T)(w * gamma[k]);
        Wf[n * K + k] = wf;
        s1 += (float)wf;
        s2 = fmaf(w, beta[k], s2);
    }
    __shared__ float r1[4], r2[4];
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = s1; r2[threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        c1[n] = (r1[0] + r1[1]) + (r1[2] + r1[3]);
        c2[n] = b[n] + ((r2[0] + r2[1]) + (r2[2] + r2[3]));
    }
}
int launch_fold_ln_weights_bf16(const float* W, const float* gamma, const float* beta, const float* b, void* Wf, float* c1, float* c2,
                                int N, int K, hipStream_t st, int f16) {
    if (f16) hipLaunchKernelGGL(fold_ln_weights_bf16_kernel<_Float16>, dim3(N), dim3(256), 0, st, W, gamma, beta, b, (_Float16*)Wf, c1, c2, K);
    else     hipLaunchKernelGGL(fold_ln_weights_bf16_kernel<bf16_t>, dim3(N), dim3(256), 0, st, W, gamma, beta, b, (bf16_t*)Wf, c1, c2, K);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

// CLS rows (row b * S) of the RAW last-layer output -> LayerNorm with the row's (mean, rstd) -> fp32 [B, D]
__global__ void gather_cls_ln_bf16_kernel(const bf16_t* __restrict__ h, const float* __restrict__ rowstat, const float* __restrict__ g,
                                          const float* __restrict__ be, float* __restrict__ out, int S, int D, const int32_t* __restrict__ cu) {
    const int64_t row = cu ? (int64_t)cu[blockIdx.x] : (int64_t)blockIdx.x * S;      // the [CLS] row of sequence b (packed input: its first row)
    const float mean = rowstat[row * 2], rstd = rowstat[row * 2 + 1];
    for (int d = threadIdx.x; d < D; d += blockDim.x) out[(int64_t)blockIdx.x * D + d] = fmaf(((float)h[row * D + d] - mean) * rstd, g[d], be[d]);
}
int launch_gather_cls_ln_bf16(const void* h, const float* rowstat, const float* g, const float* be, float* out, int B, int S, int D,
                              hipStream_t st, const int32_t* cu) {
    hipLaunchKernelGGL(gather_cls_ln_bf16_kernel, dim3(B), dim3(256), 0, st, (const bf16_t*)h, rowstat, g, be, out, S, D, cu);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

// ------------------------------------------------------------------------------------------
// LayerNorm over bf16 rows (fp32 statistics), in place allowed.  A wave owns 4 rows and requests all of them (and the affine
// vectors) before the first reduction: 8-16 loads of 16 bytes in flight per lane instead of 2, four independent reduction chains.
// (One row per wave ran at 4.2 TB/s of read + write on [32768, 768]: every wave sat through a load, two dependent wave reductions
// and a second round trip for w / b before its only stores.)
template <int NCHL>   // 16-byte chunks per lane and row: C <= 512 * NCHL
__global__ __launch_bounds__(256) void layernorm_bf16_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, bf16_t* __restrict__ y, int M,
                                                            int C, float eps) {
    constexpr int RW = 4;
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
    if (row0 >= M) return;
    const int nch = C >> 3;  // 16-byte chunks per row
    bf16x8 t[RW][NCHL];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int64_t row = row0 + r < M ? row0 + r : M - 1;     // clamped: loaded, never stored
#pragma unroll
        for (int i = 0; i < NCHL; ++i) {
            const int ch = lane + i * 64;
            t[r][i] = *reinterpret_cast<const bf16x8*>(x + row * C + (ch < nch ? ch : 0) * 8);
        }
    }
    float4 wv[NCHL][2], bv[NCHL][2];
#pragma unroll
    for (int i = 0; i < NCHL; ++i) {
        const int ch = lane + i * 64 < nch ? lane + i * 64 : 0;
        wv[i][0] = ld4(w + ch * 8); wv[i][1] = ld4(w + ch * 8 + 4);
        bv[i][0] = ld4(b + ch * 8); bv[i][1] = ld4(b + ch * 8 + 4);
    }
    const float inv_c = 1.0f / (float)C;
    float mean[RW], rstd[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NCHL; ++i)
            if (lane + i * 64 < nch)
#pragma unroll
                for (int j = 0; j < 8; ++j) s += (float)t[r][i][j];
        mean[r] = s;
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) mean[r] = wave_sum(mean[r]) * inv_c;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NCHL; ++i)
            if (lane + i * 64 < nch)
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = (float)t[r][i][j] - mean[r]; q += d * d; }
        rstd[r] = q;
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) rstd[r] = 1.0f / sqrtf(wave_sum(rstd[r]) * inv_c + eps);
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        if (row0 + r >= M) break;
#pragma unroll
        for (int i = 0; i < NCHL; ++i) {
            const int ch = lane + i * 64;
            if (ch < nch) {
                const float ww[8] = {wv[i][0].x, wv[i][0].y, wv[i][0].z, wv[i][0].w, wv[i][1].x, wv[i][1].y, wv[i][1].z, wv[i][1].w};
                const float bb[8] = {bv[i][0].x, bv[i][0].y, bv[i][0].z, bv[i][0].w, bv[i][1].x, bv[i][1].y, bv[i][1].z, bv[i][1].w};
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16_t)(((float)t[r][i][j] - mean[r]) * rstd[r] * ww[j] + bb[j]);
                *reinterpret_cast<bf16x8*>(y + (row0 + r) * C + ch * 8) = o;
            }
        }
    }
}
int launch_layernorm_bf16(const void* x, const float* w, const float* b, void* y, int M, int C, float eps, hipStream_t st) {
    MGEA_REQUIRE(C % 8 == 0 && C <= 2048, MGEA_EINVAL, "bf16 layernorm: C=%d must be a multiple of 8 and <= 2048", C);
    const dim3 grid(ceil_div(M, 16));
    if (C <= 1024) hipLaunchKernelGGL(layernorm_bf16_kernel<2>, grid, dim3(256), 0, st, (const bf16_t*)x, w, b, (bf16_t*)y, M, C, eps);
    else           hipLaunchKernelGGL(layernorm_bf16_kernel<4>, grid, dim3(256), 0, st, (const bf16_t*)x, w, b, (bf16_t*)y, M, C, eps);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

// h[m] = LN(word[ids[m]] + pos[t]) -> bf16 (fp32 tables)
__global__ __launch_bounds__(256) void bert_embed_ln_bf16_kernel(const int32_t* __restrict__ ids, const float* __restrict__ word,
                                                                const float* __restrict__ pos, const float* __restrict__ lnw,
                                                                const float* __restrict__ lnb, float eps,
                                                                bf16_t* __restrict__ h, int M, int S, int D, int vocab,
                                                                int32_t* __restrict__ err_flag, const int32_t* __restrict__ pos_ids) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int t = pos_ids ? pos_ids[row] : (int)(row % S);      // packed input carries each row's position in its sequence
    int id = ids[row];
    if ((id < 0 || id >= vocab) && err_flag && lane == 0) atomicOr(err_flag, 1);   // clamped + reported (mgea_bert_error_flags)
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const int nf4 = D >> 2;
    float4 v[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int f = lane + i * 64;
        if (f < nf4) {
            v[i] = add4(ld4(word + (int64_t)id * D + f * 4), ld4(pos + (int64_t)t * D + f * 4));
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (lane + i * 64 < nf4) {
            const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int f = lane + i * 64;
        if (f < nf4) {
            const float4 ww = ld4(lnw + f * 4), bb = ld4(lnb + f * 4);
            bf16x4 o = {(bf16_t)((v[i].x - mean) * rstd * ww.x + bb.x), (bf16_t)((v[i].y - mean) * rstd * ww.y + bb.y),
                        (bf16_t)((v[i].z - mean) * rstd * ww.z + bb.z), (bf16_t)((v[i].w - mean) * rstd * ww.w + bb.w)};
            *reinterpret_cast<bf16x4*>(h + row * D + f * 4) = o;
        }
    }
}
int launch_bert_embed_ln_bf16(const int32_t* ids, const float* word, const float* pos, const float* lnw, const float* lnb,
                              float eps, void* h, int B, int S, int D, int vocab, hipStream_t st, int32_t* err_flag, const int32_t* pos_ids) {
    MGEA_REQUIRE(D % 4 == 0 && D <= 2048, MGEA_EINVAL, "bf16 embed: dim=%d must be a multiple of 4 and <= 2048", D);
    // (packed input: B = the number of rows, S = 1)
    hipLaunchKernelGGL(bert_embed_ln_bf16_kernel, dim3(ceil_div(B * S, 4)), dim3(256), 0, st, ids, word, pos, lnw, lnb, eps,
                       (bf16_t*)h, B * S, S, D, vocab, err_flag, pos_ids);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

// CLS rows (row b*S) of a bf16 [B*S, D] matrix -> fp32 [B, D] for the small fp32 classifier head
__global__ void gather_cls_bf16_kernel(const bf16_t* __restrict__ h, float* __restrict__ out, int S, int D, const int32_t* __restrict__ cu) {
    const int64_t b = blockIdx.x, row = cu ? (int64_t)cu[b] : b * S;
    for (int d = threadIdx.x; d < D; d += blockDim.x) out[b * D + d] = (float)h[row * D + d];
}
int launch_gather_cls_bf16(const void* h, float* out, int B, int S, int D, hipStream_t st, const int32_t* cu) {
    hipLaunchKernelGGL(gather_cls_bf16_kernel, dim3(B), dim3(256), 0, st, (const bf16_t*)h, out, S, D, cu);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

// ------------------------------------------------------------------------------------------
// Reductions over the four lanes c, c + 16, c + 32, c + 48 (the lanes that share a query in the K Q^T accumulator) with gfx950's
// v_permlane32_swap / v_permlane16_swap: VALU instructions, no trip through the LDS pipeline like ds_bpermute (__shfl_xor), which
// matters at 2 waves per SIMD where nothing hides that latency.  swap(x, x) leaves [lo, lo] / [hi, hi] (32) or [r0 r0 r2 r2] / [r1 r1
// r3 r3] (16) in the two results, so one max of the pair is the xor-32 / xor-16 butterfly step.
// max(a, b) as the median of (a, b, +inf): ONE v_med3_f32.  fmaxf() on values that come out of an MFMA or a lane swap costs two extra
// instructions each -- hipcc canonicalises both inputs first (v_max_f32 x, x, x: IEEE maxNum semantics for signalling NaNs), 16 + 8 of them
// per 64-key tile of the flash attention.  No score is a NaN here; -inf (masked keys) orders as usual.
__device__ __forceinline__ float max_nc(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, INFINITY); }
__device__ __forceinline__ float quad_max(float x) {
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float m = max_nc(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    return max_nc(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

// ------------------------------------------------------------------------------------------
// Flash attention, bf16 in/out, head_dim 64.  PERSISTENT and software-pipelined: 2 workgroups per CU walk the work items
// (b, h, block of 128 queries); 4 waves x 32 queries (two 16-query MFMA column blocks per wave, every V fragment read from LDS feeds
// two MFMAs).  Keys are staged 128 at a time -- for the DistilBERT shape (T = 128) K and V of a (b, h) leave HBM exactly once --
// by LDS-DMA into one of two 32 KB stages: the DMA of unit u + 1 is issued right after the barrier that opens unit u and flies under
// u's MFMAs, softmax and output stores.  (Without it every workgroup of a round loads, computes and stores in step with all the
// others: 54 us = 27 load + 16 + 11, the phases add up.)  The DMA is issued from inline asm (glds16_hidden) so that hipcc does not
// drain vmcnt in front of every LDS read; the only wait for it is the explicit vmcnt(0) at the top of a unit, where nothing younger
// is in flight.
//   K image [key][8 x 16 B], chunk ^= key & 7            : row reads (ds_read_b128) for the A operand of K Q^T, conflict-free
//   V image [key][8 x 16 B], chunk ^= ((key >> 1) & 3) * 2: read with ds_read_b64_tr_b16, the hardware transpose (guide T10): lanes
//      16g..16g+15 fetch a block of 4 keys x 16 d and lane c gets column d = 16 dt + c of the 4 keys -- the V^T A-operand of the
//      P V product without a transposing write pass.  The 8 rows a 32-lane half touches land on 8 different 8-bank groups.
//   Both swizzles are applied on the GLOBAL side of the DMA (a lane picks which 16 bytes of its key's row it fetches); the LDS side
//   of a DMA is lane-linear.  EXEC is all ones at every transposed read (uniform control flow only).
// S^T = K Q^T keeps a query's scores lane-local (lane = query, registers = keys 16 kt + 4 g + r); the k index of the P V product is
// permuted to match (k = 8 g + j  <->  key 32 p + 16 (j >> 2) + 4 g + (j & 3)), so P goes from accumulator to operand in registers.
// The output tile goes through the (then free) K image so that every store instruction writes complete 128-byte rows.
// NW waves x 32 queries per workgroup, keys staged KB = 32 NW at a time: <4, 128> (two workgroups per CU) for short sequences -- DistilBERT: K and V
// of a (batch, head) leave HBM exactly once --, <8, 256> (one workgroup per CU, 2 x 64 KB stages) for long ones, where the per-unit costs (the
// wait for the DMA, the barrier, 8 LDS-DMA pieces per wave: 1.7 of 4.3 us per unit by in-kernel stamps at 1024 keys) are spread over four
// 64-key tiles instead of two.
template <typename E, int NW, int PIPE>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2)))
void attn_bf16_kernel(const E* __restrict__ qkv, const int32_t* __restrict__ mask, E* __restrict__ out, int T, int H,
                      int n_items, int nqb, float scale, KvPages pg, const int32_t* __restrict__ cu) {
    typedef typename X16<E>::v8 bf16x8;   // (the names below were written for bf16; E may be _Float16)
    typedef typename X16<E>::v4 bf16x4;
    typedef E bf16_t;
    constexpr int DH = 64, KB = 32 * NW, QPB = 32 * NW;           // keys per stage, queries per workgroup
    constexpr int STAGE = KB * 16;                                // 16-byte chunks: K image [KB][8], then V image [KB][8]
    constexpr int VW = KB / 64;                                   // 64-bit validity words per stage
    extern __shared__ __attribute__((aligned(16))) float4 lds[];  // [2][STAGE], then [2][VW] 64-bit validity words
    unsigned long long* sValid = reinterpret_cast<unsigned long long*>(lds + 2 * STAGE);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float4*)lds;

    const int C = H * DH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int nkb = (T + KB - 1) / KB;
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

    // DMA of one unit (item, key block) into a stage: wave w moves keys 32 w .. 32 w + 31 of K and of V (8 instructions of 1 KB)
    const int d_key = lane >> 3, d_ch = lane & 7;                 // this lane's slot inside an 8-key piece
    const unsigned dk_off = 2u * C + 16u * (d_ch ^ d_key);                              // K: chunk ^ (key & 7), key & 7 == d_key
    const unsigned dv_off = 4u * C + 16u * (d_ch ^ (((d_key >> 1) & 3) << 1));          // V: chunk ^ 2 ((key >> 1) & 3)
    // PACKED ("varlen") INPUT, cu != NULL (round 4, DistilBERT on unpadded batches): the rows of sequence b are cu[b] .. cu[b + 1] - 1 of one
    // [sum of lengths, 3 C] buffer, no padding rows and no key mask; every sequence fits one query block and one key stage (host check:
    // max length <= KB, so nqb == nkb == 1 and an item is a (sequence, head)).  T then only sizes the grid; a sequence's own length
    // bounds its key tiles, so a 20-token prompt computes one 64-key tile where the padded form computes the batch's maximum.
    auto seq_rows = [&](int bb, int& row0, int& Tb) {            // wave-uniform (scalar loads)
        if (cu) { row0 = cu[bb]; Tb = cu[bb + 1] - row0; } else { row0 = bb * T; Tb = T; }
    };
    struct ItemBase { const bf16_t* p; int Tb; };
    auto item_base = [&](int it) {                                // first qkv row of the item's (batch, head): wave-uniform, stays in SGPRs
        const int bh = it / nqb, bb = bh / H, hh = bh - bb * H;
        int row0, Tb;
        seq_rows(bb, row0, Tb);
        return ItemBase{qkv + (int64_t)row0 * 3 * C + hh * DH, Tb};
    };
    auto issue_part = [&](const ItemBase& ib, int kbi, int st, int i0, int i1) {   // pieces i0 .. i1 - 1 of this wave's four (K and V of 8 keys each)
        const bf16_t* base = ib.p;
#pragma unroll
        for (int i = i0; i < i1; ++i) {
            const int piece = wave * 4 + i, key = piece * 8 + d_key;
            int kr = kbi * KB + key; kr = kr < ib.Tb ? kr : ib.Tb - 1;
            const unsigned row = (unsigned)kr * (unsigned)(6 * C);    // bytes; a sequence's rows span < 4 GB (checked on the host)
            const unsigned dst = lds_base + (unsigned)(st * STAGE + piece * 64) * 16u;
            glds16_hidden_s(base, row + dk_off, dst);
            glds16_hidden_s(base, row + dv_off, dst + (unsigned)(KB * 8) * 16u);
        }
    };
    auto issue = [&](int it, int kbi, int st) { issue_part(item_base(it), kbi, st, 0, 4); };
    auto mask_of = [&](int it, int kbi) -> int {                  // validity of key kbi * KB + tid (the first KB / 64 waves)
        const int bh = it / nqb, bb = bh / H;
        const int kidx = kbi * KB + tid;
        int row0, Tb;
        seq_rows(bb, row0, Tb);
        int ok = (tid < KB && kidx < Tb) ? 1 : 0;
        if (ok && mask) ok = mask[(int64_t)bb * T + kidx];        // (packed input has no mask: host check)
        return ok;
    };
    bf16x8 qn[2][2];                                              // next item's raw Q fragments
    auto load_q = [&](int it) {
        const int bh = it / nqb, qb = it - bh * nqb, bb = bh / H, hh = bh - bb * H;
        int row0, Tb;
        seq_rows(bb, row0, Tb);
#pragma unroll
        for (int mq = 0; mq < 2; ++mq) {
            int qr = qb * QPB + wave * 32 + mq * 16 + c; qr = qr < Tb ? qr : Tb - 1;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                qn[mq][ks] = *reinterpret_cast<const bf16x8*>(qkv + ((int64_t)row0 + qr) * 3 * C + hh * DH + ks * 32 + 8 * g);
        }
    };
    // transposed-read offsets (bf16 elements) of this lane inside a 16-key group of the V image, one per 16-d block:
    // row 4 g + (c >> 2) of the group, 16-byte chunk (2 dt + ((c & 3) >> 1)) ^ swizzle(row), half (c & 1)
    int v_off[4];
    {
        const int vrow = 4 * g + (c >> 2), sw = ((vrow >> 1) & 3) << 1;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) v_off[dt] = vrow * 64 + (((2 * dt + ((c & 3) >> 1)) ^ sw) * 8) + 4 * (c & 1);
    }

    // The output rows of an item are held in registers and stored one unit later, behind the next DMA issue: a store issued at the
    // end of a unit would be waited for (vmcnt(0), loads and stores share the counter) at the top of the next one with nothing to hide
    // its latency behind -- 2-3 us per item.
    float4 ov0, ov1, ov2, ov3;                                    // (scalars, not an array: an array captured by a lambda went to scratch)
    bf16_t* optr = nullptr;                                       // row of ov0; ov<i> is 8 i rows further; nullptr: nothing held
    int orow = 0, oT = 0;                                         // query index of ov0 within the sequence, and that sequence's length
#define MGEA_FLUSH_OUT()                                                                               \
    if (optr) {                                                                                        \
        if (orow < oT)      *reinterpret_cast<float4*>(optr) = ov0;                                    \
        if (orow + 8 < oT)  *reinterpret_cast<float4*>(optr + (int64_t)8 * C) = ov1;                   \
        if (orow + 16 < oT) *reinterpret_cast<float4*>(optr + (int64_t)16 * C) = ov2;                  \
        if (orow + 24 < oT) *reinterpret_cast<float4*>(optr + (int64_t)24 * C) = ov3;                  \
    }
    int item = blockIdx.x, kb = 0, stage = 0, mk = 0;
    if (item < n_items) { issue(item, 0, 0); mk = mask_of(item, 0); load_q(item); }
    bf16x8 qf[2][2];
    f32x4 oacc[2][4], lacc[2];                                    // lacc: row sums of P, from an all-ones A operand (every row equal)
    float mx[2];                                                  // running maximum of the RAW scores q . k
    const float kexp = scale * 1.4426950408889634f;               // exp(scale * (s - m)) = 2^((s - m) * kexp)
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16_t)1.0f;
    [[maybe_unused]] int un = 0;                                   // (tools-only stamps: unit counter)
    while (item < n_items) {
        PH_STAMP(un * 8 + 0);
        // unit (item, kb) has landed: nothing younger than its DMA, its mask word and (kb == 0) its Q rows is in flight here
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PH_STAMP(un * 8 + 1);
        // Tell hipcc that the prefetched registers are complete on every path.  It does not read the wait above, and a load it still
        // believes pending makes it wait (by ITS count, which does not include the hidden DMAs: in effect vmcnt(0)) before those
        // registers are written again -- right after the next unit's DMA was issued, which would serialise the pipeline.
        asm volatile("" : "+v"(qn[0][0]), "+v"(qn[0][1]), "+v"(qn[1][0]), "+v"(qn[1][1]), "+v"(mk));
        int row0cur, Tcur;                                        // the current item's sequence: first row, length
        seq_rows((item / nqb) / H, row0cur, Tcur);
        if (tid < KB) {
            const unsigned long long bal = __ballot(mk != 0);
            if (lane == 0) sValid[stage * VW + wave] = bal;
        }
        __syncthreads();
        PH_STAMP(un * 8 + 2);
        if (kb == 0) {
#pragma unroll
            for (int mq = 0; mq < 2; ++mq) {
                mx[mq] = -INFINITY;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) oacc[mq][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                lacc[mq] = (f32x4){0.f, 0.f, 0.f, 0.f};
                qf[mq][0] = qn[mq][0];                               // unscaled: 1/sqrt(dh) is folded into the exponent's constant
                qf[mq][1] = qn[mq][1];
            }
        }
        // K | V OF THE REAL TOKENS -> fp16 KV PAGES (decoder prefill, round 4; replaces kv_scatter_f16_kernel, which re-read the K | V columns
        // of the qkv buffer from HBM -- 134 MB in, 134 MB out per layer at [64, 1024]).  The stage that has just landed holds exactly
        // those rows: ONE of the (batch, head)'s items writes them out (below), each wave the 32 keys it moved in -- K as
        // [d-group][token][8 halves] (lane = token: 512 contiguous bytes per d-group and instruction), V as whole 128-byte rows.
        // Only keys with their validity bit set (t < lens[b], t < T) are cached, at position t (the cache is empty: run_prefill16).
        if constexpr (sizeof(E) == 2 && !X16<E>::is_bf16) {
            if (pg.pool.base) {
                const int bh_u = item / nqb, qb_u = item - bh_u * nqb;
                if (qb_u == kb) {     // (nqb == nkb: queries per item = keys per stage.  The item of query block kb writes key block kb -- every
                                      //  workgroup a quarter of the stages; with query block 0 writing all of them a quarter of the workgroups
                                      //  did all the writing and the launch waited for them: 4.64 ms per cache fill against 4.37 with the scatter kernel)
                    const int bb = bh_u / H, hh = bh_u - bb * H;
                    const int tok0 = kb * KB + wave * 32;                                   // 32-aligned: one page
                    const unsigned long long vw = sValid[stage * VW + (wave >> 1)];
                    const unsigned vbits = (unsigned)(vw >> (32 * (wave & 1)));             // validity of this wave's 32 keys
                    if (tok0 < Tcur && (tok0 >> 6) < pg.max_pages) {
                        const int phys = pg.page_table[bb * pg.max_pages + (tok0 >> 6)];
                        const int64_t pe = pg.pool.page_elems();
                        _Float16* kpage = static_cast<_Float16*>(pg.pool.base) + pg.layer * pg.pool.layer_stride + ((int64_t)(phys * 2) * H + hh) * pe;
                        _Float16* vpage = kpage + (int64_t)H * pe;
                        const int slot0 = tok0 & 63;
                        const float4* sKs = lds + stage * STAGE;
                        const float4* sVs = sKs + KB * 8;
                        {   // K: lane -> (token lane & 31, d-group 2 i + (lane >> 5))
                            const int tk = lane & 31, key = wave * 32 + tk;
                            const bool ok = (vbits >> tk) & 1u;
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int j = 2 * i + (lane >> 5);
                                const float4 v = sKs[key * 8 + (j ^ (key & 7))];
                                if (ok) *reinterpret_cast<float4*>(kpage + ((int64_t)j * 64 + slot0 + tk) * 8) = v;
                            }
                        }
                        {   // V: lane -> (token 8 i + (lane >> 3), 16-byte chunk lane & 7): 1 KB of consecutive bytes per instruction
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int tk = 8 * i + (lane >> 3), key = wave * 32 + tk, ch = lane & 7;
                                const float4 v = sVs[key * 8 + (ch ^ (((key >> 1) & 3) << 1))];
                                if ((vbits >> tk) & 1u) *reinterpret_cast<float4*>(vpage + ((int64_t)(slot0 + tk) * 64 + ch * 8)) = v;
                            }
                        }
                    }
                }
            }
        }
        int nitem = item, nkbi = kb + 1;
        if (nkbi == nkb) { nitem = item + gridDim.x; nkbi = 0; }
        // PIPE, full stage: the next unit's 8 LDS-DMA pieces are issued inside the tile loop below, a quarter per tile, instead of as a
        // burst here (stamps, round 3: 0.85 us per unit during which all 8 waves only issue and the matrix pipe idles)
        const bool dma_in_loop = PIPE && NW == 8 && (kb + 1) * KB <= Tcur;
        const ItemBase nbase = item_base(nitem < n_items ? nitem : item);   // (two integer divisions: once per unit, not once per piece)
        if (nitem < n_items) {                                    // the other stage was last read before the barrier above
            if (!dma_in_loop) issue(nitem, nkbi, stage ^ 1);
            mk = mask_of(nitem, nkbi);
            if (nkbi == 0) load_q(nitem);
        }
        MGEA_FLUSH_OUT();
        optr = nullptr;
        PH_STAMP(un * 8 + 3);

        const float4* sK = lds + stage * STAGE;
        const bf16_t* sV = reinterpret_cast<const bf16_t*>(lds + stage * STAGE + KB * 8);
        // ---- one 64-key tile in three pieces (lambdas, so that the two loop forms below share them) ----
        // S^T = K Q^T of keys t0 .. t0 + 63 against the wave's 32 queries: two independent chains per key tile (both query blocks)
        auto qk_tile = [&](int t0, f32x4 (&sc)[2][4]) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) { sc[0][kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; sc[1][kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const int key = t0 + kt * 16 + c;
                    const float4 kv = sK[key * 8 + ((ks * 4 + g) ^ (key & 7))];
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(&kv);
                    sc[0][kt] = X16<E>::mfma(kf, qf[0][ks], sc[0][kt]);
                    sc[1][kt] = X16<E>::mfma(kf, qf[1][ks], sc[1][kt]);
                }
        };
        // key mask, tile maximum, and the (rare) move of the scaling reference with its accumulator rescale
        auto tile_max = [&](int t0, f32x4 (&sc)[2][4], bool first_tile) {
            const unsigned long long vm = sValid[stage * VW + (t0 >> 6)];
            const unsigned vm_lo = __builtin_amdgcn_readfirstlane((unsigned)vm), vm_hi = __builtin_amdgcn_readfirstlane((unsigned)(vm >> 32));
            const bool all_valid = (vm_lo & vm_hi) == 0xffffffffu;                         // wave-uniform
            const unsigned vb_lo = vm_lo >> (4 * g), vb_hi = vm_hi >> (4 * g);              // bit 16 (kt & 1) + r of word kt >> 1
            float tmax[2];
#pragma unroll
            for (int mq = 0; mq < 2; ++mq) {
                tmax[mq] = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    if (!all_valid) {
                        const unsigned w = (kt >> 1) ? vb_hi : vb_lo;
#pragma unroll
                        for (int r = 0; r < 4; ++r) sc[mq][kt][r] = ((w >> (16 * (kt & 1) + r)) & 1u) ? sc[mq][kt][r] : -INFINITY;
                    }
                    const float m4 = max_nc(max_nc(sc[mq][kt][0], sc[mq][kt][1]), max_nc(sc[mq][kt][2], sc[mq][kt][3]));
                    tmax[mq] = kt == 0 ? m4 : max_nc(tmax[mq], m4);
                }
            }
            tmax[0] = quad_max(tmax[0]);
            tmax[1] = quad_max(tmax[1]);
            // Deferred rescale (guide T13): the running maximum is only a scaling reference; P and the accumulators stay exact as long
            // as one reference is used per row.  Keep the old one while no row of this wave outgrows it by more than 2^RESCALE_LOG2
            // (p <= 2^16: nothing near an fp32 / bf16 range limit) -- the accumulator rescale and its exponential then run on the first
            // tile of an item and on the rare tile that trips the threshold instead of on every tile.
            constexpr float RESCALE_LOG2 = 16.0f;
            bool grow = first_tile;
            if (!first_tile) {
                const bool g0 = (tmax[0] - mx[0]) * kexp > RESCALE_LOG2, g1 = (tmax[1] - mx[1]) * kexp > RESCALE_LOG2;   // -inf - -inf = NaN: false
                grow = __any((g0 || g1) ? 1 : 0) != 0;                                                                  // wave-uniform
            }
            if (grow) {
#pragma unroll
                for (int mq = 0; mq < 2; ++mq) {
                    const float mnew = fmaxf(mx[mq], tmax[mq]);
                    if (!first_tile) {
                        // alpha = 2^((m_old - m_new) kexp); rows that had no valid key so far carry zeros: any finite alpha will do
                        const float alpha = (mx[mq] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((mx[mq] - mnew) * kexp);
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt) {
                            oacc[mq][dt][0] *= alpha; oacc[mq][dt][1] *= alpha; oacc[mq][dt][2] *= alpha; oacc[mq][dt][3] *= alpha;
                        }
                        lacc[mq][0] *= alpha; lacc[mq][1] *= alpha; lacc[mq][2] *= alpha; lacc[mq][3] *= alpha;
                    }
                    mx[mq] = mnew;
                }
            }
        };
        // P = 2^((s - m) kexp) as the 16-bit operand (straight from the accumulator registers), its row sums, and O += P V
        auto exp_half = [&](f32x4 (&sc)[2][4], bf16x8 (&pf)[2][2], int p) {
#pragma unroll
            for (int mq = 0; mq < 2; ++mq) {
                const float nml = (mx[mq] == -INFINITY) ? 0.f : -mx[mq] * kexp;   // a row without a valid key yet: p = 2^(-inf) = 0
#pragma unroll
                for (int kt = 2 * p; kt < 2 * p + 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        pf[mq][p][(kt & 1) * 4 + r] = (bf16_t)__builtin_amdgcn_exp2f(fmaf(sc[mq][kt][r], kexp, nml));
            }
        };
        auto pv_half = [&](int t0, bf16x8 (&pf)[2][2], int p) {
            lacc[0] = X16<E>::mfma(ones, pf[0][p], lacc[0]);      // row sums of the 16-bit P the P V product actually uses
            lacc[1] = X16<E>::mfma(ones, pf[1][p], lacc[1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16_t* vb = sV + (t0 + 32 * p) * 64 + v_off[dt];
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vb));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vb + 16 * 64));
                union { s16x4 s[2]; bf16x8 v; } u;
                u.s[0] = lo; u.s[1] = hi;
                oacc[0][dt] = X16<E>::mfma(u.v, pf[0][p], oacc[0][dt]);
                oacc[1][dt] = X16<E>::mfma(u.v, pf[1][p], oacc[1][dt]);
            }
        };
        if (PIPE && (kb + 1) * KB <= Tcur) {
            // SOFTWARE-PIPELINED form for a full stage (round 4).  In the rolled loop below a tile is Q K^T (16 MFMAs), then ~190 vector
            // instructions of softmax, then P V (20 MFMAs): matrix pipe and vector ALU take turns -- 576 + ~770 cycles per wave and
            // tile, and the two waves of a SIMD, which run the same program between the same barriers, do it in lockstep (2,600 cycles
            // per pair of tiles by in-kernel stamps, round 3: the SUM).  Here the stage's tiles are unrolled and the NEXT tile's
            // Q K^T is issued between this tile's maximum and its exponentials (scores double-buffered, +32 registers), and P V of the
            // first 32 keys runs under the exponentials of the other 32: the MFMAs of one tile sit in the shadow of the vector work of
            // its neighbours, inside ONE wave's instruction stream (hipcc interleaves what is independent within a basic block).
            constexpr int NT = KB / 64;
            f32x4 sc[2][2][4];
            qk_tile(0, sc[0]);
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                tile_max(64 * i, sc[i & 1], kb == 0 && i == 0);
                if (dma_in_loop && nitem < n_items) issue_part(nbase, nkbi, stage ^ 1, i * 4 / NT, (i + 1) * 4 / NT);
                bf16x8 pf[2][2];
                if (i + 1 < NT) qk_tile(64 * (i + 1), sc[(i + 1) & 1]);
                exp_half(sc[i & 1], pf, 0);
                exp_half(sc[i & 1], pf, 1);
                pv_half(64 * i, pf, 0);
                pv_half(64 * i, pf, 1);
            }
        } else {
#pragma unroll 1
            for (int t0 = 0; t0 < KB && kb * KB + t0 < Tcur; t0 += 64) {   // workgroup-uniform bounds (packed input: this sequence's length)
                f32x4 sc[2][4];                                   // both query blocks side by side: two independent dependency chains per wave
                bf16x8 pf[2][2];                                  // [query block][32-key half]
                qk_tile(t0, sc);
                tile_max(t0, sc, kb == 0 && t0 == 0);
                exp_half(sc, pf, 0);
                exp_half(sc, pf, 1);
                pv_half(t0, pf, 0);
                pv_half(t0, pf, 1);
            }
        }

        PH_STAMP(un * 8 + 4);
        if (kb == nkb - 1) {
            const int bh = item / nqb, qb = item - bh * nqb, bb = bh / H, hh = bh - bb * H;
            __syncthreads();                                      // every wave is done with this stage's K image
            bf16_t* so = reinterpret_cast<bf16_t*>(lds + stage * STAGE) + wave * (32 * 64);   // [32 queries][64 d], chunk ^= query & 7
#pragma unroll
            for (int mq = 0; mq < 2; ++mq) {
                // 1 ulp, far below the bf16 output rounding; a row without ANY valid key (all-zero mask) has lacc == 0 and oacc == 0:
                // it gets zeros, not rcp(0) * 0 = NaN (the reference's finfo.min mask would attend uniformly to padding there --
                // an input the tokenizer never produces: [CLS] is always valid)
                const float inv = lacc[mq][0] > 0.f ? __builtin_amdgcn_rcpf(lacc[mq][0]) : 0.f;
                const int ql = mq * 16 + c;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    bf16x4 o = {(bf16_t)(oacc[mq][dt][0] * inv), (bf16_t)(oacc[mq][dt][1] * inv), (bf16_t)(oacc[mq][dt][2] * inv),
                                (bf16_t)(oacc[mq][dt][3] * inv)};
                    const int ch = (dt * 2 + (g >> 1)) ^ (ql & 7);
                    *reinterpret_cast<bf16x4*>(so + ql * 64 + ch * 8 + 4 * (g & 1)) = o;
                }
            }
            // a wave reads back only what it wrote itself: lane -> row (lane >> 3) + 8 i, 16-byte chunk lane & 7
            {
                const int ql = lane >> 3, ch = lane & 7;             // (8 i + ql) & 7 == ql
                const bf16_t* sp = so + ql * 64 + ((ch ^ ql) * 8);
                ov0 = *reinterpret_cast<const float4*>(sp);
                ov1 = *reinterpret_cast<const float4*>(sp + 8 * 64);
                ov2 = *reinterpret_cast<const float4*>(sp + 16 * 64);
                ov3 = *reinterpret_cast<const float4*>(sp + 24 * 64);
            }
            orow = qb * QPB + wave * 32 + (lane >> 3);
            oT = Tcur;
            optr = out + ((int64_t)row0cur + orow) * C + hh * DH + (lane & 7) * 8;
        }
        item = nitem; kb = nkbi; stage ^= 1;
        PH_STAMP(un * 8 + 5);
        ++un;
    }
    MGEA_FLUSH_OUT();
#undef MGEA_FLUSH_OUT
}

template <typename E, int NW, int PIPE>
static int launch_attn16(const void* qkv, const int32_t* mask, void* out, int B, int T, int H, hipStream_t st, const KvPages& pg, const int32_t* cu) {
    constexpr int QPB = 32 * NW, KB = 32 * NW;
    const int nqb = ceil_div(T, QPB);
    const int64_t n_items = (int64_t)B * H * nqb;
    MGEA_REQUIRE(n_items < (1 << 30), MGEA_EINVAL, "16-bit attention: too many (batch, head, query block) items");
    DeviceInfo di;
    MGEA_TRY(device_info(&di));
    const int shmem = 2 * KB * 16 * 16 + 64;
    static uint64_t attr_done = 0;                     // per instantiation
    MGEA_TRY(set_max_dynamic_lds((const void*)attn_bf16_kernel<E, NW, PIPE>, shmem, di.dev, &attr_done));
    const int per_cu = NW == 4 ? 2 : 1;
    const int grid = (int)(n_items < per_cu * di.n_cu ? n_items : per_cu * di.n_cu);
    hipLaunchKernelGGL((attn_bf16_kernel<E, NW, PIPE>), dim3(grid), dim3(64 * NW), shmem, st, (const E*)qkv, mask, (E*)out, T, H, (int)n_items, nqb,
                       0.125f, pg, cu);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

int launch_attn_bf16(const void* qkv, const int32_t* mask, void* out, int B, int T, int H, int dh, hipStream_t st, int f16, const KvPages* pages,
                     const int32_t* cu) {
    MGEA_REQUIRE(dh == 64, MGEA_EINVAL, "bf16 attention: head_dim %d not supported (64)", dh);
    MGEA_REQUIRE(B > 0 && T > 0 && B <= 65535 && H <= 65535, MGEA_EINVAL, "bf16 attention: bad shape");
    MGEA_REQUIRE((int64_t)T * 6 * H * dh < ((int64_t)1 << 32), MGEA_EINVAL, "bf16 attention: one sequence of qkv rows must span < 4 GB");
    // long sequences: 256 queries / 256 keys per unit (switch attn16_wide: 0 never, 1 from 512 tokens (default), 2 always)
    const int wide = tune(TUNE_ATTN16_WIDE);
    bool w8 = wide == 2 || (wide == 1 && T >= 512);
    if (cu) {   // packed input: T = the longest sequence, which must fit one query block / key stage of the form that runs it
        MGEA_REQUIRE(!mask && T <= 256, MGEA_EINVAL, "16-bit attention over packed rows: no key mask, sequences of at most 256 tokens (got %d)", T);
        w8 = T > 128;
    }
    // the wide form runs its full stages software-pipelined (round 4; switch attn16_pipe = 0: the rolled tile loop).  The 4-wave form keeps
    // the rolled loop: at 128 keys it is memory-bound and the pipelined body spilled 31 registers (44 -> 54 us on [256, 128, 12 x 64])
    const bool pipe = tune(TUNE_ATTN16_PIPE) != 0;
    // (a static priority for waves 4..7 -- guide T5, static form -- on top of the pipelined body: 224.9 vs 227.0 us, nothing; not kept)
    const KvPages none{};
    const KvPages& pg = pages ? *pages : none;
    if (f16) {
        if (w8) return pipe ? launch_attn16<_Float16, 8, 1>(qkv, mask, out, B, T, H, st, pg, cu) : launch_attn16<_Float16, 8, 0>(qkv, mask, out, B, T, H, st, pg, cu);
        return launch_attn16<_Float16, 4, 0>(qkv, mask, out, B, T, H, st, pg, cu);
    }
    MGEA_REQUIRE(!pg.pool.base, MGEA_EINVAL, "16-bit attention: KV pages are written by the fp16 instantiation only");
    if (w8) return pipe ? launch_attn16<bf16_t, 8, 1>(qkv, mask, out, B, T, H, st, pg, cu) : launch_attn16<bf16_t, 8, 0>(qkv, mask, out, B, T, H, st, pg, cu);
    return launch_attn16<bf16_t, 4, 0>(qkv, mask, out, B, T, H, st, pg, cu);
}

// ------------------------------------------------------------------------------------------
// fp16 big-batch prefill of the decoder (MGEA_DTYPE_F16, decoder.hip: run_prefill16): the pre-LN GPT block on the kernels above with
// _Float16 operands.  Three small kernels around them:
// x[m] = f16(tok_emb[ids[m]] + pos_emb[pos]) and the (mean, rstd) of the ROUNDED row (what the folded-LayerNorm GEMM consumes);
// rows with t >= lens[b] are zero rows with the identity statistics (0, 1).  One wave per row.
__global__ __launch_bounds__(256) void dec_embed_f16_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ lens,
                                                           const int32_t* __restrict__ ctx_len, const float* __restrict__ tok_emb,
                                                           const float* __restrict__ pos_emb, _Float16* __restrict__ x,
                                                           float* __restrict__ rowstat, int32_t* __restrict__ mask_out, float eps, int M,
                                                           int T, int C, int vocab, int pos_rows, int absolute_pos,
                                                           int32_t* __restrict__ err_flag) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const int b = (int)(m / T), t = (int)(m % T);
    const bool real = lens ? (t < lens[b]) : true;
    int id = ids[m];
    if (real && (id < 0 || id >= vocab) && err_flag && lane == 0) atomicOr(err_flag, 1);
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    int pos = t + ((absolute_pos && ctx_len) ? ctx_len[b] : 0);
    pos = pos < pos_rows ? pos : pos_rows - 1;
    const int nf4 = C >> 2;
    float4 v[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int f = lane + i * 64;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < nf4 && real) {
            const float4 e = add4(ld4(tok_emb + (int64_t)id * C + f * 4), ld4(pos_emb + (int64_t)pos * C + f * 4));
            const h16x4 o = {(_Float16)e.x, (_Float16)e.y, (_Float16)e.z, (_Float16)e.w};
            v[i] = make_float4((float)o[0], (float)o[1], (float)o[2], (float)o[3]);
        }
        if (f < nf4) {
            const h16x4 o = {(_Float16)v[i].x, (_Float16)v[i].y, (_Float16)v[i].z, (_Float16)v[i].w};
            *reinterpret_cast<h16x4*>(x + m * C + f * 4) = o;
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (lane + i * 64 < nf4) {
            const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    const float var = wave_sum(q) / (float)C;
    if (lane == 0) {
        *reinterpret_cast<float2*>(rowstat + m * 2) = real ? make_float2(mean, 1.0f / sqrtf(var + eps)) : make_float2(0.f, 1.f);
        if (mask_out) mask_out[m] = real ? 1 : 0;           // the key validity the dense attention wants for ragged prompts
    }
}
int launch_dec_embed_f16(const int32_t* ids, const int32_t* lens, const int32_t* ctx_len, const float* tok_emb, const float* pos_emb,
                         void* x, float* rowstat, int32_t* mask_out, float eps, int B, int T, int C, int vocab, int pos_rows, int absolute_pos,
                         int32_t* err_flag, hipStream_t st) {
    MGEA_REQUIRE(C % 4 == 0 && C <= 2048, MGEA_EINVAL, "fp16 embed: d_model=%d must be a multiple of 4 and <= 2048", C);
    hipLaunchKernelGGL(dec_embed_f16_kernel, dim3(ceil_div(B * T, 4)), dim3(256), 0, st, ids, lens, ctx_len, tok_emb, pos_emb,
                       (_Float16*)x, rowstat, mask_out, eps, B * T, T, C, vocab, pos_rows, absolute_pos, err_flag);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

// K | V columns of the fp16 qkv rows [M, 3C] -> the fp16 KV pages of `layer` (common.h: K [dh/8][64 tokens][8], V [64 tokens][dh]):
// one thread per 16-byte group, real tokens only, position ctx_len[b] + t.
__global__ __launch_bounds__(256) void kv_scatter_f16_kernel(const _Float16* __restrict__ qkv, KvPool pool, int layer,
                                                            const int32_t* __restrict__ page_table, int max_pages,
                                                            const int32_t* __restrict__ ctx_len, const int32_t* __restrict__ lens,
                                                            int64_t n_groups, int T, int C) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= n_groups) return;
    const int gpr = (2 * C) >> 3;                       // 16-byte groups of K | V per row
    const int64_t m = gid / gpr;
    const int gi = (int)(gid - m * gpr);
    const int b = (int)(m / T), t = (int)(m % T);
    const bool real = lens ? (t < lens[b]) : true;
    if (!real) return;
    const int pos = ctx_len[b] + t;
    const int page = pos >> 6, slot = pos & 63;
    if (page >= max_pages) return;
    const int phys = page_table[b * max_pages + page];
    const int n = gi * 8;                                // column inside K | V
    const int isv = n >= C, nn = n - (isv ? C : 0), head = nn / pool.dh, d = nn % pool.dh;
    const float4 raw = *reinterpret_cast<const float4*>(qkv + m * 3 * C + C + n);
    const int64_t pe = pool.page_elems();
    _Float16* pg = static_cast<_Float16*>(pool.base) + layer * pool.layer_stride + ((int64_t)(phys * 2 + isv) * pool.H + head) * pe;
    *reinterpret_cast<float4*>(pg + (isv ? slot * pool.dh + d : ((d >> 3) * MGEA_KV_PAGE_TOKENS + slot) * 8)) = raw;
}
int launch_kv_scatter_f16(const void* qkv, const KvPool& pool, int layer, const int32_t* page_table, int max_pages, const int32_t* ctx_len,
                          const int32_t* lens, int B, int T, int C, hipStream_t st) {
    MGEA_REQUIRE(pool.f16 && pool.dh % 8 == 0 && C % 8 == 0, MGEA_EINVAL, "fp16 KV scatter needs fp16 pages and head_dim %% 8 == 0");
    const int64_t n_groups = (int64_t)B * T * ((2 * C) >> 3);
    hipLaunchKernelGGL(kv_scatter_f16_kernel, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, st, (const _Float16*)qkv, pool, layer,
                       page_table, max_pages, ctx_len, lens, n_groups, T, C);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

}  // namespace mgea
