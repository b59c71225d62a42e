// LM head of a decode step (api_cache.py:105: logits = head(x), M <= 64 rows, N = vocab, K = d_model) as ONE ROUND of the chip:
// exactly one workgroup per CU, every workgroup the same amount of matrix work.
//
// Why a kernel of its own (round 4).  The generic skinny kernel runs the head of Decoder-S (64 x 8324 x 512) as 522 tiles of 32 x 32 on
// 256 CUs: two rounds and a bit, three workgroups on some CUs and two on the others (the dispatcher decides), every workgroup
// loading everything before its first MFMA -- 13 us per step, the largest single GEMM launch of the step.  In-kernel stamps of round 1
// (profiles/r1_v5_skinny_phases.txt) had already shown workgroups of a 292-workgroup hybrid grid starting 9.7 us late: the grid was not
// co-resident.  Here the grid IS the chip:
//   * gridDim.x = G = the CU count (256), and each workgroup asks for > 80 KB of LDS (it needs 92 KB for its partial tiles anyway), so no
//     two can share a CU: 256 workgroups occupy 256 CUs by construction, whatever the dispatcher's order (speed only -- a second
//     workgroup on a CU would simply run after the first);
//   * work is dealt in UNITS of one 16 x 16 output tile over the whole K.  With R = 4 row tiles and J = ceil(N / 16) column tiles, workgroup g
//     owns the BASE = J / G column tiles g * BASE .. (all four row tiles: 4 BASE units) and, if g < 4 (J % G), ONE more unit of the
//     left-over column tiles (column tile G * BASE + g / 4, row tile g % 4): 8 or 9 units at V = 8324 where a 64 x 32 tiling has
//     workgroups of 8 and CUs with 16.  (Every workgroup issues the 9th unit's loads and MFMAs -- branch-free, a dummy for those without
//     one: the launch lasts as long as its longest workgroup either way.)
//   * the 8 waves split K exactly as in gemm_skinny_kernel (wave w owns k in [w K / 8, (w + 1) K / 8), NCH chunks of 32), every output
//     element is the same v_mfma_f32_16x16x4_f32 chain per wave and the same wave-order sum through LDS: logits BIT-IDENTICAL to the
//     generic kernel's (asserted in tests/test_gpu_ops.py), so every golden id test is unaffected by which of the two runs.
// Operands arrive in fragment order (common.h: tiled_off, launch_tile_weights): each load instruction of a wave is 1 KB of consecutive
// bytes.  Per workgroup: W 3 x 32 KB + x 128 KB.  fp32 MFMA work: 9 units x 128 k-steps / 8 waves = 144 MFMAs per wave, 2 waves per SIMD.
// F16 engines: W as _Float16 fragments, x rounded on load, one v_mfma_f32_16x16x32_f16 per (unit, chunk) (as gemm_skinny_kernel<.., F16>).
//
// Greedy decoding needs no logits in HBM: every workgroup leaves ONE (max, argmax) partial per row over the columns it owns,
// pmax[row][g], lowest index on ties (torch.argmax / topk(1) semantics, api_cache.py:171-178 with top_k = 1); the step's tail kernel
// merges the G partials of a row.
#include "common.h"

namespace mgea {

namespace {
constexpr int HB_PITCH = 20;   // floats per LDS row of a partial tile (16 + 4: the 16 rows of one ds_write_b128 start in different banks)

// TS (tools/head_phases.py only, dbg bit 21; never in an engine): lane 0 of every wave writes 100 MHz stamps of its phases to stats_out
template <int BASE, int NCH, bool F16, bool TS = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void head_balanced_kernel(SkinnyArgs a, int n_extra) {
    constexpr int NU = 4 * BASE + 1;                 // units per workgroup (the last one is the extra / dummy unit)
    constexpr int NS = 4 * (BASE + 1);               // (max, argmax) slots per row: 4 column quads per column tile
    extern __shared__ __attribute__((aligned(16))) float red[];   // [8 waves][NU][16][HB_PITCH], then pb_val[64][NS], pb_idx[64][NS]
    float* pb_val = red + 8 * NU * 16 * HB_PITCH;
    int* pb_idx = reinterpret_cast<int*>(pb_val + 64 * NS);
    asm volatile("" :: "s"(a.A), "s"(a.W), "s"(a.bias), "s"(a.M), "s"(a.N), "s"(a.K), "s"(a.out), "s"(a.ldo), "s"(a.pmax_val), "s"(a.pmax_idx));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int wg = blockIdx.x, G = gridDim.x;
    const bool has_x = wg < n_extra;                                   // this workgroup owns a real extra unit
    const int jx = has_x ? G * BASE + (wg >> 2) : wg * BASE;           // its column tile (dummy: the workgroup's own first tile, L1-hot)
    const int rx = has_x ? (wg & 3) : 0;                               // its row tile
    const unsigned mk0 = rx == 0 ? ~0u : 0u, mk1 = rx == 1 ? ~0u : 0u, mk2 = rx == 2 ? ~0u : 0u, mk3 = rx == 3 ? ~0u : 0u;
    const int kbeg = wave * 32 * NCH, chunks = a.K >> 5;
    long long* ts = TS ? reinterpret_cast<long long*>(a.stats_out + 2 * 64 * (int64_t)gridDim.x) + ((int64_t)blockIdx.x * 8 + wave) * 8 : nullptr;   // behind the (max, argmax) partials
#define HB_TS(i) do { if (TS && lane == 0) ts[i] = wall_clock64(); } while (0)
    HB_TS(0);

    // ---- every operand load of the wave, chunk-major (chunk 0 of everything first: the MFMAs of chunk 0 run under chunk 1's flight)
    float4 wf[BASE + 1][NCH][2], af[4][NCH][2];
    h16x8 wh[BASE + 1][NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
        for (int t = 0; t <= BASE; ++t) {
            const int j = t < BASE ? wg * BASE + t : jx;
            const int64_t blk = ((int64_t)j * chunks + (kbeg >> 5) + ch) * 512;
            if (F16) {
                wh[t][ch] = *reinterpret_cast<const h16x8*>(reinterpret_cast<const _Float16*>(a.W) + blk + lane * 8);
            } else {
                wf[t][ch][0] = ld4(a.W + blk + lane * 4);
                wf[t][ch][1] = ld4(a.W + blk + 256 + lane * 4);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {      // rows >= M of the 64-row group hold stale data whose products are never stored
            const float* ap = a.A + tiled_off(16 * r, kbeg + 32 * ch, a.K) + lane * 4;
            af[r][ch][0] = ld4(ap);
            af[r][ch][1] = ld4(ap + 256);
        }
    }
    HB_TS(1);
    if (TS) { asm volatile("s_waitcnt vmcnt(%0)" :: "n"((F16 ? BASE + 1 : 2 * (BASE + 1)) + 8) : "memory"); HB_TS(2); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); HB_TS(3); }
    __builtin_amdgcn_sched_barrier(0);   // every load above is in flight before the first MFMA (hipcc otherwise feeds them in among the
                                         // MFMAs two at a time to save registers: a chain of dependent round trips)
    f32x4 acc[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto mac = [&](f32x4& d, const float4 (&w)[2], const h16x8& w16, const float4 (&x)[2]) {
        if (F16) {
            d = __builtin_amdgcn_mfma_f32_16x16x32_f16(w16, to_h8(x[0], x[1]), d, 0, 0, 0);
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {     // the k order of gemm_skinny_kernel: h, then the four k of a float4
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(w[h].x, x[h].x, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(w[h].y, x[h].y, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(w[h].z, x[h].z, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(w[h].w, x[h].w, d, 0, 0, 0);
            }
        }
    };
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
        for (int t = 0; t < BASE; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) mac(acc[4 * t + r], wf[t][ch], wh[t][ch], af[r][ch]);
        // the extra unit's x fragment: row tile rx, chosen with wave-uniform selects (no runtime-indexed register array)
        // -- as bit masks, not as ?: on the array elements: LLVM turns a select of two array elements into a load from a selected ADDRESS,
        // which keeps the whole fragment array in scratch memory (28 scratch stores + indexed reloads in the first build of this kernel)
        float4 xs[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            auto pick = [&](float v0, float v1, float v2, float v3) {
                const unsigned u = (__builtin_bit_cast(unsigned, v0) & mk0) | (__builtin_bit_cast(unsigned, v1) & mk1) |
                                   (__builtin_bit_cast(unsigned, v2) & mk2) | (__builtin_bit_cast(unsigned, v3) & mk3);
                return __builtin_bit_cast(float, u);
            };
            const float4 &p0 = af[0][ch][h], &p1 = af[1][ch][h], &p2 = af[2][ch][h], &p3 = af[3][ch][h];
            xs[h] = make_float4(pick(p0.x, p1.x, p2.x, p3.x), pick(p0.y, p1.y, p2.y, p3.y), pick(p0.z, p1.z, p2.z, p3.z), pick(p0.w, p1.w, p2.w, p3.w));
        }
        mac(acc[4 * BASE], wf[BASE][ch], wh[BASE][ch], xs);
    }
    // the epilogue's bias values are requested now, so that their round trip runs under the MFMA drain, the LDS reduction and the barrier
    // (requested at kernel start they would queue behind the operands and drag an early wait in: gemm_skinny_kernel, round 1)
    auto item_n = [&](int it) { const int u = it >> 6; return (u == 4 * BASE ? jx : wg * BASE + (u >> 2)) * 16 + 4 * (it & 3); };
    auto load_bias = [&](int n) {
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias) {
            if (n + 3 < a.N) b = ld4(a.bias + n);
            else { if (n < a.N) b.x = a.bias[n]; if (n + 1 < a.N) b.y = a.bias[n + 1]; if (n + 2 < a.N) b.z = a.bias[n + 2]; }
        }
        return b;
    };
    const float4 bias0 = load_bias(item_n(tid));
    if (TS) { __builtin_amdgcn_sched_barrier(0); HB_TS(4); }
    // ---- partial tiles -> LDS: D[i = column 4 g + reg][j = row c]
#pragma unroll
    for (int u = 0; u < NU; ++u)
        *reinterpret_cast<float4*>(&red[((wave * NU + u) * 16 + c) * HB_PITCH + 4 * g]) = make_float4(acc[u][0], acc[u][1], acc[u][2], acc[u][3]);
    __syncthreads();
    HB_TS(5);

    // ---- epilogue: item = (unit u, local row lr, column quad q); 8 partials summed in wave order (deterministic)
    for (int it = tid; it < NU * 64; it += 512) {
        const int u = it >> 6, lr = (it >> 2) & 15, q = it & 3;
        const bool extra = u == 4 * BASE;
        const int row = (extra ? rx : (u & 3)) * 16 + lr;
        const int j = extra ? jx : wg * BASE + (u >> 2);
        const int n = j * 16 + 4 * q;
        float4 pv[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) pv[w] = *reinterpret_cast<const float4*>(&red[((w * NU + u) * 16 + lr) * HB_PITCH + 4 * q]);
        float4 v = pv[0];
#pragma unroll
        for (int w = 1; w < 8; ++w) v = add4(v, pv[w]);
        const float4 bb = it == tid ? bias0 : load_bias(n);
        float e[4] = {v.x + bb.x, v.y + bb.y, v.z + bb.z, v.w + bb.w};
        float best = -INFINITY;
        int bi = 0x7fffffff;
        const bool live = row < a.M && (!extra || has_x);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            if (live && n + jj < a.N && e[jj] > best) { best = e[jj]; bi = n + jj; }
        }
        if (live && a.out) {
            float* op = a.out + (int64_t)row * a.ldo + n;
            if (n + 3 < a.N && ((a.ldo | n) & 3) == 0) {
                st4(op, make_float4(e[0], e[1], e[2], e[3]));
            } else {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    if (n + jj < a.N) op[jj] = e[jj];
            }
        }
        // slot of this item in its row's partial list: own column tiles first, the extra unit's four quads last
        const int slot = extra ? 4 * BASE + q : 4 * (u >> 2) + q;
        if (!extra || has_x) {
            pb_val[row * NS + slot] = best;
            pb_idx[row * NS + slot] = bi;
        }
    }
    // rows outside the extra unit's row tile (all rows when there is none): empty slots
    if (tid < 64 * 4) {
        const int row = tid >> 2, q = tid & 3;
        if (!has_x || (row >> 4) != rx) {
            pb_val[row * NS + 4 * BASE + q] = -INFINITY;
            pb_idx[row * NS + 4 * BASE + q] = 0x7fffffff;
        }
    }
    __syncthreads();
    if (tid < 64 && tid < a.M && a.pmax_val) {
        float4 sv[NS / 4];
        int4 si[NS / 4];
#pragma unroll
        for (int s = 0; s < NS / 4; ++s) {   // all slots requested before the first compare (one LDS round trip, not NS)
            sv[s] = *reinterpret_cast<const float4*>(&pb_val[tid * NS + 4 * s]);
            si[s] = *reinterpret_cast<const int4*>(&pb_idx[tid * NS + 4 * s]);
        }
        float best = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int s = 0; s < NS / 4; ++s) {
            const float ov[4] = {sv[s].x, sv[s].y, sv[s].z, sv[s].w};
            const int oi[4] = {si[s].x, si[s].y, si[s].z, si[s].w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (ov[k] > best || (ov[k] == best && oi[k] < bi)) { best = ov[k]; bi = oi[k]; }
        }
        a.pmax_val[(int64_t)tid * G + wg] = best;
        a.pmax_idx[(int64_t)tid * G + wg] = bi;
    }
    if (TS) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); HB_TS(6); }
#undef HB_TS
}

struct HeadPlan { int G, base, n_extra, nch; };

// the shapes this kernel takes: 3..64 rows (1-2 rows run as dot products, gemv_small.hip), K = 8 waves x NCH chunks of 32,
// at least one column tile per CU and at most one extra unit per workgroup
bool head_plan(int M, int N, int K, HeadPlan* p) {
    if (!tune(TUNE_HEAD_BALANCED) || M < 3 || M > 64) return false;
    if (K != 256 && K != 512 && K != 768) return false;
    DeviceInfo di;
    if (device_info(&di) != MGEA_OK) return false;
    const int G = di.n_cu / 8 * 8;
    if (G < 8) return false;
    const int J = ceil_div(N, 16), base = J / G, rem = J - base * G;
    if (base < 1 || base > 3 || 4 * rem > G) return false;
    if (base == 3 && K == 768) return false;          // (13 accumulator tiles + 3-chunk fragments of 4 + 4 tiles: over the register budget)
    *p = HeadPlan{G, base, 4 * rem, K / 256};
    return true;
}
}  // namespace

// partial (max, argmax) entries per row the LOGITS launch of this shape writes, or 0 when the balanced kernel does not take it
int head_balanced_partials(int M, int N, int K) {
    HeadPlan p;
    return head_plan(M, N, K, &p) ? p.G : 0;
}

template <int BASE, int NCH, bool F16>
static int launch_head_t(const SkinnyArgs& a, const HeadPlan& p, hipStream_t st) {
    constexpr int NU = 4 * BASE + 1, NS = 4 * (BASE + 1);
    int shmem = (8 * NU * 16 * HB_PITCH + 64 * NS * 2) * (int)sizeof(float);
    if (shmem < 84 * 1024) shmem = 84 * 1024;           // more than half of a CU's 160 KB: one workgroup per CU
    DeviceInfo di;
    MGEA_TRY(device_info(&di));
    static uint64_t attr_done = 0;
    MGEA_TRY(set_max_dynamic_lds(reinterpret_cast<const void*>(&head_balanced_kernel<BASE, NCH, F16>), shmem, di.dev, &attr_done));
    hipLaunchKernelGGL((head_balanced_kernel<BASE, NCH, F16>), dim3(p.G), dim3(512), shmem, st, a, p.n_extra);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

// returns MGEA_OK after launching, or 1 when the shape is not this kernel's (the caller falls back to gemm_skinny_kernel)
int launch_head_balanced(const SkinnyArgs& a, hipStream_t st) {
    HeadPlan p;
    if (a.ln_c1 || (a.dbg & ~(1 << 21)) || !head_plan(a.M, a.N, a.K, &p)) return 1;
    if (a.dbg & (1 << 21)) {    // tools/head_phases.py: the stamped build of the benchmark's instantiation
        if (p.base != 2 || p.nch != 2 || a.w_f16 || !a.stats_out) return 1;
        constexpr int NU = 9, NS = 12;
        int shmem = (8 * NU * 16 * HB_PITCH + 64 * NS * 2) * (int)sizeof(float);
        if (shmem < 84 * 1024) shmem = 84 * 1024;
        DeviceInfo di;
        MGEA_TRY(device_info(&di));
        static uint64_t ts_done = 0;
        MGEA_TRY(set_max_dynamic_lds(reinterpret_cast<const void*>(&head_balanced_kernel<2, 2, false, true>), shmem, di.dev, &ts_done));
        hipLaunchKernelGGL((head_balanced_kernel<2, 2, false, true>), dim3(p.G), dim3(512), shmem, st, a, p.n_extra);
        MGEA_CHECK_HIP(hipGetLastError());
        return MGEA_OK;
    }
#define MGEA_HEAD_GO(B_, N_)                                                              \
    if (p.base == B_ && p.nch == N_)                                                      \
        return a.w_f16 ? launch_head_t<B_, N_, true>(a, p, st) : launch_head_t<B_, N_, false>(a, p, st);
    MGEA_HEAD_GO(1, 1) MGEA_HEAD_GO(1, 2) MGEA_HEAD_GO(1, 3)
    MGEA_HEAD_GO(2, 1) MGEA_HEAD_GO(2, 2) MGEA_HEAD_GO(2, 3)
    MGEA_HEAD_GO(3, 1) MGEA_HEAD_GO(3, 2)
#undef MGEA_HEAD_GO
    return 1;
}

}  // namespace mgea
