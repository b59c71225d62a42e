// Fused skinny GEMM for the decode step (M <= 512 rows, designed around M = 64): out[M, N] = f(LN?(A)[M, K] @ W[N, K]^T + bias)
// with the whole row epilogue inside the kernel, so a decoder layer is 5 launches
// (QKV, attention, out-proj, FC1, FC2) instead of 9 and no split-K slab ever goes to HBM.
//
// Reference ops covered (api_cache.py): ln1 + in_proj + cache append (:60-67), out_proj + residual
// (:68,72), ln2 + mlp.0 + GELU (:73), mlp.2 + residual (:73), head (:105) + greedy argmax.
//
// Decomposition (gfx950): one workgroup = (16*MT rows) x (16 output columns) over the FULL K; its
// NW waves split K (wave w owns k in [w*K/NW, (w+1)*K/NW)) and their partial tiles are summed
// through LDS in wave order (deterministic, no atomics, no slabs).  MT in {1,2,4} is chosen per
// GEMM so that every launch has >= ~128-512 workgroups (the N of these GEMMs is only 512..2048):
// the MFMA work of a 64-row problem has to be spread over all 1024 SIMDs, and the bytes a CU pulls
// through its L2 port (~45-70 GB/s per CU) stay at 64-256 KB.  Workgroups that share a W tile differ
// by a multiple of gridDim.x (a multiple of 8) in dispatch order -> same XCD, so W is fetched from
// HBM once and re-read from that XCD's L2 (speed only; results never depend on placement).
//   * operands go HBM/L2 -> registers directly, and BOTH are stored in MFMA-fragment order (common.h:
//     tiled_off for activations, launch_tile_weights for the matrices), so every operand load instruction
//     of a wave is 1 KB of consecutive bytes: the row-major version, 16 B per lane from 16 different lines,
//     ran at a quarter of the L1 rate and was what bounded these kernels (tools/skinny_phases.py);
//   * v_mfma_f32_16x16x4_f32 (exact fp32), issued "swapped" so a lane owns 4 consecutive output
//     columns; the k order inside the chain is a fixed permutation (k = k0 + 8g + 4h + s) applied
//     to both operands; results are run-to-run bit-identical;
//   * LayerNorm is folded out of the K loop: with W' = gamma * W (built with the tiled copy), c1 = sum_k W'[n,k]
//     and c2 = sum_k beta[k] W[n,k] + bias[n],
//         LN(x) @ W^T + bias = rstd * (x @ W'^T - mean * c1) + c2,
//     so the loop runs on the raw x like the other kernels and nothing waits for the statistics before
//     the MFMAs.  The PRODUCER of x (embedding or a residual epilogue) leaves per-row partial statistics
//     (mean, M2) per 16-column tile; the consumer merges them with Chan's formula under its MFMA tail and
//     applies (mean, rstd) in the epilogue.
#include "common.h"

namespace mgea {

// Sum over each aligned group of 16 lanes with DPP butterflies (xor 1, xor 2, then mirrors of 8 and 16 lanes, which
// equal xor 4 / xor 8 once the smaller groups are uniform): every lane gets the bit-identical total, and the
// four steps cost a few cycles each instead of four LDS-crossbar round trips.
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}

// NT = 16-column tiles per workgroup (2 only for the LM head: 32 columns per workgroup halve the A bytes per output)
// F16 (MGEA_DTYPE_F16 engines): W is stored as _Float16 fragments of v_mfma_f32_16x16x32_f16 (launch_tile_weights_f16: half the
// bytes, one 16-byte load per lane and chunk), the fp32 activations are rounded to fp16 in registers after the load -- lane
// (c, g) of the k-tiled layout already holds the 8 consecutive k (k0 + 8 g .. + 7) that instruction wants -- and ONE MFMA
// replaces the eight 16x16x4 fp32 ones of a chunk; accumulation, LayerNorm statistics and every epilogue stay fp32.  With LN
// the LayerNorm weight gamma multiplies the activations before the rounding (W stays the plain rounded matrix, so the model
// is exactly "the reference with fp16-rounded matrices"): LN(x) W^T + b = rstd (sum_k (gamma_k x_k) W_nk - mean c1_n) + c2_n.
// NCH > 0 (the decode shapes of an engine: K = 32 * 8 * NCH): the workgroup has exactly 8 waves of NCH k-chunks each, all of them
// requested up front (no refill loop, no per-chunk branches, no runtime division, no debug timestamps): tools/micro/kernel_floor.hip
// shows that a kernel with this memory shape, LDS reduction, read-modify-write epilogue and MFMA chain costs ~3.0 us in a graph
// chain where the generic instruction stream (NCH = 0: any K, any wave count) took 3.9.  Same chunks per wave, same MFMA order,
// same wave-order reduction: bit-identical results.
template <int EPI, bool LN, int MT, int NT, bool F16, int NCH>
__global__ __launch_bounds__(512) void gemm_skinny_kernel(SkinnyArgs a) {
    constexpr int ROWS = 16 * MT, COLS = 16 * NT;
    // LDS row pitch of the partial tiles: + 4 floats, so that the 16 rows one ds_write_b128 of a wave touches do not
    // all start in the same bank (unpadded: 8-way conflicts at 16 columns, 16-way at 32)
    constexpr int PITCH = COLS + 4;
    // k-chunks (32 wide) a wave keeps in flight.  Measured (tools/skinny_phases.py): deeper (4, 8) does not
    // help -- a CU's vector memory path sustains only ~40-60 GB/s of L2 hits however many loads are queued
    constexpr int DEPTH = NCH ? NCH : ((MT * NT == 1) ? 4 : 2);
    extern __shared__ __attribute__((aligned(16))) float red[];  // [NW][ROWS][PITCH] (+ LN: mean and rstd of the ROWS rows)
    const int NW = NCH ? 8 : a.nw;   // = blockDim.x / 64, passed as an argument: blockDim comes from the dispatch packet, one more cold scalar load
    float* s_mean = red + NW * ROWS * PITCH;          // all LDS in ONE array (16-B aligned carve)
    float* s_rstd = s_mean + ROWS;

    // grid.x is padded to a multiple of 8 so that the workgroups sharing a W tile (same blockIdx.x, different
    // blockIdx.y) are a multiple of 8 apart in dispatch order = same XCD / same L2 (speed only)
    // The kernel-argument block (4 x 64-B lines) is read with scalar loads the compiler otherwise sinks to
    // their first use, block by block: each is a cold scalar-cache miss of several hundred ns on the critical
    // path of a ~6 us kernel.  Asking for every field here makes them ONE batch of loads and one wait.
    asm volatile("" :: "s"(a.A), "s"(a.W), "s"(a.bias), "s"(a.M), "s"(a.N), "s"(a.K), "s"(a.ln_c1), "s"(a.eps),
                       "s"(a.stats_in), "s"(a.n_part), "s"(a.part_cnt), "s"(a.out), "s"(a.ldo), "s"(a.stats_out), "s"(a.act), "s"(a.dbg), "s"(a.nw));
    if (F16 && LN) asm volatile("" :: "s"(a.ln_g));
    if (EPI == EPI_QKV)
        asm volatile("" :: "s"(a.pool.base), "s"(a.pool.H), "s"(a.pool.dh), "s"(a.pool.layer_stride), "s"(a.pool.f16), "s"(a.layer), "s"(a.page_table),
                           "s"(a.max_pages), "s"(a.ctx_len), "s"(a.lens), "s"(a.T), "s"(a.C));
    if (EPI == EPI_LOGITS) asm volatile("" :: "s"(a.pmax_val), "s"(a.pmax_idx));
    const int n_tiles = (a.N + COLS - 1) / COLS;
    if ((int)blockIdx.x >= n_tiles) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // tools/skinny_phases.py phase timing (dbg bit 20): 100 MHz timestamps of workgroup phases, written behind
    // the statistics in stats_out; the extra waits it inserts are only there in that mode
    long long* ts = (!NCH && (a.dbg & (1 << 20))) ? reinterpret_cast<long long*>(a.stats_out + 64 * (a.N >> 4) * 2) +
                                                        (blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr;
#define MGEA_TS(i) do { if (ts && tid == 0) ts[i] = wall_clock64(); } while (0)
    MGEA_TS(0);
    const int c = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * COLS;
    const int m0 = blockIdx.y * ROWS;
    const int kw = NCH ? 32 * NCH : a.K / NW, kbeg = wave * kw;
    const int nchunk = NCH ? NCH : kw >> 5;

    // LN: the producer's per-tile (mean, M2) partials of this tile's rows, one float4 (two partials) per lane,
    // 16 lanes per row; requested first, merged after the K loop
    float4 st0, st1;
    int s_lr = 0, s_j = 0;
    bool s_ok = false;
    if (LN) {
        const int slot = tid < ROWS * 16 ? tid : 0;
        s_lr = slot >> 4; s_j = slot & 15;
        s_ok = m0 + s_lr < a.M;
        const float* sp = a.stats_in + ((int64_t)(s_ok ? m0 + s_lr : 0) * a.n_part) * 2;
        const int p0 = 2 * s_j, p1 = 32 + 2 * s_j;   // n_part <= 64 and even (host check)
        st0 = ld4(sp + 2 * (p0 < a.n_part ? p0 : 0));
        st1 = ld4(sp + 2 * (p1 < a.n_part ? p1 : 0));
        __builtin_amdgcn_sched_barrier(0);   // these two stay ahead of the operand loads (loads return in issue order)
    }
    // operands in fragment order (common.h): lane l of a wave reads bytes [16 l, 16 l + 16) of a 1 KB block.
    // A: rows >= M of the last 64-row group hold stale data whose products are never stored
    const float* atile[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) atile[mt] = a.A + tiled_off(m0 + mt * 16, kbeg, a.K) + lane * 4;
    // per (16-row tile, 32-wide chunk): 512 floats (two 1 KB halves h = 0, 1) or 512 halves (one 1 KB block)
    const float* wtile[NT];
    const _Float16* wtile16[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int64_t blk = ((int64_t)((n0 >> 4) + nt) * (a.K >> 5) + (kbeg >> 5)) * 512;
        wtile[nt] = a.W + blk + lane * 4;
        wtile16[nt] = reinterpret_cast<const _Float16*>(a.W) + blk + lane * 8;
    }
    const float* gtile = (F16 && LN) ? a.ln_g + kbeg + 8 * g : nullptr;   // gamma of this lane's 8 k per chunk

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 wf[DEPTH][NT][2], af[DEPTH][MT][2], gf[DEPTH][2];
    h16x8 wh[DEPTH][NT];
    auto load_chunk = [&](int buf, int ch) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (F16) {
                wh[buf][nt] = *reinterpret_cast<const h16x8*>(wtile16[nt] + ch * 512);
            } else {
                wf[buf][nt][0] = ld4(wtile[nt] + ch * 512);
                wf[buf][nt][1] = ld4(wtile[nt] + ch * 512 + 256);
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            af[buf][mt][0] = ld4(atile[mt] + ch * 2048);
            af[buf][mt][1] = ld4(atile[mt] + ch * 2048 + 256);
        }
        if (F16 && LN) {
            gf[buf][0] = ld4(gtile + ch * 32);
            gf[buf][1] = ld4(gtile + ch * 32 + 4);
        }
    };
    auto compute_chunk = [&](int buf) {
        if (F16) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float4 x0 = af[buf][mt][0], x1 = af[buf][mt][1];
                if (LN) {
                    const float4 g0 = gf[buf][0], g1 = gf[buf][1];
                    x0 = make_float4(x0.x * g0.x, x0.y * g0.y, x0.z * g0.z, x0.w * g0.w);
                    x1 = make_float4(x1.x * g1.x, x1.y * g1.y, x1.z * g1.z, x1.w * g1.w);
                }
                const h16x8 x8 = to_h8(x0, x1);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[buf][nt], x8, acc[mt][nt], 0, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const float4 x = af[buf][mt][h];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const float4 w4 = wf[buf][nt][h];
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.x, x.x, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.y, x.y, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.z, x.z, acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.w, x.w, acc[mt][nt], 0, 0, 0);
                }
            }
        }
    };
    // every wave puts its first DEPTH k-chunks in flight.  LN: branch-free (a short K re-reads its last chunk), so
    // that the compiler can wait for the statistics alone -- behind a branch it would wait for every load
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        if (NCH) load_chunk(d, d);
        else if (LN) load_chunk(d, d < nchunk ? d : nchunk - 1);
        else if (d < nchunk) load_chunk(d, d);
    }
    // QKV epilogue, first hop of its page lookup (branch-free, behind the operand loads): where the row's
    // new token goes.  The second hop and the other epilogue operands are requested after the K loop.
    const int e_lr = tid / (4 * NT), e_q = tid % (4 * NT);   // this thread's first epilogue item
    const int e_row = m0 + e_lr, e_n = n0 + 4 * e_q;
    int e_ctx = 0, e_len = 0;
    if (EPI == EPI_QKV) {
        const int pb = e_row < a.M ? e_row / a.T : 0;
        e_ctx = a.ctx_len[pb];
        e_len = (a.lens ? a.lens : a.ctx_len)[pb];
    }
    if (LN) {
        __builtin_amdgcn_sched_barrier(0);   // loads above, merge below: the wait for the statistics is then an exact vmcnt
        // merge the partials while the operand loads are in flight (results are only needed in the epilogue, behind the
        // workgroup barrier): every partial covers the same number of columns (part_cnt), so
        //   mean = average of the tile means,  M2 = sum M2_t + part_cnt * sum (mean_t - mean)^2
        const bool v0 = 2 * s_j < a.n_part, v1 = 32 + 2 * s_j < a.n_part;
        float sm = (v0 ? st0.x + st0.z : 0.f) + (v1 ? st1.x + st1.z : 0.f);
        sm = row16_sum(sm);
        // No IEEE division / square root here: this block sits in front of the K loop on every wave's critical path, and the two
        // divisions + sqrt of the first version were ~100 dependent VALU instructions = 0.45 us per LayerNorm kernel
        // (tools/skinny_ab.py: 4.39 vs 3.95 us for the same GEMM with and without the fold).  The host passes the reciprocals;
        // rstd comes from v_rsq_f32 (1 ulp).
        const float mean = sm * a.inv_n_part;
        const float cnt = (float)a.part_cnt;
        const float d0 = st0.x - mean, d1 = st0.z - mean, d2 = st1.x - mean, d3 = st1.z - mean;
        float m2 = (v0 ? (st0.y + cnt * d0 * d0) + (st0.w + cnt * d1 * d1) : 0.f) +
                   (v1 ? (st1.y + cnt * d2 * d2) + (st1.w + cnt * d3 * d3) : 0.f);
        m2 = row16_sum(m2);
        if (tid < ROWS * 16 && s_j == 0) {
            s_mean[s_lr] = s_ok ? mean : 0.f;
            s_rstd[s_lr] = s_ok ? __builtin_amdgcn_rsqf(fmaf(m2, a.inv_k, a.eps)) : 0.f;
        }
    }
    MGEA_TS(1);
    if (ts) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    MGEA_TS(2);
    if (NCH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) compute_chunk(d);   // every chunk is already in flight
    } else {
        for (int ch = 0; ch < nchunk; ch += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                if (ch + d < nchunk) {
                    compute_chunk(d);
                    if (ch + d + DEPTH < nchunk) load_chunk(d, ch + d + DEPTH);
                }
            }
        }
    }

    MGEA_TS(3);
    // the epilogue's own global operands are requested now, so that their round trip runs under the MFMA
    // drain, the LDS reduction and the barrier instead of after them
    float4 e_bias = make_float4(0.f, 0.f, 0.f, 0.f), e_x = e_bias, e_c1 = e_bias;
    int e_phys = 0;
    if (tid < ROWS * 4 * NT) {
        if (EPI != EPI_LOGITS && a.bias) e_bias = ld4(a.bias + e_n);
        if (LN) e_c1 = ld4(a.ln_c1 + e_n);
        if (EPI == EPI_RES && e_row < a.M) e_x = ld4(a.out + tiled_off(e_row, e_n, a.N));
        if (EPI == EPI_QKV && e_row < a.M && e_n >= a.C) {
            const int page = (e_ctx + e_row % a.T) >> 6;
            if (page < a.max_pages) e_phys = a.page_table[(e_row / a.T) * a.max_pages + page];
        }
    }
    // partial tile of this wave -> LDS: D[i = column 4g + r][j = row c]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            *reinterpret_cast<float4*>(&red[((wave * ROWS) + mt * 16 + c) * PITCH + nt * 16 + 4 * g]) =
                make_float4(acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]);
    __syncthreads();
    MGEA_TS(4);

    // epilogue: thread t -> local row t / (4*NT), columns n0 + 4*(t % (4*NT)) .. +3
    for (int t = tid; t < ROWS * 4 * NT; t += NW * 64) {
        const int lr = t / (4 * NT), q = t % (4 * NT), row = m0 + lr;
        const int n = n0 + 4 * q;
        // the NW partial tiles are read at once (a rolled loop serialises one LDS round trip per wave)
        // and summed in wave order -> deterministic
        float4 pv[8];   // NW <= 8 (pick_waves)
#pragma unroll
        for (int w = 0; w < 8; ++w)
            pv[w] = w < NW ? *reinterpret_cast<const float4*>(&red[(w * ROWS + lr) * PITCH + 4 * q]) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 v = pv[0];
#pragma unroll
        for (int w = 1; w < 8; ++w)
            if (w < NW) v = add4(v, pv[w]);
        const bool row_ok = row < a.M;
        const bool first = t == tid;   // the operands requested before the reduction belong to this pass
        if (LN) {   // rstd * (x @ W'^T - mean * c1); c2 comes in as the bias
            const float mu = s_mean[lr], rs = s_rstd[lr];
            const float4 c1 = first ? e_c1 : ld4(a.ln_c1 + n);
            v = make_float4(rs * (v.x - mu * c1.x), rs * (v.y - mu * c1.y), rs * (v.z - mu * c1.z), rs * (v.w - mu * c1.w));
        }
        if (EPI != EPI_LOGITS) {
            // N % 16 == 0 for these epilogues (checked on the host)
            if (a.bias) v = add4(v, first ? e_bias : ld4(a.bias + n));
        }
        if (EPI == EPI_ACT) {
            if (a.act == ACT_GELU) v = make_float4(gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w));
            if (a.act == ACT_RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
            if (row_ok) st4(a.out + tiled_off(row, n, a.N), v);   // hbuf is k-tiled for the FC2 kernel
        }
        if (EPI == EPI_RES) {
            float* xp = a.out + tiled_off(row, n, a.N);           // residual stream x is k-tiled
            if (row_ok) {
                v = add4(v, first ? e_x : ld4(xp));
                st4(xp, v);
            }
            // (mean, M2) of this row over the tile's 16 columns, for the next LayerNorm
            float s = (v.x + v.y) + (v.z + v.w);
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            const float mean = s * (1.0f / 16.0f);
            const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
            float m2 = (dx * dx + dy * dy) + (dz * dz + dw * dw);
            m2 += __shfl_xor(m2, 1, 64);
            m2 += __shfl_xor(m2, 2, 64);
            if ((q & 3) == 0 && row_ok && a.stats_out)
                *reinterpret_cast<float2*>(a.stats_out + ((int64_t)row * (a.N >> 4) + blockIdx.x * NT + (q >> 2)) * 2) = make_float2(mean, m2);
        }
        if (EPI == EPI_QKV) {
            const int b = row_ok ? row / a.T : 0, tt = row_ok ? row % a.T : 0;
            const bool real = row_ok && (a.lens ? (tt < (first ? e_len : a.lens[b])) : true);
            if (!real) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row_ok) st4(a.out + (int64_t)row * a.ldo + n, v);
            if (real && n >= a.C) {
                const int pos = (first ? e_ctx : a.ctx_len[b]) + tt;
                const int page = pos >> 6, slot = pos & 63;
                if (page < a.max_pages) {
                    const int phys = first ? e_phys : a.page_table[b * a.max_pages + page];
                    const int isv = n >= 2 * a.C;
                    const int nn = n - (isv ? 2 * a.C : a.C);
                    kv_store4(a.pool, a.layer, phys, isv, nn / a.pool.dh, slot, nn % a.pool.dh, v);
                }
            }
        }
        if (EPI == EPI_LOGITS) {
            float e[4] = {v.x, v.y, v.z, v.w};
            float best = -INFINITY;
            int bi = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (n + j < a.N) {
                    e[j] += a.bias ? a.bias[n + j] : 0.f;
                    if (row_ok && a.out) a.out[(int64_t)row * a.ldo + n + j] = e[j];
                    if (e[j] > best) { best = e[j]; bi = n + j; }
                }
            }
#pragma unroll
            for (int o = 1; o < 4 * NT; o <<= 1) {
                const float ov = __shfl_xor(best, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            if (q == 0 && row_ok && a.pmax_val) {
                a.pmax_val[(int64_t)row * n_tiles + blockIdx.x] = best;
                a.pmax_idx[(int64_t)row * n_tiles + blockIdx.x] = bi;
            }
        }
    }
    if (ts) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    MGEA_TS(5);
#undef MGEA_TS
}

// rows per workgroup: enough workgroups to cover the chip, as few re-reads of W as possible
static int pick_mt(int M, int N) {
    // measured on MI355X (tools/skinny_bench.py, profiles/r1_skinny_sweep.txt): 32 rows per workgroup
    // once N/16 >= 64 tiles (QKV, FC1, head), 16 rows for the N = 512 projections; 64 rows never wins
    const int tiles_n = ceil_div(N, 16);
    int mt = tiles_n >= 64 ? 2 : 1;
    if (M > 64) {   // more row tiles: the tallest tile (fewest re-reads of W through L2) that still leaves >= 256 workgroups
        mt = 4;
        while (mt > 1 && tiles_n * ceil_div(M, 16 * mt) < 256) mt /= 2;
    }
    while (mt > 1 && 16 * (mt / 2) >= M) mt /= 2;  // never more rows than the problem has
    return mt;
}

static int pick_waves(int K, bool ln, int mt) {
    // as many waves as there are 32-wide k-chunks, <= 16 (<= 8 for the register-heavy LN x 64-row
    // variant): few chunks per wave = few dependent memory round trips.  The LN merge needs
    // 4 threads per row of the tile.
    const int cap = 8;  // 16 waves per workgroup measured equal or slower on every decode shape
    for (int nw = cap; nw >= 1; --nw)
        if ((K / 32) % nw == 0) return nw;
    return 1;
}

template <int EPI, int MT, int NT = 1>
static int launch_skinny_mt(const SkinnyArgs& a_in, int nw, hipStream_t st) {
    const bool ln = a_in.ln_c1 != nullptr;
    SkinnyArgs a = a_in;
    a.nw = nw;
    if (ln) {
        a.inv_n_part = (float)(1.0 / (double)a.n_part);
        a.inv_k = (float)(1.0 / ((double)a.n_part * (double)a.part_cnt));
    }
    dim3 grid((unsigned)round_up(ceil_div(a.N, 16 * NT), 8), ceil_div(a.M, 16 * MT)), block(64 * nw);
    const size_t shmem = ((size_t)nw * 16 * MT * (16 * NT + 4) + (ln ? 2 * 16 * MT : 0)) * sizeof(float);
    // compile-time chunk counts for the decode shapes (8 waves x NCH chunks of 32): K = 512 (NCH 2), 768 (3), 2048 (8); anything
    // else, other wave counts and the timestamp mode of tools/skinny_phases.py take the generic stream (NCH 0)
    const int nch = (nw == 8 && !(a.dbg & (1 << 20)) && a.K % 256 == 0) ? a.K / 256 : 0;
    // switch skinny_one_per_cu (A/B, round 4): pad the dynamic LDS request beyond half a CU's 160 KB so that no two workgroups of a launch with
    // <= 256 workgroups can share a CU (the head kernel of head_gemm.hip relies on the same effect)
    const bool one_per_cu = tune(TUNE_SKINNY_ONE_PER_CU) && grid.x * grid.y <= 256;
    const size_t shmem_go = one_per_cu && shmem < 84 * 1024 ? 84 * 1024 : shmem;
#define MGEA_SKINNY_GO(LNV, NTV, F16V, NCHV)                                                                                              \
    do {                                                                                                                                 \
        if (one_per_cu) {                                                                                                                \
            static uint64_t attr_done = 0;                                                                                               \
            DeviceInfo di;                                                                                                               \
            MGEA_TRY(device_info(&di));                                                                                                  \
            MGEA_TRY(set_max_dynamic_lds(reinterpret_cast<const void*>(&gemm_skinny_kernel<EPI, LNV, MT, NTV, F16V, NCHV>), 84 * 1024, di.dev, &attr_done)); \
        }                                                                                                                                \
        hipLaunchKernelGGL((gemm_skinny_kernel<EPI, LNV, MT, NTV, F16V, NCHV>), grid, block, shmem_go, st, a);                           \
    } while (0)
#define MGEA_SKINNY_NCH(LNV, NTV, F16V)                                                                  \
    do {                                                                                                 \
        if (nch == 2) MGEA_SKINNY_GO(LNV, NTV, F16V, 2);                                                  \
        else if (nch == 3) MGEA_SKINNY_GO(LNV, NTV, F16V, 3);                                             \
        else if (nch == 8 && MT * NTV <= 2) MGEA_SKINNY_GO(LNV, NTV, F16V, 8);                            \
        else MGEA_SKINNY_GO(LNV, NTV, F16V, 0);                                                           \
    } while (0)
    if (a.w_f16) {
        if (ln && NT == 1) MGEA_SKINNY_NCH(true, 1, true);
        else               MGEA_SKINNY_NCH(false, NT, true);
    } else {
        if (ln && NT == 1) MGEA_SKINNY_NCH(true, 1, false);
        else               MGEA_SKINNY_NCH(false, NT, false);
    }
#undef MGEA_SKINNY_NCH
#undef MGEA_SKINNY_GO
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

template <int EPI>
static int launch_skinny_t(const SkinnyArgs& a, hipStream_t st) {
    const bool ln = a.ln_c1 != nullptr;
    MGEA_REQUIRE(a.M >= 1 && a.M <= MGEA_FUSED_MAX_ROWS, MGEA_EINVAL, "skinny gemm: M=%d not in 1..%d", a.M, MGEA_FUSED_MAX_ROWS);
    MGEA_REQUIRE(a.K % 32 == 0, MGEA_EINVAL, "skinny gemm: K=%d must be a multiple of 32", a.K);
    MGEA_REQUIRE(EPI == EPI_LOGITS || a.N % 16 == 0, MGEA_EINVAL, "skinny gemm: N=%d must be a multiple of 16", a.N);
    if (EPI == EPI_LOGITS) {   // decode-step head: one balanced round of the chip where the shape allows (head_gemm.hip)
        const int rc = launch_head_balanced(a, st);
        if (rc != 1) return rc;
    }
    if (EPI == EPI_LOGITS && !ln && a.M > 32 && a.N >= 4096 && !((a.dbg >> 8) & 0x1FF)) {
        // LM head: 32 rows x 32 columns per workgroup, W fetched from HBM once
        const int nw_head = pick_waves(a.K, false, 2);
        return launch_skinny_mt<EPI_LOGITS, 2, 2>(a, nw_head, st);
    }
    int mt = ((a.dbg >> 8) & 15) ? ((a.dbg >> 8) & 15) : pick_mt(a.M, a.N);
    int nw = pick_waves(a.K, ln, mt);
    if ((a.dbg >> 12) & 31) nw = (a.dbg >> 12) & 31;  // tools/skinny_bench.py override
    while (ln && mt > 1 && nw * 64 < 16 * mt * 16) mt /= 2;   // the LN merge wants 16 lanes per tile row
    MGEA_REQUIRE((a.K / 32) % nw == 0 && nw <= 8, MGEA_EINVAL, "skinny gemm: bad wave count %d", nw);
    MGEA_REQUIRE(!ln || a.n_part <= 64, MGEA_EINVAL, "skinny gemm: more than 64 LayerNorm partials per row (%d)", a.n_part);
    MGEA_REQUIRE(!(ln && a.w_f16) || a.ln_g, MGEA_EINVAL, "skinny gemm: the fp16 folded LayerNorm needs gamma (ln_g)");
    MGEA_REQUIRE(!ln || a.n_part % 2 == 0, MGEA_EINVAL, "skinny gemm: odd number of LayerNorm partials per row (%d)", a.n_part);
    MGEA_REQUIRE(!ln || nw * 64 >= 16 * mt * 16, MGEA_EINVAL,
                 "skinny gemm: the LayerNorm merge needs 16 lanes per tile row (K=%d, %d-row tiles, %d waves)", a.K, 16 * mt, nw);
    switch (mt) {
        case 1: return launch_skinny_mt<EPI, 1>(a, nw, st);
        case 2: return launch_skinny_mt<EPI, 2>(a, nw, st);
        case 4: return launch_skinny_mt<EPI, 4>(a, nw, st);
    }
    set_error("skinny gemm: bad row-tile count %d", mt);
    return MGEA_EINVAL;
}

// number of per-row partial (max, argmax) entries the LOGITS epilogue writes = its grid.x
int skinny_logits_tiles(int M, int N, int K) {   // K = 0: the generic kernel's count whatever the shape
    const int g = K > 0 ? head_balanced_partials(M, N, K) : 0;
    if (g) return g;
    return (M > 32 && N >= 4096) ? ceil_div(N, 32) : ceil_div(N, 16);
}

int launch_skinny(int epi, const SkinnyArgs& a, hipStream_t st) {
    switch (epi) {
        case EPI_QKV: return launch_skinny_t<EPI_QKV>(a, st);
        case EPI_RES: return launch_skinny_t<EPI_RES>(a, st);
        case EPI_ACT: return launch_skinny_t<EPI_ACT>(a, st);
        case EPI_LOGITS: return launch_skinny_t<EPI_LOGITS>(a, st);
    }
    set_error("skinny gemm: bad epilogue %d", epi);
    return MGEA_EINVAL;
}

// ------------------------------------------------------------------------------------------
// W [N, K] row-major -> fragment-ordered tiles (common.h); one thread per float4 of the output
__global__ __launch_bounds__(256) void tile_weights_kernel(const float* __restrict__ W, const float* __restrict__ gamma, int N, int K,
                                                           float* __restrict__ out, int64_t n_f4) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= n_f4) return;
    const int64_t blk = o >> 7;            // (tile, chunk) block of 128 float4
    const int in = (int)(o & 127), h = in >> 6, lane = in & 63, g = lane >> 4, c = lane & 15;
    const int chunks = K >> 5;
    const int64_t tile = blk / chunks;
    const int kc = (int)(blk % chunks);
    const int64_t n = tile * 16 + c;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const int k = kc * 32 + 8 * g + 4 * h;
    if (n < N) v = ld4(W + n * K + k);
    if (gamma) {   // folded LayerNorm: the same fp32 products ln_fold_vectors_kernel sums
        const float4 gm = ld4(gamma + k);
        v = make_float4(v.x * gm.x, v.y * gm.y, v.z * gm.z, v.w * gm.w);
    }
    st4(out + o * 4, v);
}

int launch_tile_weights(const float* W, int N, int K, float* out, hipStream_t st, const float* gamma) {
    MGEA_REQUIRE(W && out && N >= 1 && K >= 32 && K % 32 == 0, MGEA_EINVAL, "tile_weights: N=%d K=%d (K must be a multiple of 32)", N, K);
    const int64_t n_f4 = wtile_floats(N, K) / 4;
    hipLaunchKernelGGL(tile_weights_kernel, dim3((unsigned)((n_f4 + 255) / 256)), dim3(256), 0, st, W, gamma, N, K, out, n_f4);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

// fp16 fragments: one thread per 16-byte group (8 halves) of the output
__global__ __launch_bounds__(256) void tile_weights_f16_kernel(const float* __restrict__ W, int N, int K, _Float16* __restrict__ out, int64_t n_g8) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= n_g8) return;
    const int64_t blk = o >> 6;            // (tile, chunk) block of 64 lanes
    const int lane = (int)(o & 63), g = lane >> 4, c = lane & 15;
    const int chunks = K >> 5;
    const int64_t tile = blk / chunks;
    const int kc = (int)(blk % chunks);
    const int64_t n = tile * 16 + c;
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
    if (n < N) {
        v0 = ld4(W + n * K + kc * 32 + 8 * g);
        v1 = ld4(W + n * K + kc * 32 + 8 * g + 4);
    }
    *reinterpret_cast<h16x8*>(out + o * 8) = to_h8(v0, v1);
}

int launch_tile_weights_f16(const float* W, int N, int K, void* out, hipStream_t st) {
    MGEA_REQUIRE(W && out && N >= 1 && K >= 32 && K % 32 == 0, MGEA_EINVAL, "tile_weights_f16: N=%d K=%d (K must be a multiple of 32)", N, K);
    const int64_t n_g8 = wtile_floats(N, K) / 8;
    hipLaunchKernelGGL(tile_weights_f16_kernel, dim3((unsigned)((n_g8 + 255) / 256)), dim3(256), 0, st, W, N, K, static_cast<_Float16*>(out), n_g8);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

__global__ __launch_bounds__(256) void round_f16_inplace_kernel(float* __restrict__ x, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = (float)(_Float16)x[i];
}

int launch_round_f16_inplace(float* x, int64_t n, hipStream_t st) {
    hipLaunchKernelGGL(round_f16_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

// c1[n] = sum_k fl(gamma[k] * W[n,k]) (EXACT = 0: the fp32 products the tiled copy stores) or sum_k gamma[k] * W[n,k] (EXACT = 1:
// gamma applied on the activation side),  c2[n] = sum_k beta[k] * W[n,k] + bias[n]; one wave per row, fp64 sums
template <int EXACT>
__global__ __launch_bounds__(256) void ln_fold_vectors_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ bias, int N, int K,
                                                              float* __restrict__ c1, float* __restrict__ c2) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    double s1 = 0.0, s2 = 0.0;
    for (int k = lane; k < K; k += 64) {
        const float w = W[(int64_t)n * K + k];
        s1 += EXACT ? (double)w * (double)gamma[k] : (double)(w * gamma[k]);
        s2 += (double)w * (double)beta[k];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    if (lane == 0) {
        c1[n] = (float)s1;
        c2[n] = (float)(s2 + (bias ? (double)bias[n] : 0.0));
    }
}

int launch_ln_fold(const float* W, const float* gamma, const float* beta, const float* bias, int N, int K, float* out,
                   float* c1, float* c2, hipStream_t st) {
    MGEA_REQUIRE(gamma && beta && c1 && c2, MGEA_EINVAL, "ln_fold: NULL argument");
    MGEA_TRY(launch_tile_weights(W, N, K, out, st, gamma));
    hipLaunchKernelGGL(ln_fold_vectors_kernel<0>, dim3((unsigned)ceil_div(N, 4)), dim3(256), 0, st, W, gamma, beta, bias, N, K, c1, c2);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

int launch_ln_vectors(const float* W, const float* gamma, const float* beta, const float* bias, int N, int K, float* c1, float* c2,
                      hipStream_t st) {
    MGEA_REQUIRE(W && gamma && beta && c1 && c2, MGEA_EINVAL, "ln_vectors: NULL argument");
    hipLaunchKernelGGL(ln_fold_vectors_kernel<1>, dim3((unsigned)ceil_div(N, 4)), dim3(256), 0, st, W, gamma, beta, bias, N, K, c1, c2);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

__global__ __launch_bounds__(256) void tile_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int M, int N, int to_tiled) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // float4 index into the row-major matrix
    if (i >= (int64_t)M * (N >> 2)) return;
    const int row = (int)(i / (N >> 2)), n = (int)(i % (N >> 2)) * 4;
    if (to_tiled) st4(dst + tiled_off(row, n, N), ld4(src + (int64_t)row * N + n));
    else          st4(dst + (int64_t)row * N + n, ld4(src + tiled_off(row, n, N)));
}

int launch_tile_rows(const float* src, float* dst, int M, int N, int to_tiled, hipStream_t st) {
    MGEA_REQUIRE(src && dst && M >= 1 && M <= MGEA_FUSED_MAX_ROWS && N >= 32 && N % 32 == 0, MGEA_EINVAL, "tile_rows: M=%d (1..%d) N=%d (multiple of 32)", M, MGEA_FUSED_MAX_ROWS, N);
    const int64_t n = (int64_t)M * (N >> 2);
    hipLaunchKernelGGL(tile_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, M, N, to_tiled);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

// ------------------------------------------------------------------------------------------
// x[m] = tok_emb[ids[m]] + pos_emb[pos] (k-tiled); stats[m] = (mean, M2) of the row, as two half-row partials
__global__ __launch_bounds__(256) void embed_stats_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ lens,
                                                         const int32_t* __restrict__ ctx_len,
                                                         const float* __restrict__ tok_emb,
                                                         const float* __restrict__ pos_emb, float* __restrict__ x,
                                                         float* __restrict__ stats, int T, int C, int vocab,
                                                         int pos_rows, int absolute_pos, int32_t* __restrict__ err_flag) {
    __shared__ float redv[4];
    const int64_t m = blockIdx.x;
    const int b = (int)(m / T), t = (int)(m % T);
    const int nf4 = C >> 2;
    const bool real = lens ? (t < lens[b]) : true;
    int id = ids[m];
    // an id outside the vocabulary (nn.Embedding raises IndexError, api_cache.py:99) is clamped so that the loads stay in
    // bounds and reported through the engine's sticky error flag (mgea_decoder_error_flags) -- no host sync per call
    if (real && (id < 0 || id >= vocab) && err_flag && threadIdx.x == 0) atomicOr(err_flag, 1);
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    int pos = t + ((absolute_pos && ctx_len) ? ctx_len[b] : 0);
    pos = pos < pos_rows ? pos : pos_rows - 1;
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + i * 256;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < nf4) {
            if (real) v[i] = add4(ld4(tok_emb + (int64_t)id * C + f * 4), ld4(pos_emb + (int64_t)pos * C + f * 4));
            st4(x + tiled_off((int)m, f * 4, C), v[i]);
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    auto bsum = [&](float val) {
        val = wave_sum(val);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) redv[threadIdx.x >> 6] = val;
        __syncthreads();
        return (redv[0] + redv[1]) + (redv[2] + redv[3]);
    };
    const float mean = bsum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = threadIdx.x + i * 256;
        if (f < nf4) {
            const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    }
    const float m2 = bsum(q);
    // two equal half-row partials (the consumer merges pairs): (mean, M2/2) x 2 merges back to exactly (mean, M2)
    if (threadIdx.x == 0) st4(stats + m * 4, make_float4(mean, 0.5f * m2, mean, 0.5f * m2));
}

int launch_embed_stats(const int32_t* ids, const int32_t* lens, const int32_t* ctx_len, const float* tok_emb,
                       const float* pos_emb, float* x, float* stats, int B, int T, int C, int vocab, int pos_rows,
                       int absolute_pos, int32_t* err_flag, hipStream_t st) {
    MGEA_REQUIRE(C % 4 == 0 && C <= 4096, MGEA_EINVAL, "embed: d_model=%d must be a multiple of 4 and <= 4096", C);
    MGEA_REQUIRE(B * T <= MGEA_FUSED_MAX_ROWS, MGEA_EINVAL, "embed (fused path): more than %d rows", MGEA_FUSED_MAX_ROWS);
    hipLaunchKernelGGL(embed_stats_kernel, dim3(B * T), dim3(256), 0, st, ids, lens, ctx_len, tok_emb, pos_emb, x, stats,
                       T, C, vocab, pos_rows, absolute_pos, err_flag);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

// ------------------------------------------------------------------------------------------
// greedy finalize: argmax over the per-tile partials of each row, then the sampler-loop
// bookkeeping of api_cache.py:179-181 (append, EOS stop) -- one 64-thread workgroup per row.
__global__ __launch_bounds__(64) void argmax_advance_kernel(const float* __restrict__ pval, const int32_t* __restrict__ pidx,
                                                           int n_tiles, StepState s, int32_t* __restrict__ sampled) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = lane; i < n_tiles; i += 64) {
        const float v = pval[(int64_t)b * n_tiles + i];
        const int ix = pidx[(int64_t)b * n_tiles + i];
        if (v > best || (v == best && ix < bi)) { best = v; bi = ix; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) {
        const int tok = bi == 0x7fffffff ? 0 : bi;
        sampled[b] = tok;
        const int step = s.row_step[b];
        int out = -1;
        if (!s.done[b]) {
            out = tok;
            s.cur_ids[b] = tok;
            s.ctx_len[b] += 1;
            if (tok == s.eos()) {
                s.done[b] = 1;
                atomicAdd(s.n_done, 1);
            }
        }
        if (s.ids_out && step < s.n_steps) s.ids_out[(int64_t)b * s.n_steps + step] = out;
        s.row_step[b] = step + 1;
    }
}

// Same finalize fused with the NEXT step's embedding (generate() keeps x "primed"): one launch less
// per decode step.  256 threads per row: partial-argmax reduce, bookkeeping by thread 0, then
// x[row] = tok_emb[token] + pos_emb[pos] (k-tiled) and its LayerNorm partial statistics.
__global__ __launch_bounds__(256) void argmax_advance_embed_kernel(const float* __restrict__ pval,
                                                                  const int32_t* __restrict__ pidx, int n_tiles,
                                                                  TailArgs t, int32_t* __restrict__ sampled) {
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ float sh[8];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the row's bookkeeping state is requested together with the partials (one round trip instead of two)
    int st_step = 0, st_fed = 0, st_len = 0, st_done = 0;
    if (tid == 0) { st_step = t.s.row_step[b]; st_fed = t.s.cur_ids[b]; st_len = t.s.ctx_len[b]; st_done = t.s.done[b]; }
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = tid; i < n_tiles; i += 256) {
        const float v = pval[(int64_t)b * n_tiles + i];
        const int ix = pidx[(int64_t)b * n_tiles + i];
        if (v > best || (v == best && ix < bi)) { best = v; bi = ix; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) { sv[wave] = best; si[wave] = bi; }
    __syncthreads();
    int tok = 0;
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
        tok = bi == 0x7fffffff ? 0 : bi;
    }
    advance_embed_row(b, tok, t, sampled, st_step, st_fed, st_len, st_done, sh);
}

int launch_argmax_advance_embed(const float* pval, const int32_t* pidx, int n_tiles, const StepState& s, int32_t* sampled,
                                const float* tok_emb, const float* pos_emb, float* x, float* stats, int B, int C, int vocab,
                                int pos_rows, int absolute_pos, hipStream_t st) {
    MGEA_REQUIRE(B <= MGEA_FUSED_MAX_ROWS && C % 4 == 0 && C <= 4096, MGEA_EINVAL, "argmax+embed: bad shape");
    TailArgs t{s, tok_emb, pos_emb, x, stats, C, vocab, pos_rows, absolute_pos};
    hipLaunchKernelGGL(argmax_advance_embed_kernel, dim3(B), dim3(256), 0, st, pval, pidx, n_tiles, t, sampled);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

int launch_argmax_advance(const float* pval, const int32_t* pidx, int n_tiles, const StepState& s, int32_t* sampled,
                          int B, hipStream_t st) {
    hipLaunchKernelGGL(argmax_advance_kernel, dim3(B), dim3(64), 0, st, pval, pidx, n_tiles, s, sampled);
    MGEA_CHECK_HIP(hipGetLastError());
    return MGEA_OK;
}

}  // namespace mgea
