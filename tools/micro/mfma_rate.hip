// Sustained rate of v_mfma_f32_16x16x4_f32 on MI355X: CHAINS independent accumulator chains per wave,
// WAVES waves per workgroup (one workgroup per CU when WAVES = 4 or 8), no memory traffic in the loop.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o gpurun_out/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ __launch_bounds__(1024) void mfma_loop(float* out, int iters, float a0, float b0) {
    f32x4 acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (s == 123.456f) out[0] = s;
}

template <int CHAINS>
void run(int waves, int blocks, float* out) {
    const int iters = 4096 / CHAINS;   // 16384 MFMAs per wave
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<CHAINS>, dim3(blocks), dim3(64 * waves), 0, 0, out, iters, 1.0f, 2.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<CHAINS>, dim3(blocks), dim3(64 * waves), 0, 0, out, iters, 1.0f, 2.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = (double)blocks * waves * iters * 4 * CHAINS;
    const double tf = n_mfma * 2048.0 / (ms * 1e-3) / 1e12;
    // cycles per MFMA per SIMD at 2.4 GHz, assuming waves spread evenly over the 4 SIMDs of `blocks`/256 CUs
    const double per_simd = n_mfma / (256.0 * 4.0) * (blocks > 256 ? 1.0 : 256.0 / blocks);
    printf("chains %d waves/wg %2d blocks %4d: %8.3f ms  %7.1f TFLOP/s  %5.1f clk@2.4GHz per MFMA per SIMD\n", CHAINS, waves, blocks, ms, tf,
           ms * 1e-3 * 2.4e9 / per_simd);
}

int main() {
    float* out;
    hipMalloc(&out, 4);
    for (int waves : {4, 8, 16}) {
        run<1>(waves, 256, out);
        run<2>(waves, 256, out);
        run<4>(waves, 256, out);
        run<8>(waves, 256, out);
    }
    run<8>(8, 512, out);
    run<1>(8, 128, out);
    return 0;
}
