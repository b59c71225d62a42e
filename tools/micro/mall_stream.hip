// Can the decode step's KV streaming be moved under its latency-bound GEMM kernels?  The GEMM chain of a B = 64 step
// (148 us) leaves HBM idle; the paged attention (138 us) is HBM-bound.  If a cheap "touch" of the next layer's KV pages during the
// GEMM kernels leaves them in the 256 MiB Infinity Cache (MALL), the attention kernel streams from MALL instead of HBM.
// This program measures what that is worth, on a KV-sized buffer (default 135 MB):
//   cold    : 16-B-per-lane non-temporal streaming read after 1 GiB of other traffic (= today's attention: HBM)
//   warm    : the same read immediately again (data as resident as a full read leaves it)
//   touch64 / touch128 + read : one dword per 64 B / 128 B by a small grid (the prefetch), then the full streaming read
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mall_stream.hip -o tools/micro/mall_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void stream_read(const f32x4* __restrict__ p, size_t n16, float* sink, int nt) {
    // each wave reads 16 x 1 KiB pieces per iteration (as the attention kernel does: 16 KiB in flight per wave)
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
    const int lane = threadIdx.x & 63;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t base = wave * 1024; base + 1024 <= n16; base += nwaves * 1024) {
        f32x4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = nt ? __builtin_nontemporal_load(p + base + i * 64 + lane) : p[base + i * 64 + lane];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += v[i];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

__global__ __launch_bounds__(256) void touch(const float* __restrict__ p, size_t nbytes, int stride_b, float* sink, int nt) {
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
    const int lane = threadIdx.x & 63;
    const size_t per = (size_t)64 * stride_b * 16;          // bytes one wave iteration covers (16 loads x 64 lanes x stride)
    float acc = 0.f;
    for (size_t base = wave * per; base + per <= nbytes; base += nwaves * per) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float* q = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p) + base + ((size_t)i * 64 + lane) * stride_b);
            v[i] = nt ? __builtin_nontemporal_load(q) : *q;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += v[i];
    }
    if (acc == 123.456f) sink[0] = acc;
}

__global__ void fill(float* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0f;
}

int main(int argc, char** argv) {
    const size_t MB = argc > 1 ? atoi(argv[1]) : 135;
    const size_t nbytes = MB << 20, n16 = nbytes / 16;
    float *buf, *junk, *sink;
    CK(hipMalloc(&buf, nbytes)); CK(hipMalloc(&junk, (size_t)1 << 30)); CK(hipMalloc(&sink, 64));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, st, buf, nbytes / 4);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto evict = [&]() { hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, st, junk, ((size_t)1 << 30) / 4); };
    auto timed = [&](auto launch) { CK(hipEventRecord(e0, st)); launch(); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
                                   float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e3f; };
    auto rd = [&](int nt) { hipLaunchKernelGGL(stream_read, dim3(512), dim3(256), 0, st, (const f32x4*)buf, n16, sink, nt); };
    for (int rep = 0; rep < 2; ++rep) {
        evict(); float c = timed([&] { rd(1); });
        float w = timed([&] { rd(1); });
        printf("%zu MB  cold nt read %7.1f us (%5.2f TB/s) | warm nt re-read %7.1f us (%5.2f TB/s)\n", MB, c, nbytes / c / 1e6, w, nbytes / w / 1e6);
        evict(); c = timed([&] { rd(0); }); w = timed([&] { rd(0); });
        printf("%zu MB  cold    read %7.1f us (%5.2f TB/s) | warm    re-read %7.1f us (%5.2f TB/s)\n", MB, c, nbytes / c / 1e6, w, nbytes / w / 1e6);
        for (int stride = 64; stride <= 128; stride *= 2)
            for (int tnt = 0; tnt < 2; ++tnt)
                for (int grid = 128; grid <= 512; grid *= 4) {
                    evict();
                    float t = timed([&] { hipLaunchKernelGGL(touch, dim3(grid), dim3(256), 0, st, buf, nbytes, stride, sink, tnt); });
                    float r = timed([&] { rd(1); });
                    printf("%zu MB  touch 1 dword / %3d B (%s, %3d WGs) %7.1f us (%5.2f TB/s of lines) -> nt read %7.1f us (%5.2f TB/s)\n", MB, stride,
                           tnt ? "nt" : "plain", grid, t, nbytes / t / 1e6, r, nbytes / r / 1e6);
                }
    }
    return 0;
}
