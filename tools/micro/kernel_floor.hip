// What does one DEPENDENT kernel of a captured decode-step graph cost on MI355X, by ingredient?  The decode step at B <= 64 is
// a chain of 32 kernels of ~5.5 us each whatever the batch (tools/decode_batch_sweep.py): this program times chains of 64
// kernels replayed from a hipGraph, each variant adding one ingredient of the real skinny-GEMM kernels:
//   empty      : 256 workgroups x 512 threads, no memory access
//   args       : reads a 256-byte by-value argument struct (scalar loads from the kernarg segment), one store per workgroup
//   args_pre   : the same fields as plain scalar arguments (eligible for kernarg preload when built with
//                -mllvm -amdgpu-kernarg-preload-count=16)
//   load       : args + every lane loads 2 x 16 B of the PREVIOUS kernel's output (another CU wrote it) and stores 16 B
//   load_lds   : load + partial tiles through LDS and a workgroup barrier (the split-K reduction of the skinny GEMM)
//   load2      : load_lds + a second, dependent load round trip (page table -> page, or statistics -> operands)
//   big        : like the out-proj skinny GEMM's memory shape: every workgroup loads 32 KB of a static "weight" buffer (shared by
//                the 4 workgroups of a column tile) + 32 KB of the previous kernel's output (shared by the 32 of a row tile)
//   big_rmw    : big + the skinny epilogue: 8 per-wave partial tiles through LDS, ONE wave sums them, adds a residual it loaded
//                from the output buffer (read-modify-write of x) and stores 16 B per lane
//   big_mfma   : big_rmw + 16 dependent v_mfma_f32_16x16x4_f32 per wave (the exact-fp32 K loop of K = 512 split over 8 waves)
//   rot_code   : big_mfma, but consecutive kernels of the chain are 6 DIFFERENT copies of the code (like QKV / attention / out-proj / FC1 /
//                FC2 of a layer): does a kernel that follows different code pay for its instruction fetch?
//   rot_cfg    : big_mfma, consecutive kernels alternate launch configuration (128 workgroups / 8 KB LDS, 192 / 40 KB, 256 / 20 KB):
//                does the dispatcher charge for a change of grid size and LDS allocation between dependent kernels?
//   cold_w     : big_mfma with the 1 MB of "weights" of kernel i taken from one of 96 different 1 MB buffers (like the 92.7 MB of a
//                decode step: a kernel's weights were last read ~96 kernels ago, so they come from beyond the 32 MB of L2)
// build: hipcc -O3 --offload-arch=gfx950 [-mllvm -amdgpu-kernarg-preload-count=16] tools/micro/kernel_floor.hip -o tools/micro/kernel_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Args { const float* in; float* out; const int* idx; int n; int pad[57]; };   // 256 bytes like SkinnyArgs

__global__ __launch_bounds__(512) void k_empty() {}

__global__ __launch_bounds__(512) void k_args(Args a) {
    if (threadIdx.x == 0) a.out[blockIdx.x * 2048] = (float)(a.n + a.pad[56]);
}

__global__ __launch_bounds__(512) void k_args_pre(const float* in, float* out, const int* idx, int n, int p56) {
    if (threadIdx.x == 0) out[blockIdx.x * 2048] = (float)(n + p56);
}

template <int MODE>   // 0 load, 1 load + LDS reduce, 2 + dependent second round trip
__global__ __launch_bounds__(512) void k_load(Args a) {
    extern __shared__ float red[];
    const int t = threadIdx.x, b = blockIdx.x;
    int base = b;
    if (MODE == 2) base = a.idx[b];                      // dependent hop (written by nobody in the chain: L2 / MALL resident)
    const float4* src = reinterpret_cast<const float4*>(a.in) + (size_t)base * 1024 + t;
    float4 x = src[0], y = src[512];
    float4 v = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
    if (MODE >= 1) {
        reinterpret_cast<float4*>(red)[t] = v;
        __syncthreads();
        if (t < 128) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int w = 0; w < 4; ++w) { const float4 p = reinterpret_cast<float4*>(red)[t + 128 * w]; s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w; }
            reinterpret_cast<float4*>(a.out)[(size_t)b * 1024 + t] = s;
        }
    } else if (t < 128) {
        reinterpret_cast<float4*>(a.out)[(size_t)b * 1024 + t] = v;
    }
}

typedef float f4v __attribute__((ext_vector_type(4)));
template <int MODE, int ID = 0>   // 0 big, 1 big_rmw, 2 big_mfma; ID: distinct copies of the same code (different code addresses)
__global__ __launch_bounds__(512) void k_big(Args a, const float* wbuf) {
    extern __shared__ float red[];
    const int t = threadIdx.x, b = blockIdx.x, lane = t & 63, wave = t >> 6;
    const int ct = b & 31, rt = b >> 5;                   // 32 column tiles x (gridDim.x / 32) row tiles
    const float4* w = reinterpret_cast<const float4*>(wbuf) + (size_t)ct * 2048 + wave * 256 + lane;          // 32 KB per column tile
    const float4* x = reinterpret_cast<const float4*>(a.in) + (size_t)rt * 2048 + wave * 256 + lane;         // 32 KB per row tile
    float4 wv[4], xv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { wv[i] = w[i * 64]; xv[i] = x[i * 64]; }
    float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4* outp = reinterpret_cast<float4*>(a.out) + (size_t)rt * 2048 + ct * 64 + lane;   // 64 lanes x 16 B = this tile of the next input
    if (MODE >= 1 && wave == 0) r4 = *outp;               // residual, requested before the reduction
    f4v acc = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i].x, xv[i].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i].y, xv[i].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i].z, xv[i].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i].w, xv[i].w, acc, 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc[0] += wv[i].x * xv[i].x; acc[1] += wv[i].y * xv[i].y; acc[2] += wv[i].z * xv[i].z; acc[3] += wv[i].w * xv[i].w; }
    }
    if (MODE == 0) {
        if (wave == 0) *outp = make_float4(acc[0], acc[1], acc[2], acc[3]);
        else if (acc[0] == 123.456f) a.out[0] = acc[1];
        return;
    }
    reinterpret_cast<float4*>(red)[wave * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    __syncthreads();
    if (wave == 0) {
        float4 s = r4;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) { const float4 p = reinterpret_cast<float4*>(red)[w8 * 64 + lane]; s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w; }
        *outp = s;
    }
}

int main(int argc, char** argv) {
    const int NB = argc > 1 ? atoi(argv[1]) : 256, CH = 64, REP = 50;
    float *bufA, *bufB; int* idx;
    const size_t NBM = NB > 256 ? NB : 256;
    CK(hipMalloc(&bufA, NBM * 1024 * 16 * 2)); CK(hipMalloc(&bufB, NBM * 1024 * 16 * 2)); CK(hipMalloc(&idx, NB * 4));
    CK(hipMemset(bufA, 0, (size_t)NB * 1024 * 16 * 2)); CK(hipMemset(bufB, 0, (size_t)NB * 1024 * 16 * 2));
    int* h = (int*)malloc(NB * 4); for (int i = 0; i < NB; ++i) h[i] = (i * 37) % NB; CK(hipMemcpy(idx, h, NB * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    float* wbuf; CK(hipMalloc(&wbuf, (size_t)96 << 20)); CK(hipMemset(wbuf, 0, (size_t)96 << 20));
    const char* names[] = {"empty", "args", "args_pre", "load", "load_lds", "load2", "big", "big_rmw", "big_mfma", "cold_w", "rot_code", "rot_cfg"};
    for (int v = 0; v < 12; ++v) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < CH; ++i) {
            Args a{}; a.in = (i & 1) ? bufB : bufA; a.out = (i & 1) ? bufA : bufB; a.idx = idx; a.n = i;
            switch (v) {
                case 0: hipLaunchKernelGGL(k_empty, dim3(NB), dim3(512), 0, st); break;
                case 1: hipLaunchKernelGGL(k_args, dim3(NB), dim3(512), 0, st, a); break;
                case 2: hipLaunchKernelGGL(k_args_pre, dim3(NB), dim3(512), 0, st, a.in, a.out, a.idx, a.n, 0); break;
                case 3: hipLaunchKernelGGL(k_load<0>, dim3(NB), dim3(512), 0, st, a); break;
                case 4: hipLaunchKernelGGL(k_load<1>, dim3(NB), dim3(512), 8192, st, a); break;
                case 5: hipLaunchKernelGGL(k_load<2>, dim3(NB), dim3(512), 8192, st, a); break;
                case 6: hipLaunchKernelGGL(k_big<0>, dim3(NB), dim3(512), 8192, st, a, wbuf); break;
                case 7: hipLaunchKernelGGL(k_big<1>, dim3(NB), dim3(512), 8192, st, a, wbuf); break;
                case 8: hipLaunchKernelGGL(k_big<2>, dim3(NB), dim3(512), 8192, st, a, wbuf); break;
                case 9: hipLaunchKernelGGL(k_big<2>, dim3(NB), dim3(512), 8192, st, a, wbuf + (size_t)((i * 37) % 96) * (1 << 18)); break;
                case 10:
                    switch (i % 6) {
                        case 0: hipLaunchKernelGGL((k_big<2, 1>), dim3(NB), dim3(512), 8192, st, a, wbuf); break;
                        case 1: hipLaunchKernelGGL((k_big<2, 2>), dim3(NB), dim3(512), 8192, st, a, wbuf); break;
                        case 2: hipLaunchKernelGGL((k_big<2, 3>), dim3(NB), dim3(512), 8192, st, a, wbuf); break;
                        case 3: hipLaunchKernelGGL((k_big<2, 4>), dim3(NB), dim3(512), 8192, st, a, wbuf); break;
                        case 4: hipLaunchKernelGGL((k_big<2, 5>), dim3(NB), dim3(512), 8192, st, a, wbuf); break;
                        default: hipLaunchKernelGGL((k_big<2, 6>), dim3(NB), dim3(512), 8192, st, a, wbuf); break;
                    }
                    break;
                case 11:
                    switch (i % 3) {
                        case 0: hipLaunchKernelGGL((k_big<2, 1>), dim3(128), dim3(512), 8192, st, a, wbuf); break;
                        case 1: hipLaunchKernelGGL((k_big<2, 2>), dim3(192), dim3(512), 40000, st, a, wbuf); break;
                        default: hipLaunchKernelGGL((k_big<2, 3>), dim3(256), dim3(512), 20000, st, a, wbuf); break;
                    }
                    break;
            }
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms = 0.f; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-9s %4d workgroups: %6.2f us per dependent kernel\n", names[v], NB, ms * 1e3 / (REP * CH));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
