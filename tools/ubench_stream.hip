// Micro-benchmark: how fast can 256 workgroups stream a 64 MiB weight set per launch?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NT, int PER>   // NT threads, PER float4 per thread per outer iteration
__global__ __launch_bounds__(NT) void k_read(const float4* __restrict__ p, size_t n4_per_wg, float* out, int mode) {
    const float4* base = mode == 0 ? p + (size_t)blockIdx.x * n4_per_wg : p;   // mode0: tile-major; mode1: interleaved
    const size_t stride = mode == 0 ? NT : (size_t)gridDim.x * NT;
    const size_t start = mode == 0 ? threadIdx.x : (size_t)blockIdx.x * NT + threadIdx.x;
    float acc = 0.f;
    const int iters = (int)(n4_per_wg / (NT * PER));
    for (int it = 0; it < iters; ++it) {
        float4 v[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) v[i] = base[start + (size_t)(it * PER + i) * stride];
#pragma unroll
        for (int i = 0; i < PER; ++i) acc += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int NT, int PER>
__global__ __launch_bounds__(NT) void k_read_mfma(const float4* __restrict__ p, size_t n4_per_wg, float* out, int mode) {
    const float4* base = mode == 0 ? p + (size_t)blockIdx.x * n4_per_wg : p;
    const size_t stride = mode == 0 ? NT : (size_t)gridDim.x * NT;
    const size_t start = mode == 0 ? threadIdx.x : (size_t)blockIdx.x * NT + threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    const int iters = (int)(n4_per_wg / (NT * PER));
    for (int it = 0; it < iters; ++it) {
        float4 v[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) v[i] = base[start + (size_t)(it * PER + i) * stride];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v[i].x, 1.0f, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v[i].y, 1.0f, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v[i].z, 1.0f, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v[i].w, 1.0f, acc, 0, 0, 0);
        }
    }
    if (acc[0] == 123.456f) out[0] = acc[0];
}

template <typename F>
float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / reps;
}

int main() {
    const size_t bytes = 64ull << 20;
    const size_t n4 = bytes / 16;
    float4* d; float* out;
    CK(hipMalloc(&d, bytes)); CK(hipMalloc(&out, 4));
    CK(hipMemset(d, 0, bytes));
    const int G = 256;
    const size_t per = n4 / G;
    for (int mode = 0; mode < 2; ++mode) {
        printf("mode %d (%s)\n", mode, mode == 0 ? "tile-major" : "interleaved");
        float us;
        us = timeit([&] { k_read<256, 8><<<G, 256>>>(d, per, out, mode); }, 200);  printf("  read  NT=256  PER=8 : %7.2f us  %6.0f GB/s\n", us, bytes / us / 1e3);
        us = timeit([&] { k_read<256, 16><<<G, 256>>>(d, per, out, mode); }, 200); printf("  read  NT=256  PER=16: %7.2f us  %6.0f GB/s\n", us, bytes / us / 1e3);
        us = timeit([&] { k_read<512, 8><<<G, 512>>>(d, per, out, mode); }, 200);  printf("  read  NT=512  PER=8 : %7.2f us  %6.0f GB/s\n", us, bytes / us / 1e3);
        us = timeit([&] { k_read<1024, 4><<<G, 1024>>>(d, per, out, mode); }, 200); printf("  read  NT=1024 PER=4 : %7.2f us  %6.0f GB/s\n", us, bytes / us / 1e3);
        us = timeit([&] { k_read<1024, 16><<<G, 1024>>>(d, per, out, mode); }, 200); printf("  read  NT=1024 PER=16: %7.2f us  %6.0f GB/s\n", us, bytes / us / 1e3);
        us = timeit([&] { k_read_mfma<1024, 16><<<G, 1024>>>(d, per, out, mode); }, 200); printf("  mfma  NT=1024 PER=16: %7.2f us  %6.0f GB/s\n", us, bytes / us / 1e3);
        us = timeit([&] { k_read_mfma<256, 16><<<G, 256>>>(d, per, out, mode); }, 200); printf("  mfma  NT=256  PER=16: %7.2f us  %6.0f GB/s\n", us, bytes / us / 1e3);
        us = timeit([&] { k_read<256, 8><<<G * 4, 256>>>(d, per / 4, out, mode); }, 200); printf("  read  4WG/CU NT=256 PER=8: %7.2f us  %6.0f GB/s\n", us, bytes / us / 1e3);
        us = timeit([&] { k_read<256, 8><<<G * 8, 256>>>(d, per / 8, out, mode); }, 200); printf("  read  8WG/CU NT=256 PER=8: %7.2f us  %6.0f GB/s\n", us, bytes / us / 1e3);
    }
    // empty kernel launch cost
    float us = timeit([&] { k_read<256, 8><<<G, 256>>>(d, 0, out, 0); }, 500); printf("empty launch: %.2f us\n", us);
    return 0;
}
