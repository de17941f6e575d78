// Practical fp32 MFMA peak on this box (wall clock, all CUs): v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32 from registers.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma_peak.hip -o tools/ubench_mfma_peak
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int K>
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
    f32x16 acc[4];
    for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    float a = (float)threadIdx.x, b = 1.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
    }
    float r = 0.f;
    for (int n = 0; n < 4; ++n) r += acc[n][0];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
    f32x4 acc[8];
    for (int n = 0; n < 8; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = (float)threadIdx.x, b = 1.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int n = 0; n < 8; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[n], 0, 0, 0);
    }
    float r = 0.f;
    for (int n = 0; n < 8; ++n) r += acc[n][0];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    float* out; hipMalloc(&out, 2048 * 256 * 4);
    const int iters = 20000;
    for (int grid : {256, 512, 1024}) {
        k32<0><<<grid, 256>>>(out, 100); hipDeviceSynchronize();
        double t0 = now(); k32<0><<<grid, 256>>>(out, iters); hipDeviceSynchronize(); double t = now() - t0;
        printf("32x32x2 f32, grid %4d: %.1f TFLOP/s (%.1f ms)\n", grid, (double)grid * 4 * iters * 32.0 * 4096 / t / 1e12, t * 1e3);
        k16<<<grid, 256>>>(out, 100); hipDeviceSynchronize();
        t0 = now(); k16<<<grid, 256>>>(out, iters); hipDeviceSynchronize(); t = now() - t0;
        printf("16x16x4 f32, grid %4d: %.1f TFLOP/s (%.1f ms)\n", grid, (double)grid * 4 * iters * 64.0 * 2048 / t / 1e12, t * 1e3);
    }
    return 0;
}
