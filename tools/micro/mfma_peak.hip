// Sustained MFMA rate of the whole chip for the instructions the kernels of this repo are priced against:
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o gpurun_in/mfma_peak && gpurun_in/mfma_peak
// Every wave issues independent back-to-back MFMAs from registers (no memory traffic), W waves per SIMD; the printed rate is
// what a perfectly fed kernel could reach on this box at its sustained clock.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void k_peak(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    if (MODE == 0) {            // v_mfma_f32_32x32x2_f32
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        float a = lane * 1e-3f, b = 1.0f + lane * 1e-4f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 4; ++i) s += acc[i][0];
        if (s == 12345.f) out[0] = s;
    } else if (MODE == 1) {     // v_mfma_f32_32x32x16_bf16
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        bf16x8 a, b;
        for (int r = 0; r < 8; ++r) { a[r] = (__bf16)(lane * 1e-3f); b[r] = (__bf16)(1.f + r); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 4; ++i) s += acc[i][0];
        if (s == 12345.f) out[0] = s;
    } else {                    // v_mfma_f32_16x16x32_bf16
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        bf16x8 a, b;
        for (int r = 0; r < 8; ++r) { a[r] = (__bf16)(lane * 1e-3f); b[r] = (__bf16)(1.f + r); }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][0];
        if (s == 12345.f) out[0] = s;
    }
}

template <int MODE>
static void run(const char* name, double flop_per_mfma, int wgs_per_cu) {
    float* out;
    hipMalloc(&out, 4);
    const int iters = 20000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_peak<MODE><<<grid, 256>>>(out, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_peak<MODE><<<grid, 256>>>(out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)grid * 4 * iters * 16;
    printf("%-28s %d wave(s)/SIMD: %7.2f ms  %8.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", name, wgs_per_cu, ms,
           mfma * flop_per_mfma / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 16.0 * wgs_per_cu));
    hipFree(out);
}
int main() {
    run<0>("v_mfma_f32_32x32x2_f32", 4096.0, 1);
    run<0>("v_mfma_f32_32x32x2_f32", 4096.0, 2);
    run<1>("v_mfma_f32_32x32x16_bf16", 32768.0, 1);
    run<1>("v_mfma_f32_32x32x16_bf16", 32768.0, 2);
    run<2>("v_mfma_f32_16x16x32_bf16", 16384.0, 1);
    run<2>("v_mfma_f32_16x16x32_bf16", 16384.0, 2);
    return 0;
}
