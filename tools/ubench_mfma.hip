// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 with NACC independent accumulators, one wave per SIMD,
// optionally with LDS operand reads threaded through (the conv kernels' inner loop).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench_mfma.hip -o tools/ubench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NACC, int MODE>   // MODE 0: registers only; 1: + (1 + NACC) ds_read_b32 per step, double-buffered
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    __shared__ float lds[2][80][112];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2 * 80 * 112; i += 256) (&lds[0][0][0])[i] = (float)(i & 7);
    __syncthreads();
    f32x4 acc[NACC];
#pragma unroll
    for (int n = 0; n < NACC; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = (float)lane, b = 1.0f;
    const float* ap = &lds[0][lane >> 4][lane & 15];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int s = 0; s < 20; ++s) {
#pragma unroll
                for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[n], 0, 0, 0);
            }
        } else {
            float av[2], bv[2][NACC];
            av[0] = ap[0];
#pragma unroll
            for (int n = 0; n < NACC; ++n) bv[0][n] = ap[16 * n + 1];
#pragma unroll
            for (int s = 0; s < 20; ++s) {
                if (s + 1 < 20) {
                    av[(s + 1) & 1] = ap[4 * (s + 1) * 112];
#pragma unroll
                    for (int n = 0; n < NACC; ++n) bv[(s + 1) & 1][n] = ap[4 * ((s + 1) & 3) * 112 + 16 * n + ((s + 1) >> 2)];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s & 1], bv[s & 1][n], acc[n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0.f;
#pragma unroll
    for (int n = 0; n < NACC; ++n) r += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    out[blockIdx.x * 256 + tid] = r;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC, int MODE>
static int run(const char* name, float* out, unsigned long long* cyc, int grid) {
    const int iters = 200;
    k<NACC, MODE><<<grid, 256>>>(out, cyc, iters);
    CK(hipDeviceSynchronize());
    k<NACC, MODE><<<grid, 256>>>(out, cyc, iters);
    CK(hipDeviceSynchronize());
    unsigned long long c;
    CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-52s grid %4d: %6.1f cycles per MFMA\n", name, grid, (double)c / (iters * 20.0 * NACC));
    return 0;
}

int main() {
    float* out; unsigned long long* cyc;
    CK(hipMalloc(&out, 1024 * 256 * 4)); CK(hipMalloc(&cyc, 8));
    run<5, 0>("5 accumulators, registers only", out, cyc, 256);
    run<5, 0>("5 accumulators, registers only", out, cyc, 512);
    run<5, 1>("5 accumulators, 6 ds_read_b32 per 5 MFMAs", out, cyc, 256);
    run<5, 1>("5 accumulators, 6 ds_read_b32 per 5 MFMAs", out, cyc, 512);
    run<4, 1>("4 accumulators, 5 ds_read_b32 per 4 MFMAs", out, cyc, 256);
    run<6, 1>("6 accumulators, 7 ds_read_b32 per 6 MFMAs", out, cyc, 256);
    run<8, 0>("8 accumulators, registers only", out, cyc, 256);
    run<8, 1>("8 accumulators, 9 ds_read_b32 per 8 MFMAs", out, cyc, 256);
    return 0;
}
