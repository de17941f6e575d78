// Micro-benchmark for the round-2 design question: what does ONE phase of a fully persistent decoder cost?
// 256 workgroups (one per CU) stay resident; per phase every workgroup
//   (a) reads the broadcast state vector x (B x 2560 floats, written by all workgroups in the previous phase) with
//       loads that bypass the non-coherent L2,
//   (b) multiplies it with register-resident weights (stand-in: 64 MFMAs per wave),
//   (c) publishes its 4 x B new state values (write-through) and its epoch flag,
//   (d) waits until all 256 flags carry the epoch (the grid-wide hand-off).
// Reported: microseconds per phase for several variants of (a) and (d).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench_persist_step.hip -o tools/ubench_persist_step
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define NWG 256
#define XW 2560

__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// MODE bit0: read x (a); bit1: MFMA work (b); bit2: flags barrier (d).  THREADS threads per workgroup.
template <int THREADS, int MODE>
__global__ __launch_bounds__(THREADS) void k_phase(float* xs /* [2][16][XW] */, unsigned* flags /* [NWG] */, unsigned* err,
                                                   int B, int steps, float* sink, unsigned long long* cyc) {
    __shared__ float xl[16 * XW / 4];      // a quarter of x is enough to keep LDS small; reads cover all of it
    __shared__ int ok;
    const int tid = threadIdx.x, w = blockIdx.x, lane = tid & 63;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float wreg[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) wreg[i] = 0.001f * (tid + i);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int st = 1; st <= steps; ++st) {
        const float* xin = xs + (size_t)((st - 1) & 1) * 16 * XW;
        float* xout = xs + (size_t)(st & 1) * 16 * XW;
        if (MODE & 1) {
            // (a) B x 2560 floats, every workgroup reads all of it; float4 loads that bypass L2
            float s = 0.f;
            for (int i = tid; i < B * XW / 4; i += THREADS) {
                const float* p = xin + 4 * i;
                const float a0 = ld_sc1(p), a1 = ld_sc1(p + 1), a2 = ld_sc1(p + 2), a3 = ld_sc1(p + 3);
                xl[i & (16 * XW / 16 - 1)] = a0 + a1;
                s += a2 + a3;
            }
            acc[0] += s;
            __syncthreads();
        }
        if (MODE & 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[i], xl[(lane + 64 * i) & 1023], acc, 0, 0, 0);
        }
        // (c) publish 4 x B values + flag
        if (tid < 4 * B) st_sc1(xout + (size_t)(tid >> 2) * XW + 4 * w + (tid & 3), acc[0] * 1e-9f + (float)st);
        if (MODE & 4) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flags + w, (unsigned)st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (d) everybody polls one flag (tid < NWG), bounded
            for (unsigned spins = 0;; ++spins) {
                bool good = true;
                if (tid < NWG) good = __hip_atomic_load(flags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)st;
                if (__syncthreads_and(good)) break;
                if (spins > 2000000u) { if (tid == 0) *err = 1u; return; }
            }
        } else {
            __syncthreads();
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) sink[w] = acc[0] + acc[1];
    if (tid == 0 && w == 0) cyc[0] = t1 - t0;
}

template <int THREADS, int MODE>
static int run(const char* name, float* xs, unsigned* flags, unsigned* err, float* sink, unsigned long long* cyc, int B) {
    const int steps = 2000;
    CK(hipMemset(flags, 0, NWG * 4)); CK(hipMemset(err, 0, 4));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    k_phase<THREADS, MODE><<<NWG, THREADS>>>(xs, flags, err, B, steps, sink, cyc);
    hipEventRecord(b);
    CK(hipEventSynchronize(b));
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned e; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
    printf("%-62s %6.2f us/phase%s\n", name, ms * 1000.f / steps, e ? "  (TIMEOUT)" : "");
    return 0;
}

int main() {
    float *xs, *sink; unsigned *flags, *err; unsigned long long* cyc;
    CK(hipMalloc(&xs, 2 * 16 * XW * 4)); CK(hipMemset(xs, 0, 2 * 16 * XW * 4));
    CK(hipMalloc(&sink, NWG * 4)); CK(hipMalloc(&flags, NWG * 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&cyc, 8));
    run<1024, 4>("1024 thr: publish + 256-flag hand-off only", xs, flags, err, sink, cyc, 6);
    run<256, 4>("256 thr: publish + 256-flag hand-off only", xs, flags, err, sink, cyc, 6);
    run<1024, 5>("1024 thr: + read x (B=6: 61 KB per workgroup, sc1 loads)", xs, flags, err, sink, cyc, 6);
    run<1024, 7>("1024 thr: + read x + 64 MFMAs per wave", xs, flags, err, sink, cyc, 6);
    run<512, 7>("512 thr: + read x + 64 MFMAs per wave", xs, flags, err, sink, cyc, 6);
    run<1024, 7>("1024 thr: same, B=16 (164 KB per workgroup)", xs, flags, err, sink, cyc, 16);
    run<1024, 3>("1024 thr: read x + MFMA, NO hand-off (upper bound on (a)+(b))", xs, flags, err, sink, cyc, 6);
    return 0;
}
