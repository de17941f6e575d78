// Micro-benchmark for the data-gradient GEMV tiles of the decoder backward step: how should the 67 MB weight
// stream of one reverse step be cut into workgroups?  Build (from the repo root):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench_gemv_tiles.hip tacotron2-vae_amd/csrc/t2v_runtime.hip \
//         -Itacotron2-vae_amd/csrc -o tools/ubench_gemv_tiles
#include "../tacotron2-vae_amd/csrc/decoder_bwd.hip"
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// uniform cut: 256 tiles x KSPLIT workgroups of THREADS threads, ROUNDS rounds of 8 (W, k) pairs per wave
template <int THREADS, int ROUNDS, int LDSPAD, int TILE_MAJOR = 0, int SWZ = 0>
__global__ __launch_bounds__(THREADS) void k_tiles(const float4* packD, const float4* packA, const float* kvd,
                                                   const float* kva, float* out, int ksplit, int flip, int B) {
    constexpr int WAVES = THREADS / 64;
    __shared__ f32x4 red[WAVES][64];
    __shared__ float pad[LDSPAD / 4 + 1];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int w = blockIdx.x / ksplit;
    const int ks = SWZ ? (blockIdx.x % ksplit + (w >> 1)) % ksplit : blockIdx.x % ksplit;
    const bool dec = w < T2V_XW / 16;
    const int wt = dec ? w : w - T2V_XW / 16;
    const int ntile = dec ? T2V_XW / 16 : T2V_KATT / 16;
    const float* kv = dec ? kvd : kva;
    const int b = lane & 15, g = lane >> 4;
    const bool bvalid = b < B;
    const float4* p = (dec ? packD : packA) + (TILE_MAJOR ? (size_t)wt * 256 * 64 : (size_t)wt * 64) + lane;
    const size_t kstride = TILE_MAJOR ? 64 : (size_t)ntile * 64;
    const float* xrow = kv + (size_t)(bvalid ? b : 0) * T2V_G + 4 * g;
    const int kb0 = ks * (256 / ksplit) + 8 * ROUNDS * wave;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float4 wv[2][8], xv[2][8];
#define LOAD_ROUND(H)                                                                          \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                            \
        const int kb = flip ? kb0 + 8 * ROUNDS - 1 - (8 * (H) + i) : kb0 + 8 * (H) + i;        \
        wv[(H) & 1][i] = p[(size_t)kb * kstride];                                           \
        xv[(H) & 1][i] = *(const float4*)(xrow + 16 * kb);                                     \
    }
    LOAD_ROUND(0)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < ROUNDS; ++h) {
        if (h + 1 < ROUNDS) { LOAD_ROUND(h + 1) }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) { MFMA4(acc, wv[h & 1][i], xv[h & 1][i]); }
        __builtin_amdgcn_sched_barrier(0);
    }
    red[wave][lane] = acc;
    if (LDSPAD && tid == 0) pad[LDSPAD / 4] = acc[0];
    __syncthreads();
    if (wave == 0 && bvalid) {
        f32x4 s = red[0][lane];
#pragma unroll
        for (int i = 1; i < WAVES; ++i) s += red[i][lane];
        float* y = out + ((size_t)ks * 16 + b) * 4096 + 16 * w + 4 * g;
        *(float4*)y = make_float4(s[0], s[1], s[2], s[3]);
    }
}

template <typename F>
static float timeit(F f, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 10; ++i) f(i);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f(i);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / reps;
}

int main() {
    const int B = 6;
    const size_t nD = (size_t)(T2V_XW / 16) * 256 * 64, nA = (size_t)(T2V_KATT / 16) * 256 * 64;
    float4 *pD, *pA; float *kvd, *kva, *out;
    CK(hipMalloc(&pD, nD * 16)); CK(hipMalloc(&pA, nA * 16));
    CK(hipMemset(pD, 0, nD * 16)); CK(hipMemset(pA, 0, nA * 16));
    CK(hipMalloc(&kvd, 16 * T2V_G * 4)); CK(hipMalloc(&kva, 16 * T2V_G * 4));
    CK(hipMemset(kvd, 0, 16 * T2V_G * 4)); CK(hipMemset(kva, 0, 16 * T2V_G * 4));
    CK(hipMalloc(&out, 8 * 16 * 4096 * 4));
    const double mb = (nD + nA) * 16 / 1e6;
    const int reps = 4000;
    for (int i = 0; i < 30000; ++i) k_tiles<1024, 2, 0><<<256, 1024>>>(pD, pA, kvd, kva, out, 1, i & 1, B);   // clocks up
    CK(hipDeviceSynchronize());
#define RUN(NAME, THREADS, ROUNDS, LDSPAD, KSPLIT) RUN3(NAME, THREADS, ROUNDS, LDSPAD, KSPLIT, 0, 1, 0)
#define RUN2(NAME, THREADS, ROUNDS, LDSPAD, KSPLIT, TM, FLIP) RUN3(NAME, THREADS, ROUNDS, LDSPAD, KSPLIT, TM, FLIP, 0)
#define RUN3(NAME, THREADS, ROUNDS, LDSPAD, KSPLIT, TM, FLIP, SWZ)                                                            \
    {                                                                                                         \
        static_assert((THREADS / 64) * 8 * ROUNDS * KSPLIT == 256, "cut must cover 256 k-blocks");          \
        float us = timeit([&](int i) { k_tiles<THREADS, ROUNDS, LDSPAD, TM, SWZ><<<256 * KSPLIT, THREADS>>>(pD, pA, kvd, kva, out, KSPLIT, FLIP ? (i & 1) : 0, B); }, reps); \
        printf("%-44s %7.2f us  %6.2f TB/s\n", NAME, us, mb / us);                          \
    }
    RUN("1024 thr, 2 rounds, ksplit 1 (old kernel)", 1024, 2, 0, 1)
    RUN("512 thr, 4 rounds, ksplit 1", 512, 4, 0, 1)
    RUN("512 thr, 2 rounds, ksplit 2", 512, 2, 0, 2)
    RUN("256 thr, 8 rounds, ksplit 1", 256, 8, 0, 1)
    RUN("256 thr, 4 rounds, ksplit 2", 256, 4, 0, 2)
    RUN("256 thr, 2 rounds, ksplit 4", 256, 2, 0, 4)
    RUN("256 thr, 1 round,  ksplit 8", 256, 1, 0, 8)
    RUN3("256 thr, 4 rounds, ksplit 2, swizzled", 256, 4, 0, 2, 0, 1, 1)
    RUN3("256 thr, 2 rounds, ksplit 4, swizzled", 256, 2, 0, 4, 0, 1, 1)
    RUN3("256 thr, 1 round,  ksplit 8, swizzled", 256, 1, 0, 8, 0, 1, 1)
    RUN2("256 thr, 8 rounds, ksplit 1, no flip", 256, 8, 0, 1, 0, 0)
    RUN2("256 thr, 2 rounds, ksplit 4, no flip", 256, 2, 0, 4, 0, 0)
    RUN2("256 thr, 8 rounds, ksplit 1, tile-major", 256, 8, 0, 1, 1, 1)
    RUN2("256 thr, 4 rounds, ksplit 2, tile-major", 256, 4, 0, 2, 1, 1)
    RUN2("256 thr, 2 rounds, ksplit 4, tile-major", 256, 2, 0, 4, 1, 1)
    RUN2("256 thr, 1 round,  ksplit 8, tile-major", 256, 1, 0, 8, 1, 1)
    RUN2("256 thr, 2 rounds, ksplit 4, tile-major, no flip", 256, 2, 0, 4, 1, 0)
    RUN2("1024 thr, 2 rounds, ksplit 1, tile-major", 1024, 2, 0, 1, 1, 1)
    // the production kernel of the 2-launch schedule
    {
        float us = timeit([&](int i) {
            LstmBwdArgs l;
            l.packBD = pD; l.packBA = pA; l.dgd_t = kvd; l.dga_n = kva; l.YD = out; l.YA = out + 16 * T2V_XW;
            l.B = B; l.flip = i & 1;
            k_lstm_bwd<<<T2V_NWG, 1024>>>(l);
        }, reps);
        float us2 = timeit([&](int i) {
            LstmBwdArgs l;
            l.packBD = pD; l.packBA = pA; l.dgd_t = kvd; l.dga_n = kva; l.YD = out; l.YA = out + 16 * T2V_XW;
            l.B = B; l.flip = i & 1;
            k_lstm_bwd256<<<T2V_NWG, 256>>>(l);
        }, reps);
        printf("%-44s %7.2f us  %6.2f TB/s\n", "k_lstm_bwd256 (production)", us2, mb / us2);
        printf("%-44s %7.2f us  %6.2f TB/s\n", "k_lstm_bwd (1024 thr, round-1 first half)", us, mb / us);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
