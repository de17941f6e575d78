// Host launch cost on this box: eager <<<>>> launches vs hipGraphLaunch of the same chain (kernel body ~ T us of spinning).
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench_launch.hip -o tools/ubench_launch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
struct Args { float* p; long pad[24]; int spin; };
__global__ __launch_bounds__(256) void k_spin(Args a) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (unsigned long long)a.spin) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) a.p[0] += 1.f;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 1600, spin = argc > 2 ? atoi(argv[2]) : 0, grid = argc > 3 ? atoi(argv[3]) : 256;
    const int msevery = argc > 4 ? atoi(argv[4]) : 0;    // a hipMemsetAsync node after every `msevery` kernels
    float* scratch; hipMalloc(&scratch, 1 << 20);
    float* d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
    hipStream_t s; hipStreamCreate(&s);
    Args a; a.p = d; a.spin = spin;
    for (int i = 0; i < 100; ++i) k_spin<<<grid, 256, 0, s>>>(a);
    hipStreamSynchronize(s);
    for (int rep = 0; rep < 3; ++rep) {
        const double t0 = now();
        for (int i = 0; i < N; ++i) k_spin<<<grid, 256, 0, s>>>(a);
        const double t1 = now();
        hipStreamSynchronize(s);
        const double t2 = now();
        printf("eager  N=%d spin=%d: host enqueue %.2f us/launch, total %.2f us/launch\n", N, spin, (t1 - t0) * 1e6 / N, (t2 - t0) * 1e6 / N);
    }
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < N; ++i) {
        k_spin<<<grid, 256, 0, s>>>(a);
        if (msevery && i % msevery == msevery - 1) hipMemsetAsync(scratch, 0, 4096, s);
    }
    hipStreamEndCapture(s, &g);
    const double ti0 = now();
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    printf("instantiate %.2f ms\n", (now() - ti0) * 1e3);
    for (int rep = 0; rep < 4; ++rep) {
        const double t0 = now();
        hipGraphLaunch(ge, s);
        const double t1 = now();
        hipStreamSynchronize(s);
        const double t2 = now();
        printf("graph  N=%d spin=%d: hipGraphLaunch returns after %.2f us/node, total %.2f us/node\n", N, spin, (t1 - t0) * 1e6 / N, (t2 - t0) * 1e6 / N);
    }
    {   // back-to-back replays of the same executable graph without a host sync in between
        const double t0 = now();
        for (int rep = 0; rep < 6; ++rep) hipGraphLaunch(ge, s);
        const double t1 = now();
        hipStreamSynchronize(s);
        const double t2 = now();
        printf("graph x6 back-to-back: launches return after %.2f ms, total %.2f us/node\n", (t1 - t0) * 1e3, (t2 - t0) * 1e6 / (6.0 * N));
    }
    return 0;
}
