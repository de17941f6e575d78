// Do two 256-thread workgroups of a ~208-VGPR kernel share a CU?  Every workgroup sleeps ~5 us; a grid of 512 (or 304)
// workgroups takes ~5 us when two fit a CU and ~10 us when the second one has to wait for a free CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int V>
__global__ __launch_bounds__(256) void k_sleep(int* out, int n) {
    if (V == 200) asm volatile("v_mov_b32 v200, 0" ::: "v200");
    if (V == 120) asm volatile("v_mov_b32 v120, 0" ::: "v120");
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
    if (out && threadIdx.x == 0 && n < 0) out[blockIdx.x] = 1;
}
template <int V>
static float run(int grid, int n) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) k_sleep<V><<<grid, 256>>>(nullptr, n);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) k_sleep<V><<<grid, 256>>>(nullptr, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / 20;
}
int main() {
    const int grids[] = {256, 304, 512, 1024};
    for (int g : grids)
        printf("grid %4d: VGPR<=64 %.1f us   VGPR~120 %.1f us   VGPR~200 %.1f us\n", g, run<0>(g, 2), run<120>(g, 2), run<200>(g, 2));
    return 0;
}
