// Hand-off latency between two workgroups on one MI355X: what a dependent CU-to-CU hop of the persistent decoder
// kernels costs at least.  Workgroup 0 and workgroup P play ping-pong through two words in global memory with the
// same instructions as the kernels (write-through sc1 stores, sc1 loads polled until the word changes).
// Workgroups are dealt round-robin to the 8 XCDs: P = 8 shares workgroup 0's XCD (and L2), P = 1..7 does not.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/hop_latency.hip -o gpurun_out/hop_latency && gpurun_out/hop_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
// noise: workgroups >= 32 re-read a 48 KB row with agent-scope loads (what the row all-gathers of the kernels do while they
// poll) until workgroup 0 is done; `noise_sc1` = 0 makes the same loads ordinary (L2-cached) ones
__global__ void k_pingpong(unsigned* slots, int partner, int rounds, unsigned long long* cycles, unsigned* xcc, const float* noise, int noise_sc1,
                           float* sink, int stride) {
    const int wg = blockIdx.x;
    if (wg >= 32) {
        const __amdgpu_buffer_rsrc_t rn = rsrc(noise);
        const __amdgpu_buffer_rsrc_t rs = rsrc(slots);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < 100000; ++it) {
            for (int j = 0; j < 6; ++j) {
                const int off = (int)(((wg * 7 + it) % 64) * 49152 + (threadIdx.x + 512 * j) * 16);
                acc += noise_sc1 ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rn, off, 0, 16))
                                 : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rn, off, 0, 0));
            }
            if (__builtin_amdgcn_raw_buffer_load_b32(rs, 512, 0, 16) != 0u) break;
            asm volatile("" ::: "memory");
        }
        if (acc[0] == 1.2345f) sink[threadIdx.x] = acc[1];
        return;
    }
    if (threadIdx.x == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[wg] = id & 0xf;
    }
    if (wg != 0 && wg != partner) return;
    if (threadIdx.x != 0) return;
    const __amdgpu_buffer_rsrc_t r = rsrc(slots);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 1; i <= rounds; ++i) {
        if (wg == 0) {
            __builtin_amdgcn_raw_buffer_store_b32((unsigned)i, r, 4096 + i * stride, 0, 16);
            unsigned x;
            do { x = __builtin_amdgcn_raw_buffer_load_b32(r, 4096 + i * stride + 2048, 0, 16); asm volatile("" ::: "memory"); } while (x != (unsigned)i);
        } else {
            unsigned x;
            do { x = __builtin_amdgcn_raw_buffer_load_b32(r, 4096 + i * stride, 0, 16); asm volatile("" ::: "memory"); } while (x != (unsigned)i);
            __builtin_amdgcn_raw_buffer_store_b32((unsigned)i, r, 4096 + i * stride + 2048, 0, 16);
        }
    }
    if (wg == 0) {
        cycles[0] = __builtin_readcyclecounter() - t0;
        __builtin_amdgcn_raw_buffer_store_b32(1u, r, 512, 0, 16);      // stops the noise
    }
}

int main() {
    unsigned *slots, *xcc;
    unsigned long long* cyc;
    if (hipMalloc(&slots, 64 << 20) != hipSuccess || hipMalloc(&xcc, 64 * 4) != hipSuccess || hipMalloc(&cyc, 8) != hipSuccess) return 1;
    float *noise, *sink;
    if (hipMalloc(&noise, 64 * 49152) != hipSuccess || hipMalloc(&sink, 4096) != hipSuccess) return 1;
    hipMemset(noise, 0, 64 * 49152);
    const int rounds = 5000;
    for (int cfg = 0; cfg < 10; ++cfg) {
        const int stride = cfg < 7 ? 0 : (cfg == 7 ? 4096 : cfg == 8 ? 8192 : 12288);
        const int partner = 1, nwg_noise = (cfg == 0 || cfg >= 7) ? 0 : (cfg <= 3 ? (cfg == 1 ? 32 : cfg == 2 ? 96 : 200) : (cfg == 4 ? 32 : cfg == 5 ? 96 : 200)), sc1 = cfg <= 3;
        hipMemset(slots, 0, 64 << 20);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k_pingpong<<<32 + nwg_noise, 512>>>(slots, partner, rounds, cyc, xcc, noise, sc1, sink, stride);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned hx[64]; unsigned long long hc;
        hipMemcpy(hx, xcc, 64 * 4, hipMemcpyDeviceToHost); hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        printf("workgroup 0 (XCC %u) <-> workgroup %2d (XCC %u), %3d noise workgroups (%s loads), fresh-line stride %5d B: %.0f shader cycles per one-way hand-off\n", hx[0], partner,
               hx[partner], nwg_noise, sc1 ? "agent-scope" : "cached", stride, (double)hc / rounds / 2);
    }
    return 0;
}
