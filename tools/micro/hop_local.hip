// Same-XCD hand-off through the shared L2 (sc0 = workgroup.. XCD-coherent accesses) against the agent-scope (sc1) hand-off the
// persistent kernels use: ping-pong between workgroup 0 and workgroup P on fresh cache lines, with and without 200 workgroups of
// agent-scope row traffic.  Workgroups are dealt round-robin to the 8 XCDs (checked: the XCC ids are printed); every spin is bounded.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/hop_local.hip -o gpurun_in/hop_local && gpurun_in/hop_local
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
template <int AUX>
__global__ void k_pp(unsigned* slots, int partner, int rounds, unsigned long long* cycles, unsigned* xcc, const float* noise, float* sink, unsigned* fail) {
    const int wg = blockIdx.x;
    if (wg >= 32) {
        const __amdgpu_buffer_rsrc_t rn = rsrc(noise), rs = rsrc(slots);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < 100000; ++it) {
            for (int j = 0; j < 6; ++j)
                acc += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rn, (int)(((wg * 7 + it) % 64) * 49152 + (threadIdx.x + 512 * j) * 16), 0, 16));
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
    if ((wg != 0 && wg != partner) || threadIdx.x != 0) return;
    const __amdgpu_buffer_rsrc_t r = rsrc(slots);
    const unsigned long long t0 = __builtin_readcyclecounter();
    bool ok = true;
    for (int i = 1; i <= rounds && ok; ++i) {
        const int mine = 4096 + i * 4096 + (wg == 0 ? 0 : 2048), theirs = 4096 + i * 4096 + (wg == 0 ? 2048 : 0);
        unsigned x = 0;
        if (wg == 0) __builtin_amdgcn_raw_buffer_store_b32((unsigned)i, r, mine, 0, AUX);
        for (int s = 0; s < 2000000; ++s) {
            x = __builtin_amdgcn_raw_buffer_load_b32(r, theirs, 0, AUX);
            asm volatile("" ::: "memory");
            if (x == (unsigned)i) break;
        }
        if (x != (unsigned)i) { ok = false; fail[0] = (unsigned)i; }
        if (wg != 0) __builtin_amdgcn_raw_buffer_store_b32((unsigned)i, r, mine, 0, AUX);
    }
    if (wg == 0) {
        cycles[0] = __builtin_readcyclecounter() - t0;
        __builtin_amdgcn_raw_buffer_store_b32(1u, r, 512, 0, 16);
    }
}
int main() {
    unsigned *slots, *xcc, *fail;
    unsigned long long* cyc;
    float *noise, *sink;
    const int rounds = 3000;
    if (hipMalloc(&slots, 64 << 20) != hipSuccess || hipMalloc(&xcc, 256) != hipSuccess || hipMalloc(&cyc, 8) != hipSuccess || hipMalloc(&fail, 4) != hipSuccess) return 1;
    if (hipMalloc(&noise, 64 * 49152) != hipSuccess || hipMalloc(&sink, 4096) != hipSuccess) return 1;
    (void)hipMemset(noise, 0, 64 * 49152);
    for (int cfg = 0; cfg < 6; ++cfg) {
        const int partner = cfg < 2 ? 1 : 8, sc0 = cfg >= 4, nn = (cfg & 1) ? 200 : 0;
        (void)hipMemset(slots, 0, 64 << 20);
        (void)hipMemset(fail, 0, 4);
        if (sc0) k_pp<1><<<32 + nn, 512>>>(slots, partner, rounds, cyc, xcc, noise, sink, fail);
        else k_pp<16><<<32 + nn, 512>>>(slots, partner, rounds, cyc, xcc, noise, sink, fail);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        unsigned hx[64], hf; unsigned long long hc;
        (void)hipMemcpy(hx, xcc, 256, hipMemcpyDeviceToHost); (void)hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
        printf("workgroup 0 (XCC %u) <-> workgroup %d (XCC %u), %s accesses, %3d noise workgroups: %.0f shader cycles per one-way hand-off%s\n", hx[0], partner, hx[partner],
               sc0 ? "sc0 (XCD-local L2)" : "sc1 (agent scope)  ", nn, (double)hc / rounds / 2, hf ? "  ** a spin gave up **" : "");
    }
    return 0;
}
