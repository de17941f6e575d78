// Library-level plumbing: version string, error recording.
#include "t2v_common.h"
#include "t2v_kernels.h"
#include <string.h>

static thread_local char g_err[256] = "";

int t2v_check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        strncpy(g_err, hipGetErrorString(e), sizeof(g_err) - 1);
        return T2V_ERR_LAUNCH;
    }
    return T2V_OK;
}
extern "C" const char* t2v_version(void) { return "t2vae-hip 0.1 (gfx950)"; }
extern "C" const char* t2v_last_error(void) { return g_err; }

unsigned long long* g_t2v_prof = nullptr;
// Tracing aid: when set, k_attn_fwd / k_attn_bwd write s_memtime stamps at their phase boundaries
// (fwd -> slots 0..7, bwd -> slots 16..23).  NULL disables it.
extern "C" void t2v_set_phase_profile(unsigned long long* dev_buf32) { g_t2v_prof = dev_buf32; }

// Per-step parameters in device memory (optional): lets a captured HIP graph of the whole training step replay with
// fresh dropout masks / Adam bias corrections / KL weight — kernel arguments are frozen at capture time, memory is not.
// The record is looked up per STREAM (one training engine = one rank = one stream): an engine binds its
// record to its own stream with t2v_set_step_params_stream(); launches on a stream without a binding use the process
// default installed by t2v_set_step_params() (NULL: the by-value arguments of each call).
#include <mutex>
static const t2v_step_params* g_t2v_step = nullptr;
static std::mutex g_step_mu;
static struct { hipStream_t s; const t2v_step_params* p; } g_step_tab[256];
static int g_step_n = 0;
extern "C" void t2v_set_step_params(const t2v_step_params* dev) {
    std::lock_guard<std::mutex> lk(g_step_mu);
    g_t2v_step = dev;
}
extern "C" int t2v_set_step_params_stream(void* stream, const t2v_step_params* dev) {
    std::lock_guard<std::mutex> lk(g_step_mu);
    const hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < g_step_n; ++i)
        if (g_step_tab[i].s == s) {
            if (dev) g_step_tab[i].p = dev;
            else g_step_tab[i] = g_step_tab[--g_step_n];
            return T2V_OK;
        }
    if (!dev) return T2V_OK;
    if (g_step_n >= 256) return T2V_ERR_ARG;
    g_step_tab[g_step_n].s = s;
    g_step_tab[g_step_n].p = dev;
    ++g_step_n;
    return T2V_OK;
}
const t2v_step_params* t2v_step_for(hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_step_mu);
    for (int i = 0; i < g_step_n; ++i)
        if (g_step_tab[i].s == stream) return g_step_tab[i].p;
    return g_t2v_step;
}

// One launch that zeroes several device regions (grid.y = region): the per-pass resets of a decoder pass (initial
// states, sync words, granule tags) cost one kernel instead of one memset node each.
__global__ __launch_bounds__(256) void k_zero_regions(T2VZeroRegions z) {
    const int r = blockIdx.y;
    char* p = (char*)z.p[r];
    const size_t nb = z.bytes[r];
    const size_t gt = (size_t)blockIdx.x * 256 + threadIdx.x, gs = (size_t)gridDim.x * 256;
    size_t w0 = 0;
    if (!((uintptr_t)p & 15)) {
        const size_t n16 = nb >> 4;
        uint4* q = (uint4*)p;
        const uint4 zero = {0u, 0u, 0u, 0u};
        for (size_t i = gt; i < n16; i += gs) q[i] = zero;
        w0 = n16 << 2;
    }
    uint32_t* qw = (uint32_t*)p;
    for (size_t i = w0 + gt; i < (nb >> 2); i += gs) qw[i] = 0u;
}

void t2v_zero_regions(T2VZeroRegions& z, hipStream_t stream) {
    size_t mx = 0;
    int n = 0;
    for (int i = 0; i < z.n; ++i)
        if (z.p[i] && z.bytes[i]) { z.p[n] = z.p[i]; z.bytes[n] = z.bytes[i]; mx = z.bytes[i] > mx ? z.bytes[i] : mx; ++n; }
    if (!n) return;
    size_t bx = (mx + 4095) / 4096;
    if (bx > 1024) bx = 1024;
    k_zero_regions<<<dim3((unsigned)bx, (unsigned)n), 256, 0, stream>>>(z);
}

// Measurement aid: one thread writes the chip-wide 100 MHz wall clock into buf[slot].  A training engine drops these
// into its streams at phase boundaries (T2V_STAMPS=1): unlike rocprofv3 — which makes every dispatch cost >= 4.7 us and
// graph replays wait for their predecessor — the stamps show the time line of an undisturbed replay.
__global__ void k_stamp(unsigned long long* buf, int slot) {
    if (threadIdx.x == 0) buf[slot] = wall_clock64();
}
extern "C" int t2v_stamp(unsigned long long* dev_buf, int slot, void* stream) {
    if (!dev_buf || slot < 0) return T2V_ERR_ARG;
    k_stamp<<<1, 64, 0, (hipStream_t)stream>>>(dev_buf, slot);
    return t2v_check_launch();
}

// Test aid (tests/test_decoder_persist_train_gpu.py: a neighbour that holds CUs while the persistent kernels start): `wgs`
// workgroups of 256 threads that do nothing but watch the 100 MHz clock for `microseconds`.
__global__ __launch_bounds__(256) void k_spin(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while ((unsigned long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
extern "C" int t2v_debug_spin(int wgs, int microseconds, void* stream) {
    if (wgs < 1 || wgs > 4096 || microseconds < 1 || microseconds > 200000) return T2V_ERR_ARG;
    k_spin<<<wgs, 256, 0, (hipStream_t)stream>>>((unsigned long long)microseconds * 100ull);
    return t2v_check_launch();
}
