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
const t2v_step_params* g_t2v_step = nullptr;
extern "C" void t2v_set_step_params(const t2v_step_params* dev) { g_t2v_step = dev; }
