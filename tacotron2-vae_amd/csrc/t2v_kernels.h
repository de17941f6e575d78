// Internal kernel-argument structs shared by the decoder translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/t2vae.h"

enum { T2V_RNG_ATT_H = 1, T2V_RNG_ATT_C = 2, T2V_RNG_DEC_H = 3, T2V_RNG_DEC_C = 4,
       T2V_RNG_PRENET0 = 5, T2V_RNG_PRENET1 = 6 };

int t2v_check_launch();
// gemm.hip: `n` zeroed arrival counters for a launch whose workgroups find out which of them finishes last (split-K tiles,
// column-sum slices); the launch leaves them at zero
unsigned* t2v_arrival_counters(int n);
extern unsigned long long* g_t2v_prof;   // device buffer of 32 u64 or NULL (t2v_set_phase_profile)                 // records hipGetLastError() for t2v_last_error()
const t2v_step_params* t2v_step_for(hipStream_t stream);   // device-side per-step parameters bound to this stream, else the process default, else NULL

// several device regions (byte counts: multiples of 4) zeroed by ONE launch (t2v_runtime.hip)
struct T2VZeroRegions {
    void* p[10];
    size_t bytes[10];
    int n = 0;
    void add(void* ptr, size_t nbytes) { p[n] = ptr; bytes[n] = nbytes; ++n; }
};
void t2v_zero_regions(T2VZeroRegions& z, hipStream_t stream);
struct LstmFwdArgs;
struct AttnFwdArgs;

// ---- layout of the QP scratch buffer (t2v_dec_train_bufs.QP / t2v_dec_infer_bufs.QP), in floats:
//   [0, B*256*128)            per-workgroup partial queries of the current step
//   + [0, 64)                 uint32 sync / error words (attention error word = [31]; decode loop: counter [32], error [47])
//   + [64, 64 + 4096)         exchange area of the decode loop's projection/Prenet kernel
//   + [4160, 4160 + 2*B*8*Tcap)  8-byte {partial energy, epoch tag} granules of the attention forward exchange,
//                             Tcap = T_in rounded up to 16  (t2v_decoder_qp_floats() in t2vae.h returns the total)
typedef unsigned long long t2v_u64;
static inline size_t t2v_tcap(int T_in) { return (size_t)((T_in + 15) / 16) * 16; }
static inline size_t t2v_qp_sync_off(int B) { return (size_t)B * 256 * 128; }
static inline size_t t2v_qp_xchg_off(int B) { return t2v_qp_sync_off(B) + 64; }
static inline size_t t2v_qp_ex_off(int B) { return t2v_qp_sync_off(B) + 64 + 4096; }
#define T2V_MAX_T_IN 4096
static inline size_t t2v_qp_floats(int B, int T_in) { return t2v_qp_ex_off(B) + 2 * (size_t)B * 8 * t2v_tcap(T_in); }
// attention-backward slicing: positions per workgroup and slices per item
static inline int t2v_attn_bwd_js(int T_in) { return T_in <= 128 ? 16 : 32; }
static inline int t2v_attn_bwd_slices_(int T_in) { const int js = t2v_attn_bwd_js(T_in); return (T_in + js - 1) / js; }
void t2v_launch_lstm_fwd(int mode, const LstmFwdArgs& a, hipStream_t stream);
void t2v_launch_attn_fwd(const AttnFwdArgs& f, int B, int T_in, hipStream_t stream);

struct LstmFwdArgs {
    const float4* packA;
    const float4* packD;
    int k_att;
    const float* xs_prev;   // XS[t]   (B,2560) = [h_att_{t-1} | ctx_{t-1} | h_dec_{t-2}]
    float* xs_next;         // XS[t+1]
    const float* gpre_t;    // (B,4096) hoisted prenet term + biases (training) or NULL
    const float* pre_t;     // (B,256) prenet output (inference) or NULL
    const float* bias_att;
    const float* bias_dec;
    const float* ca_prev;   // CA[t]
    float* ca_cur;          // CA[t+1]
    const float* cd_prev;   // CD[t-1]
    float* cd_cur;          // CD[t]
    float* ga_t;            // GA[t]   or NULL
    float* gd_t;            // GD[t-1] or NULL
    const float* wqT;
    float* qp;
    int shared_x;           // fp32 packs: one load of the [h_att | ctx] state columns serves both cells (lstm256_stream_shared)
    int B, t, do_att, do_dec;
    float p_att, p_dec;
    uint64_t seed;
    const t2v_step_params* step;
};

struct AttnFwdArgs {
    const float* qp;
    const float* al_prev;
    const float* acum_prev;
    float* al_cur;
    float* acum_cur;
    const float* memory;
    const float* pm;
    const int32_t* lengths;
    const float* wcomb;         // (128,64) fused location filter bank (t2v_fuse_location_weights)
    const float* v;
    float* xs_next;
    float* s_save;
    int T_in;
    unsigned long long* prof;   // optional phase stamps (s_memtime) from workgroup (0,0) thread 0
    t2v_u64* ex;                // (B,8,Tcap) granules {partial energy, tag}: exchange between the 8 workgroups of an item
    unsigned* err;              // error word (bounded-spin timeout)
    unsigned epoch;             // 1-based launch index within the pass = the granule tag of this step
};

struct LstmBwdArgs {
    const float4* packBD;   // 160 tiles (uint2 per lane and k-block when the packs are bf16)
    const float4* packBA;   // 96 tiles
    const float* dgd_t;     // DGD[t]   (B,4096)
    const float* dga_n;     // DGA[t+1] (B,4096) or NULL
    float* YD;              // (B,2560)
    float* YA;              // (B,1536)
    int B;
    int flip;               // walk the k-blocks backwards (alternates per launch for L2 reuse)
};

struct AttnBwdArgs {
    const float* dHC_t;     // (B,1536)
    const float* YD;
    const float* YA;
    const float* al_cur;    // AL[t+1]
    const float* memory;
    const float* wcomb;     // (128,64) fused location filter bank
    const float* v;
    float* S_t;             // (B,T_in,128) in: tanh outputs, out: dpre
    t2v_u64* DQ_t;          // (B,S,128) granules {partial dq, tag 1}
    float* DCTX_t;          // (B,512)
    const float* ctx_t;     // XS[t+1] + 1024: attention context of step t (row stride 2560)
    const float* GP_in;     // (B,S,2,64) per-slice partial dcat rows written by reverse step t+1
    float* GP_out;          // (B,S,2,64) ... written by this step (other parity)
    float* GC;              // (B,S,Tcap) per-workgroup running copies of the cumulative-weights gradient
    float* DV;              // (B,S,128) per-slice accumulators
    int T_in;
    unsigned long long* prof;
};

struct CellBwdArgs {
    const float* YD;
    const float* YA;
    const t2v_u64* DQ_t;    // (B,S,128) granules of step t
    int S;                  // slices per item
    const float* wqT;       // (1024,128)
    const float* dHC_prev;  // dHC[t-1] (B,1536)
    const float* GA_t;      // GA[t]
    const float* CA_cur;    // CA[t+1] (c~_t)
    const float* CA_prev;   // CA[t]   (c~_{t-1})
    const float* GD_p;      // GD[t-1]
    const float* CD_cur;    // CD[t]   (c~_{t-1} of decoder_rnn)
    const float* CD_prev;   // CD[t-1]
    float* DGA_t;           // DGA[t]
    float* DGD_p;           // DGD[t-1]
    float* DCA;
    float* DCD;
    int B, t, do_att, do_dec;
    float p_att, p_dec;
    uint64_t seed;
    const t2v_step_params* step;
    unsigned* err;          // error word (bounded-spin timeout)
    unsigned long long* prof;   // optional phase stamps of cell workgroup 0 (t2v_set_phase_profile slots 24..28)
};
