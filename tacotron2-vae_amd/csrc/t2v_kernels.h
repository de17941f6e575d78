// Internal kernel-argument structs shared by the decoder translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/t2vae.h"

enum { T2V_RNG_ATT_H = 1, T2V_RNG_ATT_C = 2, T2V_RNG_DEC_H = 3, T2V_RNG_DEC_C = 4,
       T2V_RNG_PRENET0 = 5, T2V_RNG_PRENET1 = 6 };

int t2v_check_launch();
extern unsigned long long* g_t2v_prof;   // device buffer of 32 u64 or NULL (t2v_set_phase_profile)                 // records hipGetLastError() for t2v_last_error()
size_t t2v_attn_fwd_lds(int T_in);
struct LstmFwdArgs;
struct AttnFwdArgs;
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
    int B, t, do_att, do_dec;
    float p_att, p_dec;
    uint64_t seed;
    unsigned* dq_counter;   // [0] arrivals of attention slices (monotonic over the pass), [1] error word
    unsigned dq_target;
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
    const float* loc_conv;
    const float* loc_dense;
    const float* v;
    float* xs_next;
    float* s_save;
    float* conv_save;
    int T_in;
    unsigned long long* prof;   // optional phase stamps (s_memtime) from workgroup (0,0) thread 0
    float* ex;                  // (B,8,256) partial-energy exchange between the workgroups of an item
    unsigned* sync;             // [b] arrival counters (monotonic over the pass), [31] error word
    int epoch;                  // 1-based launch index within the pass (target = S * epoch)
};

struct LstmBwdArgs {
    const float4* packBD;   // 160 tiles
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
    const float* loc_conv;
    const float* loc_dense;
    const float* v;
    float* S_t;             // (B,T_in,128) in: tanh outputs, out: dpre
    float* DQ_t;            // (B,128)
    float* DCTX_t;          // (B,512)
    float* DC_t;            // (B,32,T_in)
    const float* ctx_t;     // XS[t+1] + 1024: attention context of step t (row stride 2560)
    const float* GP_in;     // (B,8,2,64) per-slice partial dcat rows written by reverse step t+1
    float* GP_out;          // (B,8,2,64) ... written by this step (other parity)
    float* GC;              // (B,8,256) per-workgroup running copies of the cumulative-weights gradient
    float* DV;              // (B,8,128) per-slice accumulators
    unsigned* dq_counter;   // bumped once per attention slice after its partial dq row is published
    int T_in;
    unsigned long long* prof;
};

struct CellBwdArgs {
    const float* YD;
    const float* YA;
    const float* DQ_t;      // (B,8,128) per-slice partials of step t
    int S;                  // slices actually written
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
    unsigned* dq_counter;   // [0] arrivals of attention slices (monotonic over the pass), [1] error word
    unsigned dq_target;
};
