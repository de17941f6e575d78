// Free-running decode (reference Decoder.inference model.py:428-464 / the loop at
// synthesizer.py:139-154).  Per emitted frame four launches, no skew (frame t feeds step t+1):
//   k_lstm_fwd<1>  attention_rnn(t), prenet columns inside K
//   k_attn_fwd     location-sensitive attention (mask = None)
//   k_lstm_fwd<2>  decoder_rnn(t)
//   k_proj_prenet  80-mel + gate projection of [h_dec_t | ctx_t], stop flag, Prenet of the new frame
#include "t2v_common.h"
#include "t2v_kernels.h"
#include <limits.h>

struct ProjPrenetArgs {
    const float* xs_cur;     // XS[t+1]: ctx_t at [1024,1536)
    const float* xs_next;    // XS[t+2]: h_dec_t at [1536,2560)
    const float* proj_w;     // (81,1536): rows 0..79 linear_projection, row 80 gate_layer
    const float* proj_b;     // (81)
    const float* w0;         // (256,80)  prenet layer 0
    const float* w1;         // (256,256) prenet layer 1
    float* mel_t;            // MEL[t]  (B,80)
    float* gate_t;           // GATE[t] (B)
    float* pre_next;         // PRE[t+1] (B,256) or NULL (caller supplies the prenet output)
    int* stop_flag;
    int B, t;
    float gate_logit_thr, p_prenet;
    uint64_t seed;
};

// one workgroup, 1024 threads (16 waves); B <= 8
__global__ __launch_bounds__(1024) void k_proj_prenet(ProjPrenetArgs a) {
    __shared__ float hc[8][T2V_H + T2V_E];
    __shared__ float melv[8][T2V_NMEL + 1];
    __shared__ float p0[8][T2V_PRE];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int HC = T2V_H + T2V_E;
    for (int i = tid; i < a.B * HC; i += 1024) {
        const int b = i / HC, k = i - b * HC;
        hc[b][k] = k < T2V_H ? a.xs_next[(size_t)b * T2V_XW + T2V_KATT + k] : a.xs_cur[(size_t)b * T2V_XW + k];
    }   // note k in [1024,1536) indexes ctx at the same offset inside the XS row
    __syncthreads();
    // projection: wave per output row
    for (int o = wave; o < T2V_NMEL + 1; o += 16) {
        const float4* wr = (const float4*)(a.proj_w + (size_t)o * HC);
        float4 wv[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) wv[i] = wr[lane + 64 * i];
        for (int b = 0; b < a.B; ++b) {
            const float4* x = (const float4*)hc[b];
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const float4 xv = x[lane + 64 * i];
                acc = fmaf(wv[i].x, xv.x, acc); acc = fmaf(wv[i].y, xv.y, acc);
                acc = fmaf(wv[i].z, xv.z, acc); acc = fmaf(wv[i].w, xv.w, acc);
            }
            acc = wave_sum(acc);
            if (lane == 0) melv[b][o] = acc + a.proj_b[o];
        }
    }
    __syncthreads();
    for (int i = tid; i < a.B * (T2V_NMEL + 1); i += 1024) {
        const int b = i / (T2V_NMEL + 1), o = i - b * (T2V_NMEL + 1);
        if (o < T2V_NMEL) a.mel_t[(size_t)b * T2V_NMEL + o] = melv[b][o];
        else a.gate_t[b] = melv[b][o];
    }
    if (tid == 0) {   // stop rule sigmoid(gate) > threshold (model.py:453; well defined for B == 1)
        bool all = true;
        for (int b = 0; b < a.B; ++b) all = all && (melv[b][T2V_NMEL] > a.gate_logit_thr);
        if (all) atomicMin(a.stop_flag, a.t);
    }
    if (!a.pre_next) return;
    // Prenet of the frame just produced (dropout always on, model.py:101)
    for (int i = tid; i < a.B * T2V_PRE; i += 1024) {
        const int b = i >> 8, o = i & 255;
        const float* w = a.w0 + (size_t)o * T2V_NMEL;
        float acc = 0.f;
#pragma unroll 8
        for (int k = 0; k < T2V_NMEL; ++k) acc = fmaf(w[k], melv[b][k], acc);
        acc = fmaxf(acc, 0.f) * t2v_drop_scale(a.seed, T2V_RNG_PRENET0, a.t + 1, (uint32_t)i, a.p_prenet);
        p0[b][o] = acc;
    }
    __syncthreads();
    for (int i = tid; i < a.B * T2V_PRE; i += 1024) {
        const int b = i >> 8, o = i & 255;
        const float4* w = (const float4*)(a.w1 + (size_t)o * T2V_PRE);
        const float4* x = (const float4*)p0[b];
        float acc = 0.f;
#pragma unroll 8
        for (int k = 0; k < T2V_PRE / 4; ++k) {
            const float4 wv = w[k], xv = x[k];
            acc = fmaf(wv.x, xv.x, acc); acc = fmaf(wv.y, xv.y, acc);
            acc = fmaf(wv.z, xv.z, acc); acc = fmaf(wv.w, xv.w, acc);
        }
        acc = fmaxf(acc, 0.f) * t2v_drop_scale(a.seed, T2V_RNG_PRENET1, a.t + 1, (uint32_t)i, a.p_prenet);
        a.pre_next[(size_t)b * T2V_PRE + o] = acc;
    }
}

extern "C" int t2v_decoder_infer_steps(const t2v_dec_weights* w, const t2v_dec_infer_bufs* s, int B, int T_in,
                                       int t_begin, int t_end, float gate_threshold, float p_prenet,
                                       int external_prenet, uint64_t seed, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!w || !s || B < 1 || B > 8 || T_in < 1 || T_in > 256 || t_begin < 0 || t_end <= t_begin) return T2V_ERR_ARG;
    if (!w->bias_att || !w->bias_dec) return T2V_ERR_ARG;
    float* qp_tail = s->QP + (size_t)B * T2V_NWG * T2V_A;
    if (t_begin == 0) (void)hipMemsetAsync(qp_tail + 32768, 0, 32 * sizeof(uint32_t), stream);
    const float thr = gate_threshold <= 0.f ? -INFINITY : (gate_threshold >= 1.f ? INFINITY : logf(gate_threshold / (1.f - gate_threshold)));
    for (int t = t_begin; t < t_end; ++t) {
        LstmFwdArgs a;
        a.packA = (const float4*)w->packF_att;
        a.packD = (const float4*)w->packF_dec;
        a.k_att = T2V_KATT_INF;
        a.gpre_t = nullptr;
        a.pre_t = s->PRE + (size_t)t * B * T2V_PRE;
        a.bias_att = w->bias_att;
        a.bias_dec = w->bias_dec;
        a.ga_t = nullptr;
        a.gd_t = nullptr;
        a.wqT = w->wqT;
        a.qp = s->QP;
        a.B = B;
        a.p_att = 0.f;      // eval mode: no state dropout (F.dropout(..., self.training))
        a.p_dec = 0.f;
        a.seed = seed;
        // attention_rnn(t): XS[t] -> XS[t+1][0:1024]
        a.xs_prev = s->XS + (size_t)t * B * T2V_XW;
        a.xs_next = s->XS + (size_t)(t + 1) * B * T2V_XW;
        a.ca_prev = s->CA + (size_t)t * B * T2V_H;
        a.ca_cur = s->CA + (size_t)(t + 1) * B * T2V_H;
        a.cd_prev = nullptr;
        a.cd_cur = nullptr;
        a.t = t;
        a.do_att = 1;
        a.do_dec = 0;
        t2v_launch_lstm_fwd(1, a, stream);

        AttnFwdArgs f;
        f.qp = s->QP;
        f.al_prev = s->AL + (size_t)t * B * T_in;
        f.acum_prev = s->ACUM + (size_t)t * B * T_in;
        f.al_cur = s->AL + (size_t)(t + 1) * B * T_in;
        f.acum_cur = s->ACUM + (size_t)(t + 1) * B * T_in;
        f.memory = s->memory;
        f.pm = s->pm;
        f.lengths = s->lengths;
        f.loc_conv = w->loc_conv;
        f.loc_dense = w->loc_dense;
        f.v = w->v;
        f.xs_next = s->XS + (size_t)(t + 1) * B * T2V_XW;
        f.s_save = nullptr;
        f.conv_save = nullptr;
        f.T_in = T_in;
        f.prof = nullptr;
        f.ex = qp_tail;
        f.sync = (unsigned*)(qp_tail + 32768);
        f.epoch = t + 1;
        t2v_launch_attn_fwd(f, B, T_in, stream);

        // decoder_rnn(t): XS[t+1] -> XS[t+2][1536:]   (time index t+1 in the kernel's skewed convention)
        a.xs_prev = s->XS + (size_t)(t + 1) * B * T2V_XW;
        a.xs_next = s->XS + (size_t)(t + 2) * B * T2V_XW;
        a.cd_prev = s->CD + (size_t)t * B * T2V_H;
        a.cd_cur = s->CD + (size_t)(t + 1) * B * T2V_H;
        a.pre_t = nullptr;
        a.t = t + 1;
        a.do_att = 0;
        a.do_dec = 1;
        t2v_launch_lstm_fwd(2, a, stream);

        ProjPrenetArgs p;
        p.xs_cur = s->XS + (size_t)(t + 1) * B * T2V_XW;
        p.xs_next = s->XS + (size_t)(t + 2) * B * T2V_XW;
        p.proj_w = s->proj_w;
        p.proj_b = s->proj_b;
        p.w0 = s->prenet_w0;
        p.w1 = s->prenet_w1;
        p.mel_t = s->MEL + (size_t)t * B * T2V_NMEL;
        p.gate_t = s->GATE + (size_t)t * B;
        p.pre_next = external_prenet ? nullptr : s->PRE + (size_t)(t + 1) * B * T2V_PRE;
        p.stop_flag = s->stop_flag;
        p.B = B;
        p.t = t;
        p.gate_logit_thr = thr;
        p.p_prenet = p_prenet;
        p.seed = seed;
        k_proj_prenet<<<1, 1024, 0, stream>>>(p);
    }
    return t2v_check_launch();
}
