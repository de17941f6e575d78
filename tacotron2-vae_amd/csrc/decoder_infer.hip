// Free-running decode (reference Decoder.inference model.py:428-464 / the loop at
// synthesizer.py:139-154).  Per emitted frame four launches, no skew (frame t feeds step t+1):
//   k_lstm_fwd<1>  attention_rnn(t), prenet columns inside K
//   k_attn_fwd     location-sensitive attention (mask = None)
//   k_lstm_fwd<2>  decoder_rnn(t)
//   k_proj_prenet  80-mel + gate projection of [h_dec_t | ctx_t], stop flag, Prenet of the new frame (layer 0 folded
//                  into the projection, one granule hop to layer 1)
#include "t2v_common.h"
#include "t2v_kernels.h"
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>

struct ProjPrenetArgs {
    const float* xs_cur;     // XS[t+1]: ctx_t at [1024,1536)
    const float* xs_next;    // XS[t+2]: h_dec_t at [1536,2560)
    const float* proj_w;     // (337,1536): rows 0..79 linear_projection, row 80 gate_layer, rows 81..336 = W0·linear_projection
    const float* proj_b;     // (337): projection / gate biases, then W0·b_projection
    const float* w1;         // (256,256) prenet layer 1
    float* mel_t;            // MEL[t]  (B,80)
    float* gate_t;           // GATE[t] (B)
    float* pre_next;         // PRE[t+1] (B,256) or NULL (caller supplies the prenet output)
    int* stop_flag;
    int B, t;
    float gate_logit_thr, p_prenet;
    uint64_t seed;
    t2v_u64* xchg;          // (8,256) granules {prenet layer-0 output, tag = frame epoch}
    unsigned* err;          // error word (bounded-spin timeout)
    unsigned epoch;         // 1-based frame index within the pass
};

// Projection + Prenet of one decoded frame: 64 workgroups x 4 waves, one output row per wave.
//   stage 1: 337 rows of length 1536 over [h_dec_t | ctx_t]: the 80 mel rows, the gate row, and — because the Prenet's
//            first layer is bias-free and LINEAR in the mel frame — its 256 pre-activations directly through the folded
//            matrix W0·P (+ W0·b), so the layer does not have to wait for the mel frame; ReLU + dropout (always on,
//            model.py:101) and an 8-byte {value, epoch} granule per output (the data is the flag)
//   stage 2: Prenet layer 1, one of its 256 rows per wave: polls the 256 granules of every item, ReLU + dropout, PRE[t+1]
// One in-kernel hop instead of two counter barriers; 2.6 MB of weights per frame spread over 64 CUs.
#define PP_NWG 64
__global__ __launch_bounds__(256) void k_proj_prenet(ProjPrenetArgs a) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int gw = blockIdx.x * 4 + wave;                 // global wave index 0..255
    const int HC = T2V_H + T2V_E;
    const int nrows = a.pre_next ? T2V_NMEL + 1 + T2V_PRE : T2V_NMEL + 1;
    // stage-2 operands requested up front: this wave's row of W1 (4 floats per lane)
    float4 w1r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.pre_next) w1r = *(const float4*)(a.w1 + (size_t)gw * T2V_PRE + 4 * lane);
    // ---- stage 1: rows gw, gw + 256 (the second pass only for the first 81 waves)
    for (int o = gw; o < nrows; o += 4 * PP_NWG) {
        const float4* wr = (const float4*)(a.proj_w + (size_t)o * HC);
        float4 wv[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) wv[i] = wr[lane + 64 * i];
        const float bias = a.proj_b[o];
        bool all_fired = true;
        for (int b = 0; b < a.B; ++b) {
            // (the Prenet-0 dropout factor of this row: a counter hash that needs none of the operands below — evaluated while
            // their loads are in flight, not behind the wave sum; round 6, as in the persistent decode kernel)
            const float drop0 = o > T2V_NMEL ? t2v_drop_scale(a.seed, T2V_RNG_PRENET0, a.t + 1, (uint32_t)(b * T2V_PRE + (o - (T2V_NMEL + 1))), a.p_prenet) : 0.f;
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int k = 4 * (lane + 64 * i);               // [h_dec_t (1024) | ctx_t (512)]
                const float* src = k < T2V_H ? a.xs_next + (size_t)b * T2V_XW + T2V_KATT + k : a.xs_cur + (size_t)b * T2V_XW + k;
                const float4 xv = *(const float4*)src;
                acc = fmaf(wv[i].x, xv.x, acc); acc = fmaf(wv[i].y, xv.y, acc);
                acc = fmaf(wv[i].z, xv.z, acc); acc = fmaf(wv[i].w, xv.w, acc);
            }
            acc = wave_sum(acc) + bias;
            if (o < T2V_NMEL) {
                if (lane == 0) a.mel_t[(size_t)b * T2V_NMEL + o] = acc;
            } else if (o == T2V_NMEL) {
                if (lane == 0) a.gate_t[b] = acc;
                // stop rule sigmoid(gate) > threshold (model.py:453; B == 1 in the reference): all items must fire.
                // This wave sees every item's gate in turn, so it can decide alone.
                all_fired = all_fired && acc > a.gate_logit_thr;
            } else if (lane == 0) {
                const int r = o - (T2V_NMEL + 1);
                float v = fmaxf(acc, 0.f) * drop0;
                __hip_atomic_store(a.xchg + (size_t)b * T2V_PRE + r, ((t2v_u64)a.epoch << 32) | (t2v_u64)__float_as_uint(v),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (o == T2V_NMEL && lane == 0 && all_fired) atomicMin(a.stop_flag, a.t);
    }
    if (!a.pre_next) return;
    // ---- stage 2: Prenet layer 1, row gw (dropout always on)
    for (int b = 0; b < a.B; ++b) {
        const t2v_u64* gq = a.xchg + (size_t)b * T2V_PRE + 4 * lane;
        const float drop1 = t2v_drop_scale(a.seed, T2V_RNG_PRENET1, a.t + 1, (uint32_t)(b * T2V_PRE + gw), a.p_prenet);      // (in front of the wait)
        float xv[4];
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const t2v_u64 x = __hip_atomic_load(gq + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                xv[i] = __uint_as_float((unsigned)x);
                ok = ok && (unsigned)(x >> 32) == a.epoch;
            }
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(2);
            if (++spins > 4000000u || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        }
        float acc = w1r.x * xv[0];
        acc = fmaf(w1r.y, xv[1], acc); acc = fmaf(w1r.z, xv[2], acc); acc = fmaf(w1r.w, xv[3], acc);
        acc = wave_sum(acc);
        if (lane == 0) {
            acc = fmaxf(acc, 0.f) * drop1;
            a.pre_next[(size_t)b * T2V_PRE + gw] = acc;
        }
    }
}

extern "C" int t2v_decoder_infer_steps(const t2v_dec_weights* w, const t2v_dec_infer_bufs* s, int B, int T_in,
                                       int t_begin, int t_end, float gate_threshold, float p_prenet,
                                       int external_prenet, uint64_t seed, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!w || !s || B < 1 || B > 8 || T_in < 1 || T_in > T2V_MAX_T_IN || t_begin < 0 || t_end <= t_begin) return T2V_ERR_ARG;
    if (!w->bias_att || !w->bias_dec || !w->wcomb) return T2V_ERR_ARG;
    unsigned* sync = (unsigned*)(s->QP + t2v_qp_sync_off(B));
    t2v_u64* ex = (t2v_u64*)(s->QP + t2v_qp_ex_off(B));
    if (t_begin == 0) {
        (void)hipMemsetAsync(sync, 0, (64 + 4096) * sizeof(uint32_t), stream);        // sync words + Prenet granule tags
        (void)hipMemsetAsync(ex, 0, sizeof(t2v_u64) * (size_t)B * 8 * t2v_tcap(T_in), stream);
    }
    const float thr = gate_threshold <= 0.f ? -INFINITY : (gate_threshold >= 1.f ? INFINITY : logf(gate_threshold / (1.f - gate_threshold)));
    const int dbgm = getenv("T2V_DEBUG_SYNC") ? atoi(getenv("T2V_DEBUG_SYNC")) : 0;
#define DBG(bit, tag) do { if (dbgm & (bit)) { hipError_t e_ = hipStreamSynchronize(stream); fprintf(stderr, "[t2v decode] t=%d %s: %s\n", t, tag, hipGetErrorString(e_)); } } while (0)
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
        a.shared_x = 0;
        a.B = B;
        a.p_att = 0.f;      // eval mode: no state dropout (F.dropout(..., self.training))
        a.p_dec = 0.f;
        a.seed = seed;
        a.step = nullptr;
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
        DBG(1, "attention_rnn");

        AttnFwdArgs f;
        f.qp = s->QP;
        f.al_prev = s->AL + (size_t)t * B * T_in;
        f.acum_prev = s->ACUM + (size_t)t * B * T_in;
        f.al_cur = s->AL + (size_t)(t + 1) * B * T_in;
        f.acum_cur = s->ACUM + (size_t)(t + 1) * B * T_in;
        f.memory = s->memory;
        f.pm = s->pm;
        f.lengths = s->lengths;
        f.wcomb = w->wcomb;
        f.v = w->v;
        f.xs_next = s->XS + (size_t)(t + 1) * B * T2V_XW;
        f.s_save = nullptr;
        f.T_in = T_in;
        f.prof = nullptr;
        f.ex = ex;
        f.err = sync + 31;
        f.epoch = (unsigned)t + 1u;
        t2v_launch_attn_fwd(f, B, T_in, stream);
        DBG(2, "attention");

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
        DBG(4, "decoder_rnn");

        ProjPrenetArgs p;
        p.xs_cur = s->XS + (size_t)(t + 1) * B * T2V_XW;
        p.xs_next = s->XS + (size_t)(t + 2) * B * T2V_XW;
        p.proj_w = s->proj_w;
        p.proj_b = s->proj_b;
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
        p.xchg = (t2v_u64*)(s->QP + t2v_qp_xchg_off(B));
        p.err = sync + 47;
        p.epoch = (unsigned)t + 1u;
        k_proj_prenet<<<PP_NWG, 256, 0, stream>>>(p);
        DBG(8, "proj_prenet");
    }
    return t2v_check_launch();
}
