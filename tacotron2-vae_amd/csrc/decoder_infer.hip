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
    float* xchg;            // [0,768) mel exchange (8 x 96), [1024,3072) layer-0 exchange (8 x 256)
    unsigned* sync;         // [0] arrival counter (monotonic over the pass), [15] error word (QP sync words 32 / 47)
    int epoch;              // 1-based frame index within the pass
};

// 16 cooperating workgroups x 256 threads (B <= 8): the 0.83 MB of projection + Prenet weights touched per frame
// are spread over 16 CUs (a single CU pulls only ~40 GB/s of non-local data), with two bounded-spin group
// barriers between the three dependent stages: [80-mel + gate projection] -> [Prenet layer 0] -> [Prenet layer 1].
#define PP_NWG 16
__device__ __forceinline__ bool pp_barrier(unsigned* cnt, unsigned* err, unsigned target, int* ok_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int good = 1;
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > 4000000u || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = 0;
                break;
            }
        }
        *ok_flag = good;
    }
    __syncthreads();
    return *ok_flag != 0;
}

__global__ __launch_bounds__(256) void k_proj_prenet(ProjPrenetArgs a) {
    __shared__ __attribute__((aligned(16))) float xin[8][T2V_PRE];     // stage input: mel (80) or layer-0 output (256)
    __shared__ int ok_flag;
    const int gidx = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int HC = T2V_H + T2V_E;
    // ---- stage 1: outputs o = gidx, gidx+16, ... of the 81-row projection; one wave per output row
    for (int o = gidx + PP_NWG * wave; o < T2V_NMEL + 1; o += PP_NWG * 4) {
        const float4* wr = (const float4*)(a.proj_w + (size_t)o * HC);
        float4 wv[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) wv[i] = wr[lane + 64 * i];
        for (int b = 0; b < a.B; ++b) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int k = 4 * (lane + 64 * i);               // [h_dec_t (1024) | ctx_t (512)]
                const float* src = k < T2V_H ? a.xs_next + (size_t)b * T2V_XW + T2V_KATT + k : a.xs_cur + (size_t)b * T2V_XW + k;
                const float4 xv = *(const float4*)src;
                acc = fmaf(wv[i].x, xv.x, acc); acc = fmaf(wv[i].y, xv.y, acc);
                acc = fmaf(wv[i].z, xv.z, acc); acc = fmaf(wv[i].w, xv.w, acc);
            }
            acc = wave_sum(acc) + a.proj_b[o];
            if (lane == 0) {
                if (o < T2V_NMEL) {
                    a.mel_t[(size_t)b * T2V_NMEL + o] = acc;
                    if (a.pre_next) __hip_atomic_store(a.xchg + b * 96 + o, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    a.gate_t[b] = acc;
                    // stop rule sigmoid(gate) > threshold (model.py:453; B == 1 in the reference): all items must fire.
                    // This wave sees every item's gate in turn, so it can decide alone.
                    xin[0][b] = acc;
                }
            }
        }
        if (o == T2V_NMEL && lane == 0) {
            bool all = true;
            for (int b = 0; b < a.B; ++b) all = all && (xin[0][b] > a.gate_logit_thr);
            if (all) atomicMin(a.stop_flag, a.t);
        }
    }
    if (!a.pre_next) return;
    if (!pp_barrier(a.sync, a.sync + 15, (unsigned)PP_NWG * (2u * (unsigned)a.epoch - 1u), &ok_flag)) return;
    // ---- stage 2: Prenet layer 0, outputs [16g, 16g+16) (dropout always on, model.py:101)
    for (int i = tid; i < a.B * T2V_NMEL; i += 256) {
        const int b = i / T2V_NMEL, k = i - b * T2V_NMEL;
        xin[b][k] = __hip_atomic_load(a.xchg + b * 96 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    {
        const int o = 16 * gidx + (tid & 15), b = tid >> 4;
        if (b < a.B) {
            const float4* w = (const float4*)(a.w0 + (size_t)o * T2V_NMEL);
            const float4* x = (const float4*)xin[b];
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < T2V_NMEL / 4; ++k) {
                const float4 wv = w[k], xv = x[k];
                acc = fmaf(wv.x, xv.x, acc); acc = fmaf(wv.y, xv.y, acc);
                acc = fmaf(wv.z, xv.z, acc); acc = fmaf(wv.w, xv.w, acc);
            }
            acc = fmaxf(acc, 0.f) * t2v_drop_scale(a.seed, T2V_RNG_PRENET0, a.t + 1, (uint32_t)(b * T2V_PRE + o), a.p_prenet);
            __hip_atomic_store(a.xchg + 1024 + b * T2V_PRE + o, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (!pp_barrier(a.sync, a.sync + 15, (unsigned)PP_NWG * 2u * (unsigned)a.epoch, &ok_flag)) return;
    // ---- stage 3: Prenet layer 1, outputs [16g, 16g+16)
    for (int i = tid; i < a.B * T2V_PRE; i += 256)
        xin[i >> 8][i & 255] = __hip_atomic_load(a.xchg + 1024 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    {
        const int o = 16 * gidx + (tid & 15), b = tid >> 4;
        if (b < a.B) {
            const float4* w = (const float4*)(a.w1 + (size_t)o * T2V_PRE);
            const float4* x = (const float4*)xin[b];
            float acc = 0.f;
#pragma unroll 16
            for (int k = 0; k < T2V_PRE / 4; ++k) {
                const float4 wv = w[k], xv = x[k];
                acc = fmaf(wv.x, xv.x, acc); acc = fmaf(wv.y, xv.y, acc);
                acc = fmaf(wv.z, xv.z, acc); acc = fmaf(wv.w, xv.w, acc);
            }
            acc = fmaxf(acc, 0.f) * t2v_drop_scale(a.seed, T2V_RNG_PRENET1, a.t + 1, (uint32_t)(b * T2V_PRE + o), a.p_prenet);
            a.pre_next[(size_t)b * T2V_PRE + o] = acc;
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
        (void)hipMemsetAsync(sync, 0, 64 * sizeof(uint32_t), stream);
        (void)hipMemsetAsync(ex, 0, sizeof(t2v_u64) * (size_t)B * 8 * t2v_tcap(T_in), stream);
    }
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
        p.xchg = s->QP + t2v_qp_xchg_off(B);
        p.sync = sync + 32;
        p.epoch = t + 1;
        k_proj_prenet<<<PP_NWG, 256, 0, stream>>>(p);
    }
    return t2v_check_launch();
}
