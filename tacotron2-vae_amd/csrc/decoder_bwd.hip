// Decoder recurrence, backward (hand-written BPTT; replaces autograd over Decoder.decode,
// reference model.py:346-389 / train.py:225).  Per reverse step t three kernels:
//   k_lstm_bwd : data-gradient GEMVs  YD = Wcat_dec^T·dgd_t,  YA = Wcat_att^T·dga_{t+1}   (MFMA)
//   k_attn_bwd : attention(t) backward (context, softmax, tanh/v, location dense + conv)
//   k_cell_bwd : LSTM cell backward for attention_rnn(t) and decoder_rnn(t-1)
// Weight gradients are NOT accumulated here: the saved per-step pre-activation gradients
// (DGA, DGD, DQ, dpre, DC, DCTX) feed time-batched GEMMs after the loop.
#include "t2v_common.h"
#include "t2v_kernels.h"

#define MFMA4(ACC, WV, XV)                  \
    ACC = mfma16x4((WV).x, (XV).x, ACC);    \
    ACC = mfma16x4((WV).y, (XV).y, ACC);    \
    ACC = mfma16x4((WV).z, (XV).z, ACC);    \
    ACC = mfma16x4((WV).w, (XV).w, ACC)

__global__ __launch_bounds__(256) void k_lstm_bwd(LstmBwdArgs a) {
    const int w = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = lane & 15, g = lane >> 4;
    const bool dec = w < T2V_XW / 16;
    const int wt = dec ? w : w - T2V_XW / 16;
    const float* kv = dec ? a.dgd_t : a.dga_n;
    if (!kv) return;   // block-uniform
    const bool bvalid = b < a.B;
    __shared__ f32x4 red[4][64];
    const float4* p = (dec ? a.packBD : a.packBA) + (size_t)wt * 256 * 64 + lane;
    const float* xrow = kv + (size_t)(bvalid ? b : 0) * T2V_G + 4 * g;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int kb0 = 64 * wave;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) {
        const int kb = kb0 + i;
        const float4 x = bvalid ? *(const float4*)(xrow + 16 * kb) : z4;
        const float4 wv = p[(size_t)kb * 64];
        MFMA4(acc, wv, x);
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && bvalid) {
        const f32x4 s = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
        float* y = dec ? a.YD + (size_t)b * T2V_XW : a.YA + (size_t)b * T2V_KATT;
        *(float4*)(y + 16 * wt + 4 * g) = make_float4(s[0], s[1], s[2], s[3]);
    }
}

// LDS carve (floats): dctx[512] | alpha[TpR] | dal[TpR] | de[TpR] | dpre[Tp*128] | dcs[32*DS] | scr[512]
static __host__ __device__ inline int attn_bwd_ds(int Tp) { return (Tp + 30) | 1; }
size_t t2v_attn_bwd_lds(int Tp) {
    const int TpR = (Tp + 3) & ~3;
    return sizeof(float) * (T2V_E + 3 * TpR + (size_t)Tp * T2V_A + T2V_F * attn_bwd_ds(Tp) + 512);
}

__global__ __launch_bounds__(256) void k_attn_bwd(AttnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Tp = a.T_in, TpR = (Tp + 3) & ~3, DS = attn_bwd_ds(Tp);
    float* dctx = smem;
    float* alpha = dctx + T2V_E;
    float* dal = alpha + TpR;
    float* de = dal + TpR;
    float* dpre = de + TpR;
    float* dcs = dpre + (size_t)Tp * T2V_A;
    float* scr = dcs + T2V_F * DS;

    // 1. total gradient of the context of step t
    for (int e = tid; e < T2V_E; e += 256) {
        const float v = a.dHC_t[(size_t)b * (T2V_H + T2V_E) + T2V_H + e] + a.YD[(size_t)b * T2V_XW + T2V_H + e] +
                        a.YA[(size_t)b * T2V_KATT + T2V_H + e];
        dctx[e] = v;
        a.DCTX_t[(size_t)b * T2V_E + e] = v;
    }
    for (int j = tid; j < Tp; j += 256) alpha[j] = a.al_cur[(size_t)b * Tp + j];
    for (int i = tid; i < T2V_F * DS; i += 256) dcs[i] = 0.f;
    __syncthreads();

    // 2. d alpha[j] = dctx . memory[j] + (grad via prev-channel of step t+1) + (grad via cumulative)
    for (int j = wave; j < Tp; j += 4) {
        const float* mrow = a.memory + ((size_t)b * Tp + j) * T2V_E;
        float acc = 0.f;
#pragma unroll
        for (int c = lane * 4; c < T2V_E; c += 256) {
            const float4 m = *(const float4*)(mrow + c);
            const float4 dd = *(const float4*)(dctx + c);
            acc = fmaf(m.x, dd.x, acc);
            acc = fmaf(m.y, dd.y, acc);
            acc = fmaf(m.z, dd.z, acc);
            acc = fmaf(m.w, dd.w, acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) dal[j] = acc + a.GPREV[(size_t)b * Tp + j] + a.GCUM[(size_t)b * Tp + j];
    }
    __syncthreads();

    // 3. softmax backward
    {
        float part = 0.f;
        for (int j = tid; j < Tp; j += 256) part = fmaf(alpha[j], dal[j], part);
        part = wave_sum(part);
        if (lane == 0) scr[wave] = part;
        __syncthreads();
        const float dot = (scr[0] + scr[1]) + (scr[2] + scr[3]);
        for (int j = tid; j < Tp; j += 256) de[j] = alpha[j] * (dal[j] - dot);
        __syncthreads();
    }

    // 4. through v . tanh(.)
    {
        const int d = tid & (T2V_A - 1), jh = tid >> 7;
        const float vd = a.v[d];
        float* sp = a.S_t + (size_t)b * Tp * T2V_A + d;
        float dq = 0.f, dv = 0.f;
        for (int j = jh; j < Tp; j += 2) {
            const float s = sp[(size_t)j * T2V_A];
            const float dej = de[j];
            const float dp = dej * vd * (1.0f - s * s);
            dpre[j * T2V_A + d] = dp;
            sp[(size_t)j * T2V_A] = dp;
            dq += dp;
            dv = fmaf(dej, s, dv);
        }
        scr[tid] = dq;
        scr[256 + tid] = dv;
        __syncthreads();
        if (tid < T2V_A) {
            a.DQ_t[(size_t)b * T2V_A + tid] = scr[tid] + scr[tid + 128];
            a.DV[(size_t)b * T2V_A + tid] += scr[256 + tid] + scr[256 + tid + 128];
        }
    }

    // 5. through location_dense: dc[f][j] = sum_d D[d][f] dpre[j][d]
    {
        const int f = tid & 31, jg = tid >> 5;
        float dreg[T2V_A];
#pragma unroll
        for (int d = 0; d < T2V_A; ++d) dreg[d] = a.loc_dense[d * T2V_F + f];
        for (int j = jg; j < Tp; j += 8) {
            const float* dp = dpre + j * T2V_A;
            float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
            for (int d = 0; d < T2V_A; d += 2) {
                acc0 = fmaf(dreg[d], dp[d], acc0);
                acc1 = fmaf(dreg[d + 1], dp[d + 1], acc1);
            }
            const float acc = acc0 + acc1;
            dcs[f * DS + 15 + j] = acc;
            a.DC_t[((size_t)b * T2V_F + f) * Tp + j] = acc;
        }
    }
    __syncthreads();

    // 6. through location_conv (transposed): dcat[ch][j] = sum_{f,k} Wc[f][ch][k] dc[f][j+15-k]
    {
        const int f = tid & 31, grp = tid >> 5;
        float wc[2][T2V_KS];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int k = 0; k < T2V_KS; ++k) wc[ch][k] = a.loc_conv[(f * 2 + ch) * T2V_KS + k];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            for (int j = grp; j < Tp; j += 8) {
                const float* row = dcs + f * DS + j;   // dcs index (15 + j + 15 - k) = j + 30 - k
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < T2V_KS; ++k) acc = fmaf(wc[ch][k], row[30 - k], acc);
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
                if (f == 0) {
                    if (ch == 0) a.GPREV[(size_t)b * Tp + j] = acc;
                    else a.GCUM[(size_t)b * Tp + j] += acc;
                }
            }
        }
    }
}

// grid = 64 blocks x 256 threads; thread = (unit U, item b)
__global__ __launch_bounds__(256) void k_cell_bwd(CellBwdArgs a) {
    const int U = blockIdx.x * 16 + (threadIdx.x >> 4), b = threadIdx.x & 15;
    if (b >= a.B) return;
    const uint32_t idx = (uint32_t)b * T2V_H + U;
    const size_t bu = (size_t)b * T2V_H + U;
    if (a.do_att) {
        const int t = a.t;
        const float* wq = a.wqT + (size_t)U * T2V_A;
        const float* dq = a.DQ_t + (size_t)b * T2V_A;
        float dot0 = 0.f, dot1 = 0.f;
#pragma unroll 8
        for (int d = 0; d < T2V_A; d += 4) {
            const float4 w4 = *(const float4*)(wq + d);
            const float4 q4 = *(const float4*)(dq + d);
            dot0 = fmaf(w4.x, q4.x, dot0);
            dot1 = fmaf(w4.y, q4.y, dot1);
            dot0 = fmaf(w4.z, q4.z, dot0);
            dot1 = fmaf(w4.w, q4.w, dot1);
        }
        const float dh = a.YD[(size_t)b * T2V_XW + U] + a.YA[(size_t)b * T2V_KATT + U] + (dot0 + dot1);
        const float fh = t2v_drop_scale(a.seed, T2V_RNG_ATT_H, t, idx, a.p_att);
        const float fc = t2v_drop_scale(a.seed, T2V_RNG_ATT_C, t, idx, a.p_att);
        const float* ga = a.GA_t + (size_t)b * T2V_G + U;
        const float gi = ga[0], gf = ga[T2V_H], gg = ga[2 * T2V_H], go = ga[3 * T2V_H];
        const float tc = tanhf(a.CA_cur[bu]);
        const float dht = dh * fh;
        const float dct = a.DCA[bu] * fc + dht * go * (1.0f - tc * tc);
        float cprev = a.CA_prev[bu];
        if (t > 0) cprev *= t2v_drop_scale(a.seed, T2V_RNG_ATT_C, t - 1, idx, a.p_att);
        float* o = a.DGA_t + (size_t)b * T2V_G + U;
        o[0] = dct * gg * gi * (1.0f - gi);
        o[T2V_H] = dct * cprev * gf * (1.0f - gf);
        o[2 * T2V_H] = dct * gi * (1.0f - gg * gg);
        o[3 * T2V_H] = dht * tc * go * (1.0f - go);
        a.DCA[bu] = dct * gf;
    }
    if (a.do_dec) {
        const int td = a.t - 1;
        const float dh = a.dHC_prev[(size_t)b * (T2V_H + T2V_E) + U] + a.YD[(size_t)b * T2V_XW + T2V_KATT + U];
        const float fh = t2v_drop_scale(a.seed, T2V_RNG_DEC_H, td, idx, a.p_dec);
        const float fc = t2v_drop_scale(a.seed, T2V_RNG_DEC_C, td, idx, a.p_dec);
        const float* gd = a.GD_p + (size_t)b * T2V_G + U;
        const float gi = gd[0], gf = gd[T2V_H], gg = gd[2 * T2V_H], go = gd[3 * T2V_H];
        const float tc = tanhf(a.CD_cur[bu]);
        const float dht = dh * fh;
        const float dct = a.DCD[bu] * fc + dht * go * (1.0f - tc * tc);
        float cprev = a.CD_prev[bu];
        if (td > 0) cprev *= t2v_drop_scale(a.seed, T2V_RNG_DEC_C, td - 1, idx, a.p_dec);
        float* o = a.DGD_p + (size_t)b * T2V_G + U;
        o[0] = dct * gg * gi * (1.0f - gi);
        o[T2V_H] = dct * cprev * gf * (1.0f - gf);
        o[2 * T2V_H] = dct * gi * (1.0f - gg * gg);
        o[3 * T2V_H] = dht * tc * go * (1.0f - go);
        a.DCD[bu] = dct * gf;
    }
}

extern "C" int t2v_decoder_train_bwd(const t2v_dec_weights* w, const t2v_dec_train_bufs* s,
                                     const t2v_dec_bwd_bufs* g, int B, int T_in, int T_out,
                                     float p_att, float p_dec, uint64_t seed, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!w || !s || !g || B < 1 || B > 16 || T_in < 1 || T_out < 1) return T2V_ERR_ARG;
    if (!w->packB_att || !w->packB_dec) return T2V_ERR_ARG;
    const size_t lds = t2v_attn_bwd_lds(T_in);
    if (lds > 160 * 1024) return T2V_ERR_ARG;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)k_attn_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipMemsetAsync(g->YD, 0, sizeof(float) * B * T2V_XW, stream);
    (void)hipMemsetAsync(g->YA, 0, sizeof(float) * B * T2V_KATT, stream);
    (void)hipMemsetAsync(g->DCA, 0, sizeof(float) * B * T2V_H, stream);
    (void)hipMemsetAsync(g->DCD, 0, sizeof(float) * B * T2V_H, stream);
    (void)hipMemsetAsync(g->GPREV, 0, sizeof(float) * B * T_in, stream);
    (void)hipMemsetAsync(g->GCUM, 0, sizeof(float) * B * T_in, stream);
    (void)hipMemsetAsync(g->DV, 0, sizeof(float) * B * T2V_A, stream);

    const size_t HC = T2V_H + T2V_E;
    for (int t = T_out; t >= 0; --t) {
        if (t < T_out) {
            LstmBwdArgs l;
            l.packBD = (const float4*)w->packB_dec;
            l.packBA = (const float4*)w->packB_att;
            l.dgd_t = g->DGD + (size_t)t * B * T2V_G;
            l.dga_n = t + 1 < T_out ? g->DGA + (size_t)(t + 1) * B * T2V_G : nullptr;
            l.YD = g->YD;
            l.YA = g->YA;
            l.B = B;
            k_lstm_bwd<<<T2V_NWG, 256, 0, stream>>>(l);

            AttnBwdArgs f;
            f.dHC_t = g->dHC + (size_t)t * B * HC;
            f.YD = g->YD;
            f.YA = g->YA;
            f.al_cur = s->AL + (size_t)(t + 1) * B * T_in;
            f.memory = s->memory;
            f.loc_conv = w->loc_conv;
            f.loc_dense = w->loc_dense;
            f.v = w->v;
            f.S_t = s->S + (size_t)t * B * T_in * T2V_A;
            f.DQ_t = g->DQ + (size_t)t * B * T2V_A;
            f.DCTX_t = g->DCTX + (size_t)t * B * T2V_E;
            f.DC_t = g->DC + (size_t)t * B * T2V_F * T_in;
            f.GPREV = g->GPREV;
            f.GCUM = g->GCUM;
            f.DV = g->DV;
            f.T_in = T_in;
            k_attn_bwd<<<B, 256, lds, stream>>>(f);
        }
        CellBwdArgs c;
        c.YD = g->YD;
        c.YA = g->YA;
        c.DQ_t = t < T_out ? g->DQ + (size_t)t * B * T2V_A : nullptr;
        c.wqT = w->wqT;
        c.dHC_prev = t >= 1 ? g->dHC + (size_t)(t - 1) * B * HC : nullptr;
        c.GA_t = t < T_out ? s->GA + (size_t)t * B * T2V_G : nullptr;
        c.CA_cur = s->CA + (size_t)(t + 1) * B * T2V_H;
        c.CA_prev = s->CA + (size_t)t * B * T2V_H;
        c.GD_p = t >= 1 ? s->GD + (size_t)(t - 1) * B * T2V_G : nullptr;
        c.CD_cur = s->CD + (size_t)t * B * T2V_H;
        c.CD_prev = t >= 1 ? s->CD + (size_t)(t - 1) * B * T2V_H : nullptr;
        c.DGA_t = t < T_out ? g->DGA + (size_t)t * B * T2V_G : nullptr;
        c.DGD_p = t >= 1 ? g->DGD + (size_t)(t - 1) * B * T2V_G : nullptr;
        c.DCA = g->DCA;
        c.DCD = g->DCD;
        c.B = B;
        c.t = t;
        c.do_att = t < T_out;
        c.do_dec = t >= 1;
        c.p_att = p_att;
        c.p_dec = p_dec;
        c.seed = seed;
        if (c.do_att || c.do_dec) k_cell_bwd<<<T2V_H / 16, 256, 0, stream>>>(c);
    }
    return t2v_check_launch();
}
