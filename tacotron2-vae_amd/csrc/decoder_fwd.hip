// Decoder recurrence, forward: weight packing + the two per-step kernels.
//   k_lstm_fwd : attention_rnn(t) and decoder_rnn(t-1) gate GEMVs on MFMA (weights streamed in
//                fragment order), LSTM cell update, state dropout, partial query projection.
//   k_attn_fwd : location-sensitive attention of step t (query sum, location conv + dense,
//                tanh/v energies, masked softmax, context).
// Reference semantics: Decoder.decode model.py:346-389, Attention.forward model.py:67-88.
#include "t2v_common.h"
#include "t2v_kernels.h"

// ------------------------------------------------------------------------------------------
// Weight packing.  Forward tile w (16 rows = 4 units x 4 gates, unit-major) of a (4096,K)
// gate-major matrix:   P[w][kb][lane][i] = W[(lane&3)*H + 4w + ((lane&15)>>2)][16kb + 4(lane>>4) + i]
// so one float4 per lane feeds four v_mfma_f32_16x16x4_f32 A operands (k = 16kb+4g+i, i=0..3).
__global__ void k_pack_fwd(const float* __restrict__ W, int K, float4* __restrict__ P) {
    const int nkb = K / 16;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // float4 index
    const size_t total = (size_t)T2V_NWG * nkb * 64;
    if (idx >= total) return;
    const int lane = idx & 63;
    const int kb = (idx >> 6) % nkb;
    const int w = (idx >> 6) / nkb;
    const int arow = lane & 15, g = lane >> 4;
    const int row = (arow & 3) * T2V_H + 4 * w + (arow >> 2);
    const float* src = W + (size_t)row * K + 16 * kb + 4 * g;
    P[idx] = make_float4(src[0], src[1], src[2], src[3]);
}
// Backward (transposed) tile n-tile w' of W^T (N = K columns of W become rows), reduction dim
// = 4096 gate rows:  PB[w'][kb][lane][i] = W[16kb + 4(lane>>4) + i][16w' + (lane&15)]
__global__ void k_pack_bwd(const float* __restrict__ W, int K, int ncols, float4* __restrict__ P) {
    const int nkb = T2V_G / 16;   // 256
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)(ncols / 16) * nkb * 64;
    if (idx >= total) return;
    const int lane = idx & 63;
    const int kb = (idx >> 6) % nkb;
    const int wt = (idx >> 6) / nkb;
    const int n = 16 * wt + (lane & 15);
    const int k0 = 16 * kb + 4 * (lane >> 4);
    P[idx] = make_float4(W[(size_t)(k0 + 0) * K + n], W[(size_t)(k0 + 1) * K + n],
                         W[(size_t)(k0 + 2) * K + n], W[(size_t)(k0 + 3) * K + n]);
}

// ------------------------------------------------------------------------------------------
#define MFMA4(ACC, WV, XV)                  \
    ACC = mfma16x4((WV).x, (XV).x, ACC);    \
    ACC = mfma16x4((WV).y, (XV).y, ACC);    \
    ACC = mfma16x4((WV).z, (XV).z, ACC);    \
    ACC = mfma16x4((WV).w, (XV).w, ACC)

__global__ __launch_bounds__(256) void k_lstm_fwd(LstmFwdArgs a) {
    const int w = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = lane & 15, g = lane >> 4;
    const bool bvalid = b < a.B;
    __shared__ f32x4 red[2][4][64];
    __shared__ float hs[16][4];

    const int nkbA = a.k_att / 16;                       // 96 (train) or 112 (inference)
    const float4* pa = a.packA + ((size_t)w * nkbA) * 64 + lane;
    const float4* pd = a.packD + ((size_t)w * (T2V_XW / 16)) * 64 + lane;
    const float* xrow = a.xs_prev + (size_t)(bvalid ? b : 0) * T2V_XW + 4 * g;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);

    f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accD = {0.f, 0.f, 0.f, 0.f};
    // shared-x region: k-blocks [0,96) = [h_att | ctx]; this wave owns 24 of them
    {
        const int kb0 = 24 * wave;
        if (a.do_att && a.do_dec) {
#pragma unroll 6
            for (int i = 0; i < 24; ++i) {
                const int kb = kb0 + i;
                const float4 x = bvalid ? *(const float4*)(xrow + 16 * kb) : z4;
                const float4 wa = pa[(size_t)kb * 64];
                const float4 wd = pd[(size_t)kb * 64];
                MFMA4(accA, wa, x);
                MFMA4(accD, wd, x);
            }
        } else if (a.do_att) {
#pragma unroll 8
            for (int i = 0; i < 24; ++i) {
                const int kb = kb0 + i;
                const float4 x = bvalid ? *(const float4*)(xrow + 16 * kb) : z4;
                const float4 wa = pa[(size_t)kb * 64];
                MFMA4(accA, wa, x);
            }
        } else {
#pragma unroll 8
            for (int i = 0; i < 24; ++i) {
                const int kb = kb0 + i;
                const float4 x = bvalid ? *(const float4*)(xrow + 16 * kb) : z4;
                const float4 wd = pd[(size_t)kb * 64];
                MFMA4(accD, wd, x);
            }
        }
    }
    if (a.do_dec) {   // decoder_rnn recurrent part: k-blocks [96,160) = h_dec
        const int kb0 = 96 + 16 * wave;
#pragma unroll 8
        for (int i = 0; i < 16; ++i) {
            const int kb = kb0 + i;
            const float4 x = bvalid ? *(const float4*)(xrow + 16 * kb) : z4;
            const float4 wd = pd[(size_t)kb * 64];
            MFMA4(accD, wd, x);
        }
    }
    if (a.do_att && a.pre_t) {   // inference: prenet columns are part of K (k-blocks [96,112))
        const float* prow = a.pre_t + (size_t)(bvalid ? b : 0) * T2V_PRE + 4 * g;
        const int kb0 = 96 + 4 * wave;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kb = kb0 + i;
            const float4 x = bvalid ? *(const float4*)(prow + 16 * (kb - 96)) : z4;
            const float4 wa = pa[(size_t)kb * 64];
            MFMA4(accA, wa, x);
        }
    }
    red[0][wave][lane] = accA;
    red[1][wave][lane] = accD;
    __syncthreads();

    // cell update: wave 0 -> attention_rnn(t), wave 1 -> decoder_rnn(t-1).
    // lane = (unit u = lane>>4, item b = lane&15); acc[r] = gate r (i,f,g,o) of unit 4w+u.
    if (wave < 2) {
        const int which = wave;
        const bool on = which == 0 ? a.do_att : a.do_dec;
        if (on && bvalid) {
            const f32x4 s = red[which][0][lane] + red[which][1][lane] + red[which][2][lane] + red[which][3][lane];
            const int U = 4 * w + g;
            const int tt = which == 0 ? a.t : a.t - 1;            // time index of this cell
            const float p = which == 0 ? a.p_att : a.p_dec;
            const uint32_t st_h = which == 0 ? T2V_RNG_ATT_H : T2V_RNG_DEC_H;
            const uint32_t st_c = which == 0 ? T2V_RNG_ATT_C : T2V_RNG_DEC_C;
            float pre[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float add;
                if (which == 0) add = a.gpre_t ? a.gpre_t[(size_t)b * T2V_G + r * T2V_H + U] : a.bias_att[r * T2V_H + U];
                else add = a.bias_dec[r * T2V_H + U];
                pre[r] = s[r] + add;
            }
            const float gi = sigmoidf_(pre[0]), gf = sigmoidf_(pre[1]), gg = tanhf(pre[2]), go = sigmoidf_(pre[3]);
            const float* cprev_p = which == 0 ? a.ca_prev : a.cd_prev;
            float* ccur_p = which == 0 ? a.ca_cur : a.cd_cur;
            const uint32_t idx = (uint32_t)b * T2V_H + U;
            float cprev = cprev_p[(size_t)b * T2V_H + U];
            if (tt > 0) cprev *= t2v_drop_scale(a.seed, st_c, tt - 1, idx, p);
            const float c = gf * cprev + gi * gg;
            const float h = go * tanhf(c);
            ccur_p[(size_t)b * T2V_H + U] = c;
            float* gsave = which == 0 ? a.ga_t : a.gd_t;
            if (gsave) {
                gsave[(size_t)b * T2V_G + 0 * T2V_H + U] = gi;
                gsave[(size_t)b * T2V_G + 1 * T2V_H + U] = gf;
                gsave[(size_t)b * T2V_G + 2 * T2V_H + U] = gg;
                gsave[(size_t)b * T2V_G + 3 * T2V_H + U] = go;
            }
            const float hd = h * t2v_drop_scale(a.seed, st_h, tt, idx, p);
            a.xs_next[(size_t)b * T2V_XW + (which == 0 ? U : T2V_KATT + U)] = hd;
            if (which == 0) hs[b][g] = hd;
        }
    }
    __syncthreads();
    if (a.do_att) {   // partial processed query of this workgroup's 4 hidden units
        for (int idx = tid; idx < a.B * T2V_A; idx += 256) {
            const int bb = idx >> 7, d = idx & (T2V_A - 1);
            const float* wq = a.wqT + (size_t)(4 * w) * T2V_A + d;
            const float q = wq[0] * hs[bb][0] + wq[T2V_A] * hs[bb][1] + wq[2 * T2V_A] * hs[bb][2] + wq[3 * T2V_A] * hs[bb][3];
            a.qp[((size_t)bb * T2V_NWG + w) * T2V_A + d] = q;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Attention step.  grid = (B, SE): every workgroup of an item recomputes the (cheap) energies
// and softmax; workgroup `se` produces context columns [se*512/SE, (se+1)*512/SE).
// LDS carve (floats): q[128] | ap[2][Tp+30] | cs[32][Tp] | e[Tp] | scr[512]
__global__ __launch_bounds__(256) void k_attn_fwd(AttnFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x, se = blockIdx.y, SE = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Tp = a.T_in, TpH = Tp + 30;
    float* q = smem;
    float* ap = q + T2V_A;
    float* cs = ap + 2 * TpH;
    float* e = cs + T2V_F * Tp;
    float* scr = e + ((Tp + 3) & ~3);
    const int len = a.lengths ? a.lengths[b] : Tp;

    // ---- 1. processed query = sum of the 256 per-workgroup partials (fixed order)
    {
        const int d = tid & (T2V_A - 1), hh = tid >> 7;
        const float* p = a.qp + ((size_t)b * T2V_NWG + 128 * hh) * T2V_A + d;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 8
        for (int i = 0; i < 128; i += 4) {
            s0 += p[(size_t)(i + 0) * T2V_A];
            s1 += p[(size_t)(i + 1) * T2V_A];
            s2 += p[(size_t)(i + 2) * T2V_A];
            s3 += p[(size_t)(i + 3) * T2V_A];
        }
        scr[tid] = (s0 + s1) + (s2 + s3);
    }
    // ---- previous / cumulative weights with a 15-wide zero halo
    for (int i = tid; i < 2 * TpH; i += 256) {
        const int ch = i / TpH, jj = i - ch * TpH - 15;
        float v = 0.f;
        if (jj >= 0 && jj < Tp) v = (ch == 0 ? a.al_prev : a.acum_prev)[(size_t)b * Tp + jj];
        ap[i] = v;
    }
    __syncthreads();
    if (tid < T2V_A) q[tid] = scr[tid] + scr[tid + 128];

    // ---- 2. location conv: cs[f][j] = sum_{ch,k} Wc[f][ch][k] * ap[ch][j+k]
    {
        const int f = tid & 31;
        float wc[2][T2V_KS];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int k = 0; k < T2V_KS; ++k) wc[ch][k] = a.loc_conv[(f * 2 + ch) * T2V_KS + k];
        const int njb = (Tp + 3) >> 2;
        for (int jb = tid >> 5; jb < njb; jb += 8) {
            const int j0 = 4 * jb;
            float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const float* apc = ap + ch * TpH + j0;
                float win[T2V_KS + 3];
#pragma unroll
                for (int k = 0; k < T2V_KS + 3; ++k) win[k] = (j0 + k < TpH) ? apc[k] : 0.f;
#pragma unroll
                for (int k = 0; k < T2V_KS; ++k) {
                    o0 = fmaf(wc[ch][k], win[k], o0);
                    o1 = fmaf(wc[ch][k], win[k + 1], o1);
                    o2 = fmaf(wc[ch][k], win[k + 2], o2);
                    o3 = fmaf(wc[ch][k], win[k + 3], o3);
                }
            }
            if (j0 + 0 < Tp) cs[f * Tp + j0 + 0] = o0;
            if (j0 + 1 < Tp) cs[f * Tp + j0 + 1] = o1;
            if (j0 + 2 < Tp) cs[f * Tp + j0 + 2] = o2;
            if (j0 + 3 < Tp) cs[f * Tp + j0 + 3] = o3;
        }
    }
    __syncthreads();
    if (se == 0 && a.conv_save) {
        float* dst = a.conv_save + (size_t)b * T2V_F * Tp;
        for (int i = tid; i < T2V_F * Tp; i += 256) dst[i] = cs[i];
    }

    // ---- 3. energies e[j] = sum_d v[d] * tanh(q[d] + pm[j][d] + sum_f D[d][f] cs[f][j])
    {
        const int d = tid & (T2V_A - 1), jh = tid >> 7;
        float dw[T2V_F];
#pragma unroll
        for (int f = 0; f < T2V_F; ++f) dw[f] = a.loc_dense[d * T2V_F + f];
        const float qd = q[d], vd = a.v[d];
        const float* pmb = a.pm + (size_t)b * Tp * T2V_A + d;
        float* ssave = (se == 0 && a.s_save) ? a.s_save + (size_t)b * Tp * T2V_A + d : nullptr;
        for (int j = jh; j < Tp; j += 2) {
            float acc = 0.f;
#pragma unroll
            for (int f = 0; f < T2V_F; ++f) acc = fmaf(dw[f], cs[f * Tp + j], acc);
            const float s = tanhf(qd + acc + pmb[(size_t)j * T2V_A]);
            if (ssave) ssave[(size_t)j * T2V_A] = s;
            const float part = wave_sum(vd * s);   // 64 of the 128 d's
            if (lane == 0) scr[256 + 2 * j + (wave & 1)] = part;
        }
    }
    __syncthreads();
    for (int j = tid; j < Tp; j += 256) {
        const float ev = scr[256 + 2 * j] + scr[256 + 2 * j + 1];
        e[j] = j < len ? ev : -INFINITY;
    }
    __syncthreads();

    // ---- 4. softmax over j (max-subtracted, masked -> exactly 0)
    float m = -INFINITY;
    for (int j = tid; j < Tp; j += 256) m = fmaxf(m, e[j]);
    m = wave_max(m);
    if (lane == 0) scr[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(scr[0], scr[1]), fmaxf(scr[2], scr[3]));
    __syncthreads();
    float sum = 0.f;
    for (int j = tid; j < Tp; j += 256) {
        const float ex = expf(e[j] - m);
        e[j] = ex;
        sum += ex;
    }
    sum = wave_sum(sum);
    if (lane == 0) scr[4 + wave] = sum;
    __syncthreads();
    sum = (scr[4] + scr[5]) + (scr[6] + scr[7]);
    const float inv = 1.0f / sum;
    __syncthreads();
    for (int j = tid; j < Tp; j += 256) {
        const float al = e[j] * inv;
        e[j] = al;
        if (se == 0) {
            a.al_cur[(size_t)b * Tp + j] = al;
            a.acum_cur[(size_t)b * Tp + j] = ap[TpH + 15 + j] + al;
        }
    }
    __syncthreads();

    // ---- 5. context slice: ctx[c] = sum_j alpha[j] * memory[b][j][c]
    {
        const int ES = T2V_E / SE;              // columns per workgroup
        const int parts = 256 / ES;             // j-partitions (>=1)
        const int c = tid % ES, part = tid / ES;
        float acc = 0.f;
        if (part < parts) {
            const float* mb = a.memory + (size_t)b * Tp * T2V_E + se * ES + c;
            for (int j = part; j < len; j += parts) acc = fmaf(e[j], mb[(size_t)j * T2V_E], acc);
        }
        scr[tid] = acc;
        __syncthreads();
        if (tid < ES) {
            float tot = 0.f;
            for (int pp = 0; pp < parts; ++pp) tot += scr[pp * ES + tid];
            a.xs_next[(size_t)b * T2V_XW + T2V_H + se * ES + tid] = tot;
        }
    }
}

// ------------------------------------------------------------------------------------------
extern "C" int t2v_pack_lstm_weights(const float* wcat_att, int k_att, const float* wcat_dec,
                                     float* packF_att, float* packF_dec, float* packB_att,
                                     float* packB_dec, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!wcat_att || !wcat_dec || !packF_att || !packF_dec) return T2V_ERR_ARG;
    if (k_att != T2V_KATT && k_att != T2V_KATT_INF) return T2V_ERR_DIMS;
    {
        const size_t n = (size_t)T2V_NWG * (k_att / 16) * 64;
        k_pack_fwd<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(wcat_att, k_att, (float4*)packF_att);
    }
    {
        const size_t n = (size_t)T2V_NWG * (T2V_XW / 16) * 64;
        k_pack_fwd<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(wcat_dec, T2V_XW, (float4*)packF_dec);
    }
    if (packB_att) {   // only the recurrent 1536 columns matter for the data gradient
        const size_t n = (size_t)(T2V_KATT / 16) * 256 * 64;
        k_pack_bwd<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(wcat_att, k_att, T2V_KATT, (float4*)packB_att);
    }
    if (packB_dec) {
        const size_t n = (size_t)(T2V_XW / 16) * 256 * 64;
        k_pack_bwd<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(wcat_dec, T2V_XW, T2V_XW, (float4*)packB_dec);
    }
    return t2v_check_launch();
}

size_t t2v_attn_fwd_lds(int T_in) {
    const int TpH = T_in + 30;
    return sizeof(float) * (T2V_A + 2 * TpH + T2V_F * T_in + ((T_in + 3) & ~3) + 256 + 2 * T_in + 8);
}

extern "C" int t2v_decoder_train_fwd(const t2v_dec_weights* w, const t2v_dec_train_bufs* s,
                                     int B, int T_in, int T_out, float p_att, float p_dec,
                                     uint64_t seed, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!w || !s || B < 1 || B > 16 || T_in < 1 || T_out < 1) return T2V_ERR_ARG;
    const size_t lds = t2v_attn_fwd_lds(T_in);
    if (lds > 160 * 1024) return T2V_ERR_ARG;
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void*)k_attn_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int SE = 4;
    for (int t = 0; t <= T_out; ++t) {
        LstmFwdArgs a;
        a.packA = (const float4*)w->packF_att;
        a.packD = (const float4*)w->packF_dec;
        a.k_att = T2V_KATT;
        a.xs_prev = s->XS + (size_t)t * B * T2V_XW;
        a.xs_next = s->XS + (size_t)(t + 1) * B * T2V_XW;
        a.gpre_t = t < T_out ? s->gpre + (size_t)t * B * T2V_G : nullptr;
        a.pre_t = nullptr;
        a.bias_att = w->bias_att;
        a.bias_dec = w->bias_dec;
        a.ca_prev = s->CA + (size_t)t * B * T2V_H;
        a.ca_cur = s->CA + (size_t)(t + 1) * B * T2V_H;
        a.cd_prev = t >= 1 ? s->CD + (size_t)(t - 1) * B * T2V_H : nullptr;
        a.cd_cur = t >= 1 ? s->CD + (size_t)t * B * T2V_H : nullptr;
        a.ga_t = t < T_out ? s->GA + (size_t)t * B * T2V_G : nullptr;
        a.gd_t = t >= 1 ? s->GD + (size_t)(t - 1) * B * T2V_G : nullptr;
        a.wqT = w->wqT;
        a.qp = s->QP;
        a.B = B;
        a.t = t;
        a.do_att = t < T_out;
        a.do_dec = t >= 1;
        a.p_att = p_att;
        a.p_dec = p_dec;
        a.seed = seed;
        k_lstm_fwd<<<T2V_NWG, 256, 0, stream>>>(a);
        if (t < T_out) {
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
            f.s_save = s->S ? s->S + (size_t)t * B * T_in * T2V_A : nullptr;
            f.conv_save = s->CONV ? s->CONV + (size_t)t * B * T2V_F * T_in : nullptr;
            f.T_in = T_in;
            k_attn_fwd<<<dim3(B, SE), 256, lds, stream>>>(f);
        }
    }
    return t2v_check_launch();
}
