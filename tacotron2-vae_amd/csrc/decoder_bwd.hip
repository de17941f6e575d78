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

#define LSTM_WAVES 16
__global__ __launch_bounds__(1024) void k_lstm_bwd(LstmBwdArgs a) {
    const int w = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = lane & 15, g = lane >> 4;
    const bool dec = w < T2V_XW / 16;
    const int wt = dec ? w : w - T2V_XW / 16;
    const float* kv = dec ? a.dgd_t : a.dga_n;
    if (!kv) return;   // block-uniform
    const bool bvalid = b < a.B;
    __shared__ f32x4 red[LSTM_WAVES][64];
    const int ntile = dec ? T2V_XW / 16 : T2V_KATT / 16;
    const float4* p = (dec ? a.packBD : a.packBA) + (size_t)wt * 64 + lane;   // + kb * ntile * 64
    const float* xrow = kv + (size_t)(bvalid ? b : 0) * T2V_G + 4 * g;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int kb0 = 16 * wave;     // 256 k-blocks / 16 waves
    // successive launches walk the k-blocks in opposite directions (a.flip): whatever part of the
    // 67 MB stream the previous launch left in the XCD L2s is requested first, before it is evicted
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float4 wv[8], xv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int kb = a.flip ? kb0 + 15 - (8 * h + i) : kb0 + 8 * h + i;
            wv[i] = p[(size_t)kb * ntile * 64];
            xv[i] = *(const float4*)(xrow + 16 * kb);   // lanes b>=B read row 0 (unused D columns)
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { MFMA4(acc, wv[i], xv[i]); }
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && bvalid) {
        f32x4 s = red[0][lane];
#pragma unroll
        for (int i = 1; i < LSTM_WAVES; ++i) s += red[i][lane];
        float* y = dec ? a.YD + (size_t)b * T2V_XW : a.YA + (size_t)b * T2V_KATT;
        *(float4*)(y + 16 * wt + 4 * g) = make_float4(s[0], s[1], s[2], s[3]);
    }
}

// attention(t) backward.  grid = B, 512 threads, JP = ceil(T_in/4) register rows.
// LDS carve (floats): dctx[512] | alpha[TpR] | dal[TpR] | de[TpR] | dpT[128*TpP] | dcs[32*DS] | wcl[32*63] | scr[1024]
//   dpT = dpre transposed [d][TpP] (TpP odd -> conflict-free column writes, MFMA B reads along j)
#define ATB_THREADS 512
#define ATB_R (ATB_THREADS / 128)
static __host__ __device__ inline int attn_bwd_ds(int Tp) { return (16 * ((Tp + 15) / 16) + 46) | 1; }
static __host__ __device__ inline int attn_bwd_tpp(int Tp) { return (16 * ((Tp + 15) / 16)) | 1; }
size_t t2v_attn_bwd_lds(int Tp) {
    const int TpR = (Tp + 3) & ~3;
    return sizeof(float) * (T2V_E + 3 * TpR + (size_t)T2V_A * attn_bwd_tpp(Tp) + T2V_F * attn_bwd_ds(Tp) + T2V_F * 63 + 1024);
}

template <int JP>
__global__ __launch_bounds__(ATB_THREADS) void k_attn_bwd(AttnBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Tp = a.T_in, TpR = (Tp + 3) & ~3, DS = attn_bwd_ds(Tp), TpP = attn_bwd_tpp(Tp);
    float* dctx = smem;
    float* alpha = dctx + T2V_E;
    float* dal = alpha + TpR;
    float* de = dal + TpR;
    float* dpT = de + TpR;
    float* dcs = dpT + (size_t)T2V_A * TpP;
    float* wcl = dcs + T2V_F * DS;
    float* scr = wcl + T2V_F * 63;
    const int d = tid & (T2V_A - 1), j4 = tid >> 7;
    const int g = lane >> 4, c16 = lane & 15;

    T2V_STAMP(a, 0);
    // ---- entry: issue the global reads
    float sreg[JP];
    {
        const float* sp = a.S_t + (size_t)b * Tp * T2V_A + d;
#pragma unroll
        for (int i = 0; i < JP; ++i) {
            const int j = j4 + ATB_R * i;
            sreg[i] = j < Tp ? sp[(size_t)j * T2V_A] : 0.f;
        }
    }
    const float vd = a.v[d];
    // 1. total gradient of the context of step t
    for (int e = tid; e < T2V_E; e += ATB_THREADS) {
        const float v = a.dHC_t[(size_t)b * (T2V_H + T2V_E) + T2V_H + e] + a.YD[(size_t)b * T2V_XW + T2V_H + e] +
                        a.YA[(size_t)b * T2V_KATT + T2V_H + e];
        dctx[e] = v;
        a.DCTX_t[(size_t)b * T2V_E + e] = v;
    }
    for (int j = tid; j < Tp; j += ATB_THREADS) {
        alpha[j] = a.al_cur[(size_t)b * Tp + j];
        dal[j] = a.GPREV[(size_t)b * Tp + j] + a.GCUM[(size_t)b * Tp + j];
    }
    for (int i = tid; i < T2V_F * DS; i += ATB_THREADS) dcs[i] = 0.f;
    if (tid < T2V_F * 62 / 4) {
        const float4 w4 = ((const float4*)a.loc_conv)[tid];
        const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) { const int i = 4 * tid + c; wcl[(i / 62) * 63 + (i % 62)] = wv[c]; }
    }
    __syncthreads();

    T2V_STAMP(a, 1);
    // 2. d alpha[j] += dctx . memory[j]   (wave per row, 8 rows in flight per wave)
    {
        const float4 d0 = *(const float4*)(dctx + lane * 4), d1 = *(const float4*)(dctx + 256 + lane * 4);
        constexpr int NW = ATB_THREADS / 64, NB = 8;
        for (int base = wave; base < Tp; base += NW * NB) {
            float4 m0[NB], m1[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                const int j = base + NW * r;
                const float* mrow = a.memory + ((size_t)b * Tp + (j < Tp ? j : 0)) * T2V_E + lane * 4;
                m0[r] = *(const float4*)mrow;
                m1[r] = *(const float4*)(mrow + 256);
            }
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                const int j = base + NW * r;
                float acc = m0[r].x * d0.x;
                acc = fmaf(m0[r].y, d0.y, acc); acc = fmaf(m0[r].z, d0.z, acc); acc = fmaf(m0[r].w, d0.w, acc);
                acc = fmaf(m1[r].x, d1.x, acc); acc = fmaf(m1[r].y, d1.y, acc);
                acc = fmaf(m1[r].z, d1.z, acc); acc = fmaf(m1[r].w, d1.w, acc);
                acc = wave_sum(acc);
                if (lane == 0 && j < Tp) dal[j] += acc;
            }
        }
    }
    __syncthreads();
    // location_dense as the MFMA A operand of phase 5: A[f = f0+c16][k = d = 4st+g]
    float areg[32];
    {
        const int f0 = 16 * (wave & 1);
#pragma unroll
        for (int st = 0; st < 32; ++st) areg[st] = a.loc_dense[(4 * st + g) * T2V_F + f0 + c16];
    }

    T2V_STAMP(a, 2);
    // 3. softmax backward (every wave reduces redundantly)
    {
        float part = 0.f;
        for (int j = lane; j < Tp; j += 64) part = fmaf(alpha[j], dal[j], part);
        const float dot = wave_sum(part);
        for (int j = tid; j < Tp; j += ATB_THREADS) de[j] = alpha[j] * (dal[j] - dot);
    }
    __syncthreads();

    T2V_STAMP(a, 3);
    // 4. through v . tanh(.)
    {
        float* sp = a.S_t + (size_t)b * Tp * T2V_A + d;
        float dq = 0.f, dv = 0.f;
#pragma unroll
        for (int i = 0; i < JP; ++i) {
            const int j = j4 + ATB_R * i;
            if (j < Tp) {
                const float s = sreg[i];
                const float dej = de[j];
                const float dp = dej * vd * (1.0f - s * s);
                dpT[d * TpP + j] = dp;
                sp[(size_t)j * T2V_A] = dp;
                dq += dp;
                dv = fmaf(dej, s, dv);
            }
        }
        // columns j in [Tp, 16*ceil(Tp/16)) feed discarded MFMA columns: keep them finite
        for (int j = Tp + j4; j < TpP - 1; j += ATB_R) dpT[d * TpP + j] = 0.f;
        scr[tid] = dq;
        scr[ATB_THREADS + tid] = dv;
        __syncthreads();
        if (tid < T2V_A) {
            float q = 0.f, v = 0.f;
#pragma unroll
            for (int i = 0; i < ATB_R; ++i) { q += scr[i * T2V_A + tid]; v += scr[ATB_THREADS + i * T2V_A + tid]; }
            a.DQ_t[(size_t)b * T2V_A + tid] = q;
            a.DV[(size_t)b * T2V_A + tid] += v;
        }
    }

    T2V_STAMP(a, 4);
    // 5. through location_dense on MFMA: dc[f][j] = sum_d D[d][f] dpre[j][d]; tile = 16 f x 16 j, K = 128
    {
        const int f0 = 16 * (wave & 1);
        const int NJ = (Tp + 15) >> 4;
        for (int jt = wave >> 1; jt < NJ; jt += ATB_THREADS / 128) {
            const int j = 16 * jt + c16;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 32; ++st) acc = mfma16x4(areg[st], dpT[(4 * st + g) * TpP + j], acc);
            if (j < Tp) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = f0 + 4 * g + r;
                    dcs[f * DS + 15 + j] = acc[r];
                    a.DC_t[((size_t)b * T2V_F + f) * Tp + j] = acc[r];
                }
            }
        }
    }
    __syncthreads();

    T2V_STAMP(a, 5);
    // 6. through location_conv (transposed): dcat[ch][j] = sum_{f,k} Wc[f][ch][k] dc[f][j+15-k]
    //    thread = (f, block of 6 consecutive j): sliding window in registers, then a 32-lane sum over f
    {
        const int f = tid & 31;
        float wc[2][T2V_KS];
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
#pragma unroll
            for (int k = 0; k < T2V_KS; ++k) wc[ch][k] = wcl[f * 63 + ch * T2V_KS + k];
        for (int j0 = 6 * (tid >> 5); j0 < Tp; j0 += 6 * (ATB_THREADS / 32)) {
            float win[36];
            const float* row = dcs + f * DS + j0;      // dcs index of dc[f][jj] is 15 + jj
#pragma unroll
            for (int i = 0; i < 36; ++i) win[i] = row[i];   // dc[f][j0-15 .. j0+20]
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < T2V_KS; ++k)
#pragma unroll
                    for (int jj = 0; jj < 6; ++jj) acc[jj] = fmaf(wc[ch][k], win[jj + 30 - k], acc[jj]);
#pragma unroll
                for (int jj = 0; jj < 6; ++jj) {
                    float v = row16_sum(acc[jj]);
                    v += __shfl_xor(v, 16, 64);
                    const int j = j0 + jj;
                    if (f == 0 && j < Tp) {
                        if (ch == 0) a.GPREV[(size_t)b * Tp + j] = v;
                        else a.GCUM[(size_t)b * Tp + j] += v;
                    }
                }
            }
        }
    }
    T2V_STAMP(a, 6);
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
#pragma unroll 16
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
        const float tc = tanhf_(a.CA_cur[bu]);
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
        const float tc = tanhf_(a.CD_cur[bu]);
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
    if (T_in > 256) return T2V_ERR_ARG;
#define ATB_LAUNCH(JPV)                                                                                   \
    do {                                                                                                  \
        if (lds > 64 * 1024)                                                                              \
            (void)hipFuncSetAttribute((const void*)k_attn_bwd<JPV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        k_attn_bwd<JPV><<<B, ATB_THREADS, lds, stream>>>(f);                                              \
    } while (0)
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
            l.flip = t & 1;
            k_lstm_bwd<<<T2V_NWG, 1024, 0, stream>>>(l);

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
            f.prof = g_t2v_prof ? g_t2v_prof + 16 : nullptr;
            if (T_in <= 22 * ATB_R) ATB_LAUNCH(22);
            else if (T_in <= 32 * ATB_R) ATB_LAUNCH(32);
            else ATB_LAUNCH(64);
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
