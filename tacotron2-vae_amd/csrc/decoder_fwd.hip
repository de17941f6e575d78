// Decoder recurrence, forward: weight packing + the two per-step kernels.
//   k_lstm_fwd : attention_rnn(t) and decoder_rnn(t-1) gate GEMVs on MFMA (weights streamed in
//                fragment order), LSTM cell update, state dropout, partial query projection.
//   k_attn_fwd : location-sensitive attention of step t (query sum, location conv + dense,
//                tanh/v energies, masked softmax, context).
// Reference semantics: Decoder.decode model.py:346-389, Attention.forward model.py:67-88.
#include "t2v_common.h"
#include "t2v_kernels.h"
#include <stdlib.h>

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
    const int kb = (idx >> 6) % nkb;         // tile-major: a workgroup streams one contiguous
    const int w = (idx >> 6) / nkb;          // 96/112/160 KiB region (tile stride is NOT a power of
    const int arow = lane & 15, g = lane >> 4;   // two, so L2/HBM channels are evenly loaded)
    const int row = (arow & 3) * T2V_H + 4 * w + (arow >> 2);
    const float* src = W + (size_t)row * K + 16 * kb + 4 * g;
    P[idx] = make_float4(src[0], src[1], src[2], src[3]);
}
// Backward (transposed) tile n-tile w' of W^T (N = K columns of W become rows), reduction dim
// = 4096 gate rows, tile-major like the forward pack (a workgroup's 256 KB are contiguous):
//   PB[w'][kb][lane][i] = W[16kb + 4(lane>>4) + i][16w' + (lane&15)]
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

// 16 waves per workgroup: the K dimension of both gate GEMVs is split 16 ways so that every
// wave issues ALL of its weight/x loads (26 x 1 KiB) before the first MFMA — the kernel is a
// pure weight stream (262 KB per CU per launch) and needs the bytes in flight, not occupancy.
#define LSTM_WAVES 16
template <int MODE>   // 0: both cells (skewed); 1: attention_rnn only, prenet columns in K (inference);
                      // 2: decoder_rnn only; 3: attention_rnn only (training, hoisted prenet term)
__global__ __launch_bounds__(1024) void k_lstm_fwd(LstmFwdArgs a) {
    constexpr bool INFER = MODE == 1;
    constexpr bool DO_A = MODE != 2, DO_D = MODE == 0 || MODE == 2;
    const int w = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = lane & 15, g = lane >> 4;
    const bool bvalid = b < a.B;
    __shared__ f32x4 red[2][LSTM_WAVES][64];
    __shared__ float hs[16][4];

        const float4* pa = a.packA + ((size_t)w * (a.k_att / 16)) * 64 + lane;
    const float4* pd = a.packD + ((size_t)w * (T2V_XW / 16)) * 64 + lane;
    const float* xrow = a.xs_prev + (size_t)(bvalid ? b : 0) * T2V_XW + 4 * g;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);

    // ---- tail operands are fetched FIRST so the serial epilogue never waits on global memory:
    // waves 0/1 own the cell update of attention_rnn(t) / decoder_rnn(t-1) for (unit g, item b)
    const int which = wave;                               // meaningful for wave < 2 only
    const bool cell_on = wave < 2 && bvalid && (which == 0 ? a.do_att : a.do_dec);
    const int U = 4 * w + g;
    float addv[4] = {0.f, 0.f, 0.f, 0.f}, cprev = 0.f;
    if (cell_on) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (which == 0) addv[r] = a.gpre_t ? a.gpre_t[(size_t)b * T2V_G + r * T2V_H + U] : a.bias_att[r * T2V_H + U];
            else addv[r] = a.bias_dec[r * T2V_H + U];
        }
        cprev = (which == 0 ? a.ca_prev : a.cd_prev)[(size_t)b * T2V_H + U];
    }

    f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accD = {0.f, 0.f, 0.f, 0.f};
    // shared-x region: k-blocks [0,96) = [h_att | ctx] (6 per wave); decoder_rnn recurrent part:
    // k-blocks [96,160) = h_dec (4 per wave)
    float4 xs[6], wa[6], wd[6], xr[4], wr[4];
    {
        // odd steps walk this wave's k-blocks backwards: the tail of the previous launch's weight
        // stream is still in the XCD L2 and gets requested first (measured -13 % on k_lstm_bwd)
        const bool flip = a.t & 1;
        const int kb0 = 6 * wave, kr0 = 96 + 4 * wave;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int kb = kb0 + (flip ? 5 - i : i);
            xs[i] = *(const float4*)(xrow + 16 * kb);   // lanes b>=B read row 0: their D columns are never used
            if (DO_A) wa[i] = pa[(size_t)kb * 64];   // training: both cells are always computed,
            if (DO_D) wd[i] = pd[(size_t)kb * 64];   // do_att/do_dec only gate the cell update (t=0 / t=T)
        }
        if (DO_D) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kb = kr0 + (flip ? 3 - i : i);
                xr[i] = *(const float4*)(xrow + 16 * kb);
                wr[i] = pd[(size_t)kb * 64];
            }
        }
    }
    float4 xp = z4, wp = z4;
    if (INFER) {   // inference: prenet columns are part of K (k-blocks [96,112))
        const float* prow = a.pre_t + (size_t)(bvalid ? b : 0) * T2V_PRE + 4 * g;
        xp = *(const float4*)(prow + 16 * wave);
        wp = pa[(size_t)(96 + wave) * 64];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        if (DO_A) { MFMA4(accA, wa[i], xs[i]); }
        if (DO_D) { MFMA4(accD, wd[i], xs[i]); }
    }
    if (INFER) { MFMA4(accA, wp, xp); }
    if (DO_D) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { MFMA4(accD, wr[i], xr[i]); }
    }
    red[0][wave][lane] = accA;
    red[1][wave][lane] = accD;
    // query partial: thread (d = tid&127) handles items bb = tid>>7, tid>>7 + 8; its 4 query-weight values are
    // fetched here so the latency hides under the reduction + cell update
    float wqr[4] = {0.f, 0.f, 0.f, 0.f};
    const bool q_on = a.do_att && (tid >> 7) < a.B;
    if (q_on) {
        const float* wq = a.wqT + (size_t)(4 * w) * T2V_A + (tid & (T2V_A - 1));
        wqr[0] = wq[0]; wqr[1] = wq[T2V_A]; wqr[2] = wq[2 * T2V_A]; wqr[3] = wq[3 * T2V_A];
    }
    __syncthreads();

    // cell update: wave 0 -> attention_rnn(t), wave 1 -> decoder_rnn(t-1).
    // lane = (unit u = lane>>4, item b = lane&15); acc[r] = gate r (i,f,g,o) of unit 4w+u.
    if (cell_on) {
        f32x4 s = red[which][0][lane];
#pragma unroll
        for (int i = 1; i < LSTM_WAVES; ++i) s += red[which][i][lane];
        const int tt = which == 0 ? a.t : a.t - 1;            // time index of this cell
        const float p = which == 0 ? a.p_att : a.p_dec;
        const uint32_t st_h = which == 0 ? T2V_RNG_ATT_H : T2V_RNG_DEC_H;
        const uint32_t st_c = which == 0 ? T2V_RNG_ATT_C : T2V_RNG_DEC_C;
        const float gi = sigmoidf_(s[0] + addv[0]), gf = sigmoidf_(s[1] + addv[1]);
        const float gg = tanhf_(s[2] + addv[2]), go = sigmoidf_(s[3] + addv[3]);
        float* ccur_p = which == 0 ? a.ca_cur : a.cd_cur;
        const uint32_t idx = (uint32_t)b * T2V_H + U;
        if (tt > 0) cprev *= t2v_drop_scale(a.seed, st_c, tt - 1, idx, p);
        const float c = gf * cprev + gi * gg;
        const float h = go * tanhf_(c);
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
    __syncthreads();
    if (q_on) {   // partial processed query of this workgroup's 4 hidden units
        const int d = tid & (T2V_A - 1);
        for (int bb = tid >> 7; bb < a.B; bb += 8) {
            const float q = wqr[0] * hs[bb][0] + wqr[1] * hs[bb][1] + wqr[2] * hs[bb][2] + wqr[3] * hs[bb][3];
            a.qp[((size_t)bb * T2V_NWG + w) * T2V_A + d] = q;
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

#define ATT_THREADS_HOST 512
// ---- side stream for the overlapped schedule (created lazily, one per host thread)
struct T2vSide { hipStream_t stream = nullptr; hipEvent_t ev[8]; hipEvent_t join; bool ok = false; };
static thread_local T2vSide g_side;
static bool side_ready() {
    if (g_side.ok) return true;
    if (hipStreamCreateWithFlags(&g_side.stream, hipStreamNonBlocking) != hipSuccess) return false;
    for (int i = 0; i < 8; ++i)
        if (hipEventCreateWithFlags(&g_side.ev[i], hipEventDisableTiming) != hipSuccess) return false;
    if (hipEventCreateWithFlags(&g_side.join, hipEventDisableTiming) != hipSuccess) return false;
    g_side.ok = true;
    return true;
}
extern "C" int t2v_overlap_enabled(void) {
    static int v = -1;
    if (v < 0) { const char* e = getenv("T2V_OVERLAP"); v = (e && e[0] == '1') ? 1 : 0; }
    return v;
}

// Training-step version of k_lstm_fwd<0> with FOUR waves per workgroup (tools/ubench_gemv_tiles.hip: for this 67 MB
// stream about 64 KB in flight per CU is the sweet spot — 16 waves that each put 26 KB in flight queue too much).
// Workgroup w = gate-row tile w of BOTH cells; wave v walks attention_rnn k-blocks [24v, 24v+24) then decoder_rnn
// k-blocks [40v, 40v+40): 8 rounds of 8 (W, x) float4 pairs, two rounds in flight; odd steps walk the same
// sequence backwards (the tail of the previous launch's stream is requested first).  Epilogue as k_lstm_fwd.
template <bool FLIP>
__device__ __forceinline__ void lstm256_stream(const float4* pa, const float4* pd, const float* xrow, int wave,
                                               f32x4& accA, f32x4& accD) {
    float4 wv[2][8], xv[2][8];
    // round r of the walk: FLIP ? 7 - r : r;  rounds 0..2 = attention_rnn, 3..7 = decoder_rnn
#define L256_LOAD(R)                                                                              \
    {                                                                                             \
        constexpr int RR = FLIP ? 7 - (R) : (R);                                                  \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                           \
            const int ii = FLIP ? 7 - i : i;                                                      \
            const int kb = RR < 3 ? 24 * wave + 8 * RR + ii : 40 * wave + 8 * (RR - 3) + ii;      \
            wv[(R) & 1][i] = (RR < 3 ? pa : pd)[(size_t)kb * 64];                                 \
            xv[(R) & 1][i] = *(const float4*)(xrow + 16 * kb);                                    \
        }                                                                                         \
    }
#define L256_MATH(R)                                                                              \
    {                                                                                             \
        constexpr int RR = FLIP ? 7 - (R) : (R);                                                  \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                           \
            if (RR < 3) { MFMA4(accA, wv[(R) & 1][i], xv[(R) & 1][i]); }                          \
            else { MFMA4(accD, wv[(R) & 1][i], xv[(R) & 1][i]); }                                 \
        }                                                                                         \
    }
#define L256_STEP(R, NEXT)                      \
    NEXT                                        \
    __builtin_amdgcn_sched_barrier(0);          \
    L256_MATH(R)                                \
    __builtin_amdgcn_sched_barrier(0);
    L256_LOAD(0)
    __builtin_amdgcn_sched_barrier(0);
    L256_STEP(0, L256_LOAD(1))
    L256_STEP(1, L256_LOAD(2))
    L256_STEP(2, L256_LOAD(3))
    L256_STEP(3, L256_LOAD(4))
    L256_STEP(4, L256_LOAD(5))
    L256_STEP(5, L256_LOAD(6))
    L256_STEP(6, L256_LOAD(7))
    L256_STEP(7, )
#undef L256_STEP
#undef L256_MATH
#undef L256_LOAD
}

__global__ __launch_bounds__(256) void k_lstm_fwd256(LstmFwdArgs a) {
    const int w = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = lane & 15, g = lane >> 4;
    const bool bvalid = b < a.B;
    __shared__ f32x4 red[2][4][64];
    __shared__ float hs[16][4];
    const float4* pa = a.packA + ((size_t)w * (T2V_KATT / 16)) * 64 + lane;
    const float4* pd = a.packD + ((size_t)w * (T2V_XW / 16)) * 64 + lane;
    const float* xrow = a.xs_prev + (size_t)(bvalid ? b : 0) * T2V_XW + 4 * g;   // lanes b>=B read row 0 (unused D columns)
    // tail operands first: waves 0/1 own the cell update of attention_rnn(t) / decoder_rnn(t-1) for (unit g, item b)
    const int which = wave;
    const bool cell_on = wave < 2 && bvalid && (which == 0 ? a.do_att : a.do_dec);
    const int U = 4 * w + g;
    float addv[4] = {0.f, 0.f, 0.f, 0.f}, cprev = 0.f;
    if (cell_on) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            addv[r] = which == 0 ? a.gpre_t[(size_t)b * T2V_G + r * T2V_H + U] : a.bias_dec[r * T2V_H + U];
        cprev = (which == 0 ? a.ca_prev : a.cd_prev)[(size_t)b * T2V_H + U];
    }
    float wqr[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.do_att) {
        const float* wq = a.wqT + (size_t)(4 * w) * T2V_A + (tid & (T2V_A - 1));
        wqr[0] = wq[0]; wqr[1] = wq[T2V_A]; wqr[2] = wq[2 * T2V_A]; wqr[3] = wq[3 * T2V_A];
    }
    f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accD = {0.f, 0.f, 0.f, 0.f};
    if (a.t & 1) lstm256_stream<true>(pa, pd, xrow, wave, accA, accD);
    else lstm256_stream<false>(pa, pd, xrow, wave, accA, accD);
    red[0][wave][lane] = accA;
    red[1][wave][lane] = accD;
    __syncthreads();
    if (cell_on) {
        const f32x4 s = (red[which][0][lane] + red[which][1][lane]) + (red[which][2][lane] + red[which][3][lane]);
        const int tt = which == 0 ? a.t : a.t - 1;
        const float p = which == 0 ? a.p_att : a.p_dec;
        const uint32_t st_h = which == 0 ? T2V_RNG_ATT_H : T2V_RNG_DEC_H;
        const uint32_t st_c = which == 0 ? T2V_RNG_ATT_C : T2V_RNG_DEC_C;
        const float gi = sigmoidf_(s[0] + addv[0]), gf = sigmoidf_(s[1] + addv[1]);
        const float gg = tanhf_(s[2] + addv[2]), go = sigmoidf_(s[3] + addv[3]);
        const uint32_t idx = (uint32_t)b * T2V_H + U;
        if (tt > 0) cprev *= t2v_drop_scale(a.seed, st_c, tt - 1, idx, p);
        const float c = gf * cprev + gi * gg;
        const float h = go * tanhf_(c);
        (which == 0 ? a.ca_cur : a.cd_cur)[(size_t)b * T2V_H + U] = c;
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
    __syncthreads();
    if (a.do_att) {   // partial processed query of this workgroup's 4 hidden units
        const int d = tid & (T2V_A - 1);
        for (int bb = tid >> 7; bb < a.B; bb += 2) {
            const float q = wqr[0] * hs[bb][0] + wqr[1] * hs[bb][1] + wqr[2] * hs[bb][2] + wqr[3] * hs[bb][3];
            a.qp[((size_t)bb * T2V_NWG + w) * T2V_A + d] = q;
        }
    }
}

// Single-cell four-wave kernels for the free-running decode loop (decoder_infer.hip), same stream shape as
// k_lstm_fwd256.  ATT = true : attention_rnn(t) with the prenet columns inside K (112 k-blocks: 28 per wave, 4
// rounds of 7); ATT = false: decoder_rnn (160 k-blocks: 40 per wave, 5 rounds of 8).
template <bool ATT>
__global__ __launch_bounds__(256) void k_lstm_one256(LstmFwdArgs a) {
    constexpr int RN = ATT ? 7 : 8, NR = ATT ? 4 : 5, KBW = RN * NR;
    const int w = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = lane & 15, g = lane >> 4;
    const bool bvalid = b < a.B;
    __shared__ f32x4 red[4][64];
    __shared__ float hs[16][4];
    const float4* pw = ATT ? a.packA + ((size_t)w * (T2V_KATT_INF / 16)) * 64 + lane
                           : a.packD + ((size_t)w * (T2V_XW / 16)) * 64 + lane;
    const float* xrow = a.xs_prev + (size_t)(bvalid ? b : 0) * T2V_XW + 4 * g;
    const float* prow = ATT ? a.pre_t + (size_t)(bvalid ? b : 0) * T2V_PRE + 4 * g : nullptr;
    const bool cell_on = wave == 0 && bvalid;
    const int U = 4 * w + g;
    float addv[4] = {0.f, 0.f, 0.f, 0.f}, cprev = 0.f;
    if (cell_on) {
#pragma unroll
        for (int r = 0; r < 4; ++r) addv[r] = (ATT ? a.bias_att : a.bias_dec)[r * T2V_H + U];
        cprev = (ATT ? a.ca_prev : a.cd_prev)[(size_t)b * T2V_H + U];
    }
    float wqr[4] = {0.f, 0.f, 0.f, 0.f};
    if (ATT) {
        const float* wq = a.wqT + (size_t)(4 * w) * T2V_A + (tid & (T2V_A - 1));
        wqr[0] = wq[0]; wqr[1] = wq[T2V_A]; wqr[2] = wq[2 * T2V_A]; wqr[3] = wq[3 * T2V_A];
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int kb0 = KBW * wave;
    const bool flip = a.t & 1;
    float4 wv[2][RN], xv[2][RN];
#define ONE_LOAD(H)                                                                               \
    _Pragma("unroll") for (int i = 0; i < RN; ++i) {                                              \
        const int kk = RN * (H) + i;                                                              \
        const int kb = kb0 + (flip ? KBW - 1 - kk : kk);                                          \
        wv[(H) & 1][i] = pw[(size_t)kb * 64];                                                     \
        const float* src = (ATT && kb >= T2V_KATT / 16) ? prow + 16 * (kb - T2V_KATT / 16) : xrow + 16 * kb; \
        xv[(H) & 1][i] = *(const float4*)src;                                                     \
    }
    ONE_LOAD(0)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < NR; ++h) {
        if (h + 1 < NR) { ONE_LOAD(h + 1) }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < RN; ++i) { MFMA4(acc, wv[h & 1][i], xv[h & 1][i]); }
        __builtin_amdgcn_sched_barrier(0);
    }
#undef ONE_LOAD
    red[wave][lane] = acc;
    __syncthreads();
    if (cell_on) {
        const f32x4 s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        const int tt = ATT ? a.t : a.t - 1;
        const float p = ATT ? a.p_att : a.p_dec;
        const float gi = sigmoidf_(s[0] + addv[0]), gf = sigmoidf_(s[1] + addv[1]);
        const float gg = tanhf_(s[2] + addv[2]), go = sigmoidf_(s[3] + addv[3]);
        const uint32_t idx = (uint32_t)b * T2V_H + U;
        if (tt > 0) cprev *= t2v_drop_scale(a.seed, ATT ? T2V_RNG_ATT_C : T2V_RNG_DEC_C, tt - 1, idx, p);
        const float c = gf * cprev + gi * gg;
        const float h = go * tanhf_(c);
        (ATT ? a.ca_cur : a.cd_cur)[(size_t)b * T2V_H + U] = c;
        const float hd = h * t2v_drop_scale(a.seed, ATT ? T2V_RNG_ATT_H : T2V_RNG_DEC_H, tt, idx, p);
        a.xs_next[(size_t)b * T2V_XW + (ATT ? U : T2V_KATT + U)] = hd;
        if (ATT) hs[b][g] = hd;
    }
    if (ATT) {
        __syncthreads();
        const int d = tid & (T2V_A - 1);
        for (int bb = tid >> 7; bb < a.B; bb += 2) {
            const float q = wqr[0] * hs[bb][0] + wqr[1] * hs[bb][1] + wqr[2] * hs[bb][2] + wqr[3] * hs[bb][3];
            a.qp[((size_t)bb * T2V_NWG + w) * T2V_A + d] = q;
        }
    }
}

static void fill_lstm_args(LstmFwdArgs& a, const t2v_dec_weights* w, const t2v_dec_train_bufs* s, int B, int T_out,
                           int t, float p_att, float p_dec, uint64_t seed) {
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
}

// mask bits: 1 = fused k_lstm_fwd<0> (serial schedule), 2 = k_attn_fwd, 4 = k_lstm_fwd<2> (decoder_rnn only),
// 8 = k_lstm_fwd<3> (attention_rnn only); 16 = run the overlapped two-stream schedule
static int launch_train_fwd(const t2v_dec_weights* w, const t2v_dec_train_bufs* s, int B, int T_in, int T_out,
                            float p_att, float p_dec, uint64_t seed, void* stream_, int mask) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!w || !s || B < 1 || B > 16 || T_in < 1 || T_out < 1) return T2V_ERR_ARG;
    if (T_in > 256) return T2V_ERR_ARG;
    // scratch tail of QP: [0,32768) partial-energy exchange, then 32 uint32 arrival counters / error word
    float* qp_tail = s->QP + (size_t)B * T2V_NWG * T2V_A;
    (void)hipMemsetAsync(qp_tail + 32768, 0, 32 * sizeof(uint32_t), stream);
    const bool overlap = (mask & 16) && side_ready();
    hipStream_t sb = overlap ? g_side.stream : stream;
    for (int t = 0; t <= T_out; ++t) {
        LstmFwdArgs a;
        fill_lstm_args(a, w, s, B, T_out, t, p_att, p_dec, seed);
        if (overlap) {
            // stream A: attention_rnn(t) -> attention(t); stream B: decoder_rnn(t-1) once attention_rnn(t) is done,
            // so its 42 MB weight stream runs underneath the latency-bound attention kernel
            LstmFwdArgs aa = a, ad = a;
            aa.do_dec = 0;      // each single-cell kernel must not touch the other cell's state
            ad.do_att = 0;
            if (t < T_out) {
                k_lstm_fwd<3><<<T2V_NWG, 1024, 0, stream>>>(aa);
                (void)hipEventRecord(g_side.ev[t & 7], stream);
            } else {
                (void)hipEventRecord(g_side.ev[t & 7], stream);     // after attention(T-1)
            }
            if (t >= 1) {
                (void)hipStreamWaitEvent(sb, g_side.ev[t & 7], 0);
                k_lstm_fwd<2><<<T2V_NWG, 1024, 0, sb>>>(ad);
            }
        } else {
            if (mask & 1) k_lstm_fwd256<<<T2V_NWG, 256, 0, stream>>>(a);
            LstmFwdArgs aa = a, ad = a;
            aa.do_dec = 0;
            ad.do_att = 0;
            if ((mask & 8) && t < T_out) k_lstm_fwd<3><<<T2V_NWG, 1024, 0, stream>>>(aa);
            if ((mask & 4) && t >= 1) k_lstm_fwd<2><<<T2V_NWG, 1024, 0, stream>>>(ad);
        }
        if (t < T_out && (mask & 2)) {
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
            f.prof = g_t2v_prof;
            f.ex = qp_tail;
            f.sync = (unsigned*)(qp_tail + 32768);
            f.epoch = t + 1;
            t2v_launch_attn_fwd(f, B, T_in, stream);
        }
    }
    if (overlap) {
        (void)hipEventRecord(g_side.join, sb);
        (void)hipStreamWaitEvent(stream, g_side.join, 0);
    }
    return t2v_check_launch();
}

extern "C" int t2v_decoder_train_fwd(const t2v_dec_weights* w, const t2v_dec_train_bufs* s,
                                     int B, int T_in, int T_out, float p_att, float p_dec,
                                     uint64_t seed, void* stream_) {
    return launch_train_fwd(w, s, B, T_in, T_out, p_att, p_dec, seed, stream_, t2v_overlap_enabled() ? (2 | 16) : 3);
}

extern "C" int t2v_decoder_replay_fwd_kernels(const t2v_dec_weights* w, const t2v_dec_train_bufs* s,
                                              int B, int T_in, int T_out, float p_att, float p_dec,
                                              uint64_t seed, int kernel_mask, void* stream_) {
    return launch_train_fwd(w, s, B, T_in, T_out, p_att, p_dec, seed, stream_, kernel_mask & 15);
}

void t2v_launch_lstm_fwd(int mode, const LstmFwdArgs& a, hipStream_t stream) {
    if (mode == 3) k_lstm_fwd<3><<<T2V_NWG, 1024, 0, stream>>>(a);
    else if (mode == 1) k_lstm_one256<true><<<T2V_NWG, 256, 0, stream>>>(a);      // decode: attention_rnn + prenet columns
    else if (mode == 2) k_lstm_one256<false><<<T2V_NWG, 256, 0, stream>>>(a);     // decode: decoder_rnn
    else k_lstm_fwd<0><<<T2V_NWG, 1024, 0, stream>>>(a);
}
