// Decoder recurrence, forward: weight packing + the two per-step kernels.
//   k_lstm_fwd : attention_rnn(t) and decoder_rnn(t-1) gate GEMVs on MFMA (weights streamed in
//                fragment order), LSTM cell update, state dropout, partial query projection.
//   k_attn_fwd : location-sensitive attention of step t (query sum, location conv + dense,
//                tanh/v energies, masked softmax, context).
// Reference semantics: Decoder.decode model.py:346-389, Attention.forward model.py:67-88.
#include <stdlib.h>
#include <type_traits>
#include "t2v_common.h"
#include "t2v_kernels.h"

// ------------------------------------------------------------------------------------------
// Weight packing.  The logical (4096, K) gate-major matrices
//   Wcat_att = [weight_hh | weight_ih[:, 256:768] (| weight_ih[:, 0:256] for the decode loop)]    K = 1536 / 1792
//   Wcat_dec = [weight_ih (h_att | ctx) | weight_hh]                                              K = 2560
// are never materialised: the pack kernels read the nn.LSTMCell tensors through a three-segment column map
// (segment boundaries are multiples of 4, so every float4 of 4 consecutive k stays inside one segment).
struct WSrc {
    const float* p[3];
    int ld[3], off[3], end[3];     // segment i covers logical columns [end[i-1], end[i]) -> p[i][row*ld[i] + off[i] + (c - end[i-1])]
    __device__ __forceinline__ const float* at(int row, int c) const {
        if (c < end[0]) return p[0] + (size_t)row * ld[0] + off[0] + c;
        if (c < end[1]) return p[1] + (size_t)row * ld[1] + off[1] + (c - end[0]);
        return p[2] + (size_t)row * ld[2] + off[2] + (c - end[1]);
    }
};
// Forward tile w (16 rows = 4 units x 4 gates, unit-major):
//   P[w][kb][lane][i] = W[(lane&3)*H + 4w + ((lane&15)>>2)][16kb + 4(lane>>4) + i]
// so one float4 per lane feeds four v_mfma_f32_16x16x4_f32 A operands (k = 16kb+4g+i, i=0..3).
template <bool BF16>
__global__ void k_pack_fwd(WSrc W, int K, void* __restrict__ Pv) {
    const int nkb = BF16 ? K / 32 : K / 16;     // bf16 packs: 32-column blocks, 8 values (one uint4) per lane
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // element index
    const size_t total = (size_t)T2V_NWG * nkb * 64;
    if (idx >= total) return;
    const int lane = idx & 63;
    const int kb = (idx >> 6) % nkb;         // tile-major: a workgroup streams one contiguous
    const int w = (idx >> 6) / nkb;          // 96/112/160 KiB region (tile stride is NOT a power of
    const int arow = lane & 15, g = lane >> 4;   // two, so L2/HBM channels are evenly loaded)
    const int row = (arow & 3) * T2V_H + 4 * w + (arow >> 2);
    if (BF16) {
        ((uint4*)Pv)[idx] = t2v_pack_bf16x8(*(const float4*)W.at(row, 32 * kb + 8 * g), *(const float4*)W.at(row, 32 * kb + 8 * g + 4));
    } else {
        ((float4*)Pv)[idx] = *(const float4*)W.at(row, 16 * kb + 4 * g);
    }
}
// Backward (transposed) tile n-tile w' of W^T (N = K columns of W become rows), reduction dim
// = 4096 gate rows, tile-major like the forward pack (a workgroup's 256 KB are contiguous):
//   PB[w'][kb][lane][i] = W[16kb + 4(lane>>4) + i][16w' + (lane&15)]
template <bool BF16>
__global__ void k_pack_bwd(WSrc W, int ncols, void* __restrict__ Pv) {
    const int nkb = BF16 ? T2V_G / 32 : T2V_G / 16;   // 128 / 256
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)(ncols / 16) * nkb * 64;
    if (idx >= total) return;
    const int lane = idx & 63;
    const int kb = (idx >> 6) % nkb;
    const int wt = (idx >> 6) / nkb;
    const int n = 16 * wt + (lane & 15);
    if (BF16) {
        const int k0 = 32 * kb + 8 * (lane >> 4);
        ((uint4*)Pv)[idx] = t2v_pack_bf16x8(make_float4(*W.at(k0 + 0, n), *W.at(k0 + 1, n), *W.at(k0 + 2, n), *W.at(k0 + 3, n)),
                                            make_float4(*W.at(k0 + 4, n), *W.at(k0 + 5, n), *W.at(k0 + 6, n), *W.at(k0 + 7, n)));
    } else {
        const int k0 = 16 * kb + 4 * (lane >> 4);
        ((float4*)Pv)[idx] = make_float4(*W.at(k0 + 0, n), *W.at(k0 + 1, n), *W.at(k0 + 2, n), *W.at(k0 + 3, n));
    }
}

// ------------------------------------------------------------------------------------------
#define MFMA4(ACC, WV, XV)                  \
    ACC = mfma16x4((WV).x, (XV).x, ACC);    \
    ACC = mfma16x4((WV).y, (XV).y, ACC);    \
    ACC = mfma16x4((WV).z, (XV).z, ACC);    \
    ACC = mfma16x4((WV).w, (XV).w, ACC)
#define MFMAW(ACC, WV, XV) mfma_block(ACC, WV, XV)

// ------------------------------------------------------------------------------------------
template <bool BF16>
static int pack_lstm_weights_impl(const float* w_ih_att, const float* w_hh_att, const float* w_ih_dec,
                                  const float* w_hh_dec, int k_att, float* packF_att, float* packF_dec,
                                  float* packB_att, float* packB_dec, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!w_ih_att || !w_hh_att || !w_ih_dec || !w_hh_dec) return T2V_ERR_ARG;
    if (!packF_att != !packF_dec || (!packF_att && !packB_att && !packB_dec)) return T2V_ERR_ARG;
    if (k_att != T2V_KATT && k_att != T2V_KATT_INF) return T2V_ERR_DIMS;
    const int IH = T2V_PRE + T2V_E;       // 768 input columns of attention_rnn: [prenet | ctx]
    WSrc A, D;
    A.p[0] = w_hh_att; A.ld[0] = T2V_H; A.off[0] = 0;       A.end[0] = T2V_H;
    A.p[1] = w_ih_att; A.ld[1] = IH;    A.off[1] = T2V_PRE; A.end[1] = T2V_KATT;
    A.p[2] = w_ih_att; A.ld[2] = IH;    A.off[2] = 0;       A.end[2] = T2V_KATT_INF;
    D.p[0] = w_ih_dec; D.ld[0] = T2V_KATT; D.off[0] = 0; D.end[0] = T2V_KATT;
    D.p[1] = w_hh_dec; D.ld[1] = T2V_H;    D.off[1] = 0; D.end[1] = T2V_XW;
    D.p[2] = w_hh_dec; D.ld[2] = T2V_H;    D.off[2] = 0; D.end[2] = T2V_XW;
    if (packF_att) {       // NULL (both): the caller runs the forward pass on the persistent kernel, which reads the tensors themselves
        const size_t n = (size_t)T2V_NWG * (k_att / (BF16 ? 32 : 16)) * 64;
        k_pack_fwd<BF16><<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(A, k_att, packF_att);
    }
    if (packF_dec) {
        const size_t n = (size_t)T2V_NWG * (T2V_XW / (BF16 ? 32 : 16)) * 64;
        k_pack_fwd<BF16><<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(D, T2V_XW, packF_dec);
    }
    if (packB_att) {   // only the recurrent 1536 columns matter for the data gradient
        const size_t n = (size_t)(T2V_KATT / 16) * (BF16 ? 128 : 256) * 64;
        k_pack_bwd<BF16><<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(A, T2V_KATT, packB_att);
    }
    if (packB_dec) {
        const size_t n = (size_t)(T2V_XW / 16) * (BF16 ? 128 : 256) * 64;
        k_pack_bwd<BF16><<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(D, T2V_XW, packB_dec);
    }
    return t2v_check_launch();
}

extern "C" int t2v_pack_lstm_weights(const float* w_ih_att, const float* w_hh_att, const float* w_ih_dec,
                                     const float* w_hh_dec, int k_att, float* packF_att, float* packF_dec,
                                     float* packB_att, float* packB_dec, void* stream_) {
    return pack_lstm_weights_impl<false>(w_ih_att, w_hh_att, w_ih_dec, w_hh_dec, k_att, packF_att, packF_dec, packB_att, packB_dec, stream_);
}
extern "C" int t2v_pack_lstm_weights_bf16(const float* w_ih_att, const float* w_hh_att, const float* w_ih_dec,
                                          const float* w_hh_dec, int k_att, void* packF_att, void* packF_dec,
                                          void* packB_att, void* packB_dec, void* stream_) {
    return pack_lstm_weights_impl<true>(w_ih_att, w_hh_att, w_ih_dec, w_hh_dec, k_att, (float*)packF_att, (float*)packF_dec,
                                        (float*)packB_att, (float*)packB_dec, stream_);
}

// Training step: FOUR waves per workgroup (tools/ubench_gemv_tiles.hip: for this 67 MB
// stream about 64 KB in flight per CU is the sweet spot — 16 waves that each put 26 KB in flight queue too much).
// Workgroup w = gate-row tile w of BOTH cells; wave v walks attention_rnn k-blocks [24v, 24v+24) then decoder_rnn
// k-blocks [40v, 40v+40): 8 rounds of 8 (W, x) float4 pairs, two rounds in flight; odd steps walk the same
// sequence backwards (the tail of the previous launch's stream is requested first).  Epilogue as k_lstm_fwd.
template <bool FLIP, bool WBF>       // WBF: the packs hold bf16 (uint2 per lane and k-block), one 16x16x16 bf16 MFMA per block
__device__ __forceinline__ void lstm256_stream(const float4* pa4, const float4* pd4, const float* xrow, int wave,
                                               f32x4& accA, f32x4& accD) {
    typedef typename std::conditional<WBF, uint2, float4>::type wt_t;
    const wt_t* pa = (const wt_t*)pa4;
    const wt_t* pd = (const wt_t*)pd4;
    wt_t wv[2][8];
    float4 xv[2][8];
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
            if (RR < 3) { MFMAW(accA, wv[(R) & 1][i], xv[(R) & 1][i]); }                          \
            else { MFMAW(accD, wv[(R) & 1][i], xv[(R) & 1][i]); }                                 \
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

// fp32, state loads shared between the two cells.  attention_rnn and decoder_rnn multiply the SAME columns [h_att | ctx]
// (k-blocks 0..95 of both packs), so a wave that owns the same k-range of both tiles loads those state values once:
// wave v walks shared blocks [24v, 24v+24) (6 rounds of 4: 4 W_att + 4 W_dec + 4 x loads) and then decoder_rnn's h_dec
// blocks [96+16v, 96+16v+16) (2 rounds of 8 pairs): 64 weight + 40 state loads per wave instead of 64 + 64 — the kernel
// is bound by the vector-memory instructions it issues.  Two rounds in flight = 16 weight loads, as before.
template <bool FLIP>
__device__ __forceinline__ void lstm256_stream_shared(const float4* pa, const float4* pd, const float* xrow, int wave,
                                                      f32x4& accA, f32x4& accD) {
    float4 wv[2][8], xv[2][8];
#define LS_LOAD(R)                                                                                \
    {                                                                                             \
        constexpr int RR = FLIP ? 7 - (R) : (R);                                                  \
        if (RR < 6) {                                                                             \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                       \
                const int ii = FLIP ? 3 - i : i;                                                  \
                const int kb = 24 * wave + 4 * RR + ii;                                           \
                wv[(R) & 1][i] = pa[(size_t)kb * 64];                                             \
                wv[(R) & 1][4 + i] = pd[(size_t)kb * 64];                                         \
                xv[(R) & 1][i] = *(const float4*)(xrow + 16 * kb);                                \
            }                                                                                     \
        } else {                                                                                  \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                       \
                const int ii = FLIP ? 7 - i : i;                                                  \
                const int kb = 96 + 16 * wave + 8 * (RR - 6) + ii;                                \
                wv[(R) & 1][i] = pd[(size_t)kb * 64];                                             \
                xv[(R) & 1][i] = *(const float4*)(xrow + 16 * kb);                                \
            }                                                                                     \
        }                                                                                         \
    }
#define LS_MATH(R)                                                                                \
    {                                                                                             \
        constexpr int RR = FLIP ? 7 - (R) : (R);                                                  \
        if (RR < 6) {                                                                             \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                       \
                MFMA4(accA, wv[(R) & 1][i], xv[(R) & 1][i]);                                      \
                MFMA4(accD, wv[(R) & 1][4 + i], xv[(R) & 1][i]);                                  \
            }                                                                                     \
        } else {                                                                                  \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) { MFMA4(accD, wv[(R) & 1][i], xv[(R) & 1][i]); } \
        }                                                                                         \
    }
#define LS_STEP(R, NEXT)                        \
    NEXT                                        \
    __builtin_amdgcn_sched_barrier(0);          \
    LS_MATH(R)                                  \
    __builtin_amdgcn_sched_barrier(0);
    LS_LOAD(0)
    __builtin_amdgcn_sched_barrier(0);
    LS_STEP(0, LS_LOAD(1))
    LS_STEP(1, LS_LOAD(2))
    LS_STEP(2, LS_LOAD(3))
    LS_STEP(3, LS_LOAD(4))
    LS_STEP(4, LS_LOAD(5))
    LS_STEP(5, LS_LOAD(6))
    LS_STEP(6, LS_LOAD(7))
    LS_STEP(7, )
#undef LS_STEP
#undef LS_MATH
#undef LS_LOAD
}

// bf16 packs: 32-column blocks, one uint4 weight load and two float4 state loads per block, v_mfma_f32_16x16x32_bf16.
// Like lstm256_stream_shared a wave owns the same column range of both cells: shared blocks [12v, 12v+12) (3 rounds of
// 4: 4 W_att + 4 W_dec + 8 state loads), then decoder_rnn's h_dec blocks [48+8v, 48+8v+8) (2 rounds of 4): 32 weight +
// 40 state loads per wave.  Two rounds in flight.
template <bool FLIP>
__device__ __forceinline__ void lstm256_stream_bf(const uint4* pa, const uint4* pd, const float* xrow8, int wave, f32x4& accA, f32x4& accD) {
    uint4 wv[2][8];
    float4 xv[2][8];
#define LB_LOAD(R)                                                                                \
    {                                                                                             \
        constexpr int RR = FLIP ? 4 - (R) : (R);                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
            const int ii = FLIP ? 3 - i : i;                                                      \
            const int jb = RR < 3 ? 12 * wave + 4 * RR + ii : 48 + 8 * wave + 4 * (RR - 3) + ii;  \
            if (RR < 3) wv[(R) & 1][i] = pa[(size_t)jb * 64];                                     \
            wv[(R) & 1][4 + i] = pd[(size_t)jb * 64];                                             \
            xv[(R) & 1][2 * i] = *(const float4*)(xrow8 + 32 * jb);                               \
            xv[(R) & 1][2 * i + 1] = *(const float4*)(xrow8 + 32 * jb + 4);                       \
        }                                                                                         \
    }
#define LB_MATH(R)                                                                                \
    {                                                                                             \
        constexpr int RR = FLIP ? 4 - (R) : (R);                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
            if (RR < 3) accA = mfma16x32_bf16(wv[(R) & 1][i], xv[(R) & 1][2 * i], xv[(R) & 1][2 * i + 1], accA); \
            accD = mfma16x32_bf16(wv[(R) & 1][4 + i], xv[(R) & 1][2 * i], xv[(R) & 1][2 * i + 1], accD);         \
        }                                                                                         \
    }
#define LB_STEP(R, NEXT)                        \
    NEXT                                        \
    __builtin_amdgcn_sched_barrier(0);          \
    LB_MATH(R)                                  \
    __builtin_amdgcn_sched_barrier(0);
    LB_LOAD(0)
    __builtin_amdgcn_sched_barrier(0);
    LB_STEP(0, LB_LOAD(1))
    LB_STEP(1, LB_LOAD(2))
    LB_STEP(2, LB_LOAD(3))
    LB_STEP(3, LB_LOAD(4))
    LB_STEP(4, )
#undef LB_STEP
#undef LB_MATH
#undef LB_LOAD
}

template <bool WBF>
__global__ __launch_bounds__(256) void k_lstm_fwd256(LstmFwdArgs a) {
    const uint64_t seed = t2v_step_seed(a.seed, a.step);
    const int w = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = lane & 15, g = lane >> 4;
    const bool bvalid = b < a.B;
    __shared__ f32x4 red[2][4][64];
    __shared__ float hs[16][4];
    // element = one lane's share of one k-block: float4 (fp32 packs) or uint2 (bf16 packs); same element indexing
    const float4* pa = WBF ? (const float4*)((const uint2*)a.packA + ((size_t)w * (T2V_KATT / 16)) * 64 + lane)
                           : a.packA + ((size_t)w * (T2V_KATT / 16)) * 64 + lane;
    const float4* pd = WBF ? (const float4*)((const uint2*)a.packD + ((size_t)w * (T2V_XW / 16)) * 64 + lane)
                           : a.packD + ((size_t)w * (T2V_XW / 16)) * 64 + lane;
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
    // the state-dropout factors (two 64-bit counter hashes, ~300 cycles each) do not depend on the weight stream: evaluated here, in
    // front of it, not behind the gate sums at the end of the launch (round 6)
    float fh_drop = 1.0f;
    if (cell_on) {
        const int tt0 = which == 0 ? a.t : a.t - 1;
        const float p0 = which == 0 ? a.p_att : a.p_dec;
        const uint32_t idx0 = (uint32_t)b * T2V_H + U;
        if (tt0 > 0) cprev *= t2v_drop_scale(seed, which == 0 ? T2V_RNG_ATT_C : T2V_RNG_DEC_C, tt0 - 1, idx0, p0);
        fh_drop = t2v_drop_scale(seed, which == 0 ? T2V_RNG_ATT_H : T2V_RNG_DEC_H, tt0, idx0, p0);
    }
    float wqr[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.do_att) {
        const float* wq = a.wqT + (size_t)(4 * w) * T2V_A + (tid & (T2V_A - 1));
        wqr[0] = wq[0]; wqr[1] = wq[T2V_A]; wqr[2] = wq[2 * T2V_A]; wqr[3] = wq[3 * T2V_A];
    }
    f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accD = {0.f, 0.f, 0.f, 0.f};
    if constexpr (WBF) {
        const uint4* pa8 = (const uint4*)a.packA + ((size_t)w * (T2V_KATT / 32)) * 64 + lane;
        const uint4* pd8 = (const uint4*)a.packD + ((size_t)w * (T2V_XW / 32)) * 64 + lane;
        const float* xrow8 = a.xs_prev + (size_t)(bvalid ? b : 0) * T2V_XW + 8 * g;
        if (a.t & 1) lstm256_stream_bf<true>(pa8, pd8, xrow8, wave, accA, accD);
        else lstm256_stream_bf<false>(pa8, pd8, xrow8, wave, accA, accD);
    } else {
        if (a.shared_x) {
            if (a.t & 1) lstm256_stream_shared<true>(pa, pd, xrow, wave, accA, accD);
            else lstm256_stream_shared<false>(pa, pd, xrow, wave, accA, accD);
        } else {
            if (a.t & 1) lstm256_stream<true, false>(pa, pd, xrow, wave, accA, accD);
            else lstm256_stream<false, false>(pa, pd, xrow, wave, accA, accD);
        }
    }
    red[0][wave][lane] = accA;
    red[1][wave][lane] = accD;
    __syncthreads();
    if (cell_on) {
        const f32x4 s = (red[which][0][lane] + red[which][1][lane]) + (red[which][2][lane] + red[which][3][lane]);
        const float gi = sigmoidf_(s[0] + addv[0]), gf = sigmoidf_(s[1] + addv[1]);
        const float gg = tanhf_(s[2] + addv[2]), go = sigmoidf_(s[3] + addv[3]);
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
        const float hd = h * fh_drop;
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
    const uint64_t seed = a.seed;       // decode loop: eval mode, state dropout off
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
        if (tt > 0) cprev *= t2v_drop_scale(seed, ATT ? T2V_RNG_ATT_C : T2V_RNG_DEC_C, tt - 1, idx, p);
        const float c = gf * cprev + gi * gg;
        const float h = go * tanhf_(c);
        (ATT ? a.ca_cur : a.cd_cur)[(size_t)b * T2V_H + U] = c;
        const float hd = h * t2v_drop_scale(seed, ATT ? T2V_RNG_ATT_H : T2V_RNG_DEC_H, tt, idx, p);
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
                           int t, float p_att, float p_dec, uint64_t seed, const t2v_step_params* step) {
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
    a.ga_t = (s->GA && t < T_out) ? s->GA + (size_t)t * B * T2V_G : nullptr;       // GA / GD NULL: forward only
    a.gd_t = (s->GD && t >= 1) ? s->GD + (size_t)(t - 1) * B * T2V_G : nullptr;
    a.wqT = w->wqT;
    a.qp = s->QP;
    {
        static const int sx = getenv("T2V_NO_SHARED_X") ? 0 : 1;
        a.shared_x = sx;
    }
    a.B = B;
    a.t = t;
    a.do_att = t < T_out;
    a.do_dec = t >= 1;
    a.p_att = p_att;
    a.p_dec = p_dec;
    a.seed = seed;
    a.step = step;
}

// mask bits: 1 = k_lstm_fwd256 (both cells), 2 = k_attn_fwd
static int launch_train_fwd(const t2v_dec_weights* w, const t2v_dec_train_bufs* s, int B, int T_in, int T_out,
                            float p_att, float p_dec, uint64_t seed, void* stream_, int mask) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!w || !s || B < 1 || B > 16 || T_in < 1 || T_in > T2V_MAX_T_IN || T_out < 1) return T2V_ERR_ARG;
    if (!w->wcomb) return T2V_ERR_ARG;
    // scratch behind the query partials: sync / error words, then the energy-exchange granules (t2v_kernels.h);
    // tags and words are zeroed per pass (a memset node that replays first under graph capture)
    unsigned* sync = (unsigned*)(s->QP + t2v_qp_sync_off(B));
    t2v_u64* ex = (t2v_u64*)(s->QP + t2v_qp_ex_off(B));
    T2VZeroRegions z;
    z.add(sync, 64 * sizeof(uint32_t));
    z.add(ex, sizeof(t2v_u64) * (size_t)B * 8 * t2v_tcap(T_in));
    if (mask == 3) {    // the initial states of the pass (reference model.py:280-296 initialize_decoder_states): rows 0 of
                        // the state arenas; row 1 of XS holds h_dec_{-1} = 0 beside fields that step 0 overwrites
        z.add(s->XS, sizeof(float) * 2 * B * T2V_XW);
        z.add(s->CA, sizeof(float) * B * T2V_H);
        z.add(s->CD, sizeof(float) * B * T2V_H);
        z.add(s->AL, sizeof(float) * B * T_in);
        z.add(s->ACUM, sizeof(float) * B * T_in);
    }
    t2v_zero_regions(z, stream);
    const t2v_step_params* step = t2v_step_for(stream);
    for (int t = 0; t <= T_out; ++t) {
        if (mask & 1) {
            LstmFwdArgs a;
            fill_lstm_args(a, w, s, B, T_out, t, p_att, p_dec, seed, step);
            if (w->packs_bf16) k_lstm_fwd256<true><<<T2V_NWG, 256, 0, stream>>>(a);
            else k_lstm_fwd256<false><<<T2V_NWG, 256, 0, stream>>>(a);
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
            f.wcomb = w->wcomb;
            f.v = w->v;
            f.xs_next = s->XS + (size_t)(t + 1) * B * T2V_XW;
            f.s_save = s->S ? s->S + (size_t)t * B * T_in * T2V_A : nullptr;
            f.T_in = T_in;
            f.prof = g_t2v_prof;
            f.ex = ex;
            f.err = sync + 31;
            f.epoch = (unsigned)t + 1u;
            t2v_launch_attn_fwd(f, B, T_in, stream);
        }
    }
    return t2v_check_launch();
}

extern "C" int t2v_decoder_train_fwd(const t2v_dec_weights* w, const t2v_dec_train_bufs* s,
                                     int B, int T_in, int T_out, float p_att, float p_dec,
                                     uint64_t seed, void* stream_) {
    return launch_train_fwd(w, s, B, T_in, T_out, p_att, p_dec, seed, stream_, 3);
}

extern "C" int t2v_decoder_replay_fwd_kernels(const t2v_dec_weights* w, const t2v_dec_train_bufs* s,
                                              int B, int T_in, int T_out, float p_att, float p_dec,
                                              uint64_t seed, int kernel_mask, void* stream_) {
    return launch_train_fwd(w, s, B, T_in, T_out, p_att, p_dec, seed, stream_, kernel_mask & 3);
}

extern "C" long t2v_decoder_qp_floats(int B, int T_in) {
    return (B < 1 || T_in < 1) ? 0 : (long)t2v_qp_floats(B, T_in);
}

void t2v_launch_lstm_fwd(int mode, const LstmFwdArgs& a, hipStream_t stream) {
    if (mode == 1) k_lstm_one256<true><<<T2V_NWG, 256, 0, stream>>>(a);      // decode: attention_rnn + prenet columns
    else k_lstm_one256<false><<<T2V_NWG, 256, 0, stream>>>(a);               // decode: decoder_rnn
}
