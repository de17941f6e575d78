// Teacher-forced decoder recurrence, forward, for hparams.bf16_run (BASELINE configs[4]: B = 16 per GPU) as ONE persistent
// launch (reference Decoder.forward's time loop model.py:415-421 -> Decoder.decode 346-389 -> Attention.forward 67-88;
// replaces fp16_optimizer.py's fp16 model copy on this path: bf16 weight operands, fp32 accumulation / cell state /
// saved activations, exactly what the launch-per-step pair k_lstm_fwd256<true> + k_attn_fwd computes).
//
// Round 5.  decoder_train_persist.hip keeps the fp32 weights of 4-5 hidden units per workgroup in 160 VGPRs per thread and
// multiplies them with packed VALU FMAs: that is what limits it to B <= 6.  Here the weights are rounded to bf16 (the same
// RNE rounding as t2v_pack_lstm_weights_bf16), a workgroup owns EIGHT hidden units of both cells (two 16-row MFMA tiles per
// cell: 4 units x 4 gates, unit-major), and the batch is the N dimension of v_mfma_f32_16x16x32_bf16: B <= 16 is one tile.
//
// Roles (one 512-thread workgroup per CU, all co-resident):
//   T : workgroups [0, 8B)      — attention slice (item b = wg / 8, s = wg % 8): the role of decoder_train_persist.hip,
//                                 unchanged arithmetic (fp32: bf16_run keeps the attention in fp32)
//   L : workgroups [128, 256)   — LSTM rows of both cells: workgroup j owns units [8j, 8j + 8).  K is split over the 8 waves:
//                                 wave w multiplies h_att k-blocks [4w, 4w+4), ctx k-blocks [32+2w, 32+2w+2) and h_dec
//                                 k-blocks [48+4w, 48+4w+4) (32 columns each) of both cells — 48 + 80 weight registers.
//   (workgroups [8B, 128) leave at once when B < 16)
//
// The state exchange IS the MFMA operand.  Row r of GH = [h_att(r-1) | ctx(r-1) | h_dec(r-2)] rounded to bf16, laid out
// [k / 8][item 0..15][8 consecutive k] — 16 bytes per (k-group, item) — so the B operand of k-block kb for lane l (item
// l & 15, k-group l >> 4) is the 16 bytes at kb * 1024 + 16 l: a wave polls ITS OWN k-blocks with fully coalesced 1-KB loads
// straight into the registers the MFMA reads.  No LDS staging of the state, no all-threads gather, no barrier between
// "arrived" and "multiplied"; what a workgroup pulls per step is 80 KB (B = 16) where the fp32 kernel pulls 61 KB at B = 6.
// Sentinel protocol as in decoder_train_persist.hip: rows are pre-filled with 0xFFFFFFFF (a pair of bf16 NaNs that rounding
// a finite fp32 never produces), producers store write-through (sc1), consumers poll the payload with sc1 loads.
// The attention slices read h_att(t) in fp32 from HX (T rows x 16 items x 1024) — the attention stays fp32 like the
// launch-per-step path — and publish their 64 context columns as eight 16-byte bf16 chunks.
//
// Per step the chain is  A-finish(ctx(t-1): 4 MFMAs) -> cross-wave reduce -> cell -> h_att hop -> attention(t) -> ctx hop;
// decoder_rnn(t-1) finishes behind the same barrier, everything else (h_att / h_dec parts of both products) runs in the shadow
// of attention(t).  Writes the same arena as the launch-per-step loop (XS, CA, CD, GA, GD, AL, ACUM, S), same counter-based
// dropout masks: either backward runs on it.
//
// Measured on the way (same-box A/B through a run-time switch, B = 16, T_in = 84, T = 400; none kept): TWO polls in flight on
// the ctx / h_att hand-offs 9.63 -> 10.2 / 9.9 us per step (the polling traffic is what stretches a look's round trip to ~1 us);
// re-polling only the k-blocks that are still missing 9.63 -> 9.65; partial energies as 16-byte stores + 16-byte polls through
// an LDS transpose 9.51 -> 9.63 (the extra barrier costs more than the narrower stores); a gentler nap rule: no change.
#include "t2v_common.h"
#include "t2v_kernels.h"

#define P16_THREADS 512
#define P16_MAXB 16
#define P16_MAXT 224                 // LDS-resident W_q / processed-memory slices up to here
#define P16_MAXT_LONG 560            // register-resident ones beyond (k_dec_train_persist16<true>; decoder_train_persist.hip has the arithmetic)
#define P16_NTI_LONG 5               // 16-position tiles per wave of the long form
#define P16_NL 128
#define P16_L0 (T2V_NWG - P16_NL)
#define P16_SPIN 1500000u
#define P16_SENT 0xFFFFFFFFu
#define P16_GROW (T2V_XW / 8 * 16 * 16)        // bytes per GH row: 320 k-groups x 16 items x 16 B = 81 920
#define P16_HROW (16 * T2V_H * 4)              // bytes per HX row: 16 items x 1024 fp32 = 65 536

struct P16Args {
    const float* w_ih_att; const float* w_hh_att; const float* w_ih_dec; const float* w_hh_dec;
    const float* bias_dec; const float* wq; const float* wcomb; const float* v;
    const float* gpre; const float* memory; const float* pm; const int32_t* lengths;
    float* XS; float* CA; float* CD; float* GA; float* GD; float* AL; float* ACUM; float* S;
    void* GH;                 // (T+2) rows x 81 920 B, sentinel-filled: bf16 state rows in MFMA-operand order
    float* HX;                // T rows x 16 x 1024 fp32: h_att(t) for the attention slices
    float* EX;                // T x B x 8 x Tcap partial energies
    unsigned* err;
    int B, T_in, T_out;
    float p_att, p_dec;
    uint64_t seed;
    const t2v_step_params* step;
    unsigned long long* prof;
};
#define P16_STAMP(COND, I) do { if (a.prof && (COND) && (threadIdx.x & 63) == 0) a.prof[(I)] = __builtin_readcyclecounter(); } while (0)
#define P16_WALL(COND, I) do { if (a.prof && (COND) && (threadIdx.x & 63) == 0) a.prof[(I)] = wall_clock64(); } while (0)
// per-workgroup time line of ONE step (t = T/2) on the chip-wide 100 MHz counter: prof[64 + workgroup * 8 + slot]
// (tools/dbg/persist16_prof.py passes a buffer of 64 + 256 * 8 words)
#define P16_RT(SLOT) do { if (a.prof && t == a.T_out / 2 && threadIdx.x == 0) a.prof[64 + blockIdx.x * 8 + (SLOT)] = __builtin_amdgcn_s_memrealtime(); } while (0)

typedef unsigned p16_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned p16_u32x2 __attribute__((ext_vector_type(2)));
#define P16_SC1 16
__device__ __forceinline__ __amdgpu_buffer_rsrc_t p16_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ p16_u32x4 p16_ld16(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, P16_SC1);
}
__device__ __forceinline__ unsigned p16_ld4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, P16_SC1);
}
__device__ __forceinline__ void p16_st16(__amdgpu_buffer_rsrc_t r, unsigned off, p16_u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)off, 0, P16_SC1);
}
__device__ __forceinline__ void p16_st8(__amdgpu_buffer_rsrc_t r, unsigned off, p16_u32x2 v) {
    __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)off, 0, P16_SC1);
}
__device__ __forceinline__ void p16_st4(__amdgpu_buffer_rsrc_t r, unsigned off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)off, 0, P16_SC1);
}
__device__ __forceinline__ bool p16_ok4(p16_u32x4 v) {
    return v[0] != P16_SENT && v[1] != P16_SENT && v[2] != P16_SENT && v[3] != P16_SENT;
}
__device__ __forceinline__ f32x4 p16_mfma(p16_u32x4 w, p16_u32x4 x, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(t2v_bf16x8, w), __builtin_bit_cast(t2v_bf16x8, x), c, 0, 0, 0);
}

// Poll N consecutive k-blocks of one GH row straight into MFMA B operands.  off = row + kb0 * 1024 + 16 * lane.  A lane
// whose item does not exist (live == false) never waits and reads zeros.  Wave-uniform loop; returns the failed rounds.
template <int N>
__device__ __forceinline__ int p16_poll(p16_u32x4 (&x)[N], __amdgpu_buffer_rsrc_t rG, unsigned off, bool live, int nap,
                                        unsigned* err, int* flag) {
    for (int i = 0; i < nap; i += 8) __builtin_amdgcn_s_sleep(8);
    int rounds = 0;
    for (;;) {
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] = p16_ld16(rG, off + 1024u * (unsigned)i);
        bool ok = true;
#pragma unroll
        for (int i = 0; i < N; ++i) ok = ok && p16_ok4(x[i]);
        if (__all(ok || !live)) break;
        __builtin_amdgcn_s_sleep(2);
        if (++rounds > (int)(P16_SPIN / 4) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = 0;
            break;
        }
    }
    if (!live) {
#pragma unroll
        for (int i = 0; i < N; ++i) x[i] = p16_u32x4{0u, 0u, 0u, 0u};
    }
    return rounds;
}

// bf16x8 A operand of one (tile, k-block): 8 consecutive columns of one gate row, read from the nn.LSTMCell tensors
__device__ __forceinline__ p16_u32x4 p16_wload(const float* p) {
    const float4 lo = *(const float4*)p, hi = *(const float4*)(p + 4);
    const uint4 u = t2v_pack_bf16x8(lo, hi);
    return p16_u32x4{u.x, u.y, u.z, u.w};
}

template <bool LONG>          // the attention role for 224 < T_in <= 560, as in k_dec_train_persist<.., true>
__global__ __launch_bounds__(P16_THREADS) void k_dec_train_persist16(P16Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const uint64_t seed = t2v_step_seed(a.seed, a.step);
    const int wg = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int B = a.B, Tp = a.T_in, T = a.T_out;
    const int NT = 8 * B;
    const __amdgpu_buffer_rsrc_t rG = p16_rsrc(a.GH), rH = p16_rsrc(a.HX), rE = p16_rsrc(a.EX);
    const int Tcap = (Tp + 15) & ~15;

#ifndef P16_ONLY_T     // (per-role register reports: tools/dbg/role_regs.sh builds the kernel with one role compiled out)
    if (wg >= P16_L0) {
        // =========================================================================== L role: 8 units of both cells
        f32x4* red = (f32x4*)lds;                            // [parity 2][cell 2][wave 8][tile 2][lane 64]
        float* hs = lds + 2 * 2 * 8 * 2 * 64 * 4;            // [cell wave 4][item 16][4 units]
        int* flag = (int*)(hs + 4 * 64);
        const int j = wg - P16_L0, u0 = 8 * j;
        const int n = lane & 15, g = lane >> 4;              // MFMA: item column / k-group (operands), unit of the tile (results)
        const bool live = n < B;
        // ---- weights: tile m, A row (lane & 15) = 4 * unit + gate -> gate row gate * 1024 + u0 + 4 m + unit; k-group g
        p16_u32x4 wa[2][6], wd[2][10];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const size_t row = (size_t)(n & 3) * T2V_H + u0 + 4 * m + (n >> 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = 32 * (4 * wave + i) + 8 * g;                       // h_att columns
                wa[m][i] = p16_wload(a.w_hh_att + row * T2V_H + k);
                wd[m][i] = p16_wload(a.w_ih_dec + row * T2V_KATT + k);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int k = 32 * (2 * wave + i) + 8 * g;                       // context columns
                wa[m][4 + i] = p16_wload(a.w_ih_att + row * (T2V_PRE + T2V_E) + T2V_PRE + k);
                wd[m][4 + i] = p16_wload(a.w_ih_dec + row * T2V_KATT + T2V_H + k);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = 32 * (4 * wave + i) + 8 * g;                       // h_dec columns
                wd[m][6 + i] = p16_wload(a.w_hh_dec + row * T2V_H + k);
            }
        }
        if (tid == 0) flag[0] = 1;
        // cell waves: 0, 1 = attention_rnn tiles 0, 1; 2, 3 = decoder_rnn tiles 0, 1.  Lane (item n, unit g of the tile)
        const bool is_cell = wave < 4;
        const int cm = wave & 1, ccell = wave >> 1;          // tile, cell (0 att, 1 dec)
        const int U = u0 + 4 * cm + g;
        const bool cell_on = is_cell && live;
        float cst = 0.f;                                     // this lane's cell state (pre-dropout), whole pass
        float bias[4] = {0.f, 0.f, 0.f, 0.f};
        if (cell_on && ccell == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bias[r] = a.bias_dec[r * T2V_H + U];
        }
        if (is_cell) hs[wave * 64 + lane] = 0.f;
        __syncthreads();
        // byte offsets of this wave's k-blocks inside a GH row
        const unsigned off_h = (unsigned)(4 * wave) * 1024u + 16u * (unsigned)lane;
        const unsigned off_c = (unsigned)(32 + 2 * wave) * 1024u + 16u * (unsigned)lane;
        const unsigned off_d = (unsigned)(48 + 4 * wave) * 1024u + 16u * (unsigned)lane;
        p16_u32x4 xc[2] = {p16_u32x4{0u, 0u, 0u, 0u}, p16_u32x4{0u, 0u, 0u, 0u}};     // ctx(-1) = 0
        f32x4 pA[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};         // h_att part of attention_rnn(t)
        f32x4 pD[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};         // h_att + h_dec parts of decoder_rnn(t-1)
        int nap_h = 0, nap_c = 0;
        // state-dropout factors of the NEXT cell evaluation of this lane (counter-based: functions of (seed, t, unit, item) alone),
        // drawn in the shadow of attention(t) instead of on the chain: two 64-bit hash rounds per cell and step
        float f_c = 1.0f, f_h = 1.0f;
        if (cell_on && ccell == 0) f_h = t2v_drop_scale(seed, T2V_RNG_ATT_H, 0, (uint32_t)n * T2V_H + U, a.p_att);

        for (int t = 0; t <= T; ++t) {
            const bool do_att = t < T, do_dec = t >= 1;
            const unsigned grow = (unsigned)(t + 1) * (unsigned)P16_GROW;          // row t+1 = [h_att(t) | ctx(t) | h_dec(t-1)]
            f32x4* redp = red + (size_t)(t & 1) * (2 * 8 * 2 * 64);
            P16_STAMP(wg == P16_L0 && wave == 0 && t == T / 2, 0);
            P16_RT(0);
            // Prenet term of this step for the attention_rnn cell lanes (issued before the products: latency hidden)
            float gp[4] = {0.f, 0.f, 0.f, 0.f};
            if (cell_on && ccell == 0 && do_att) {
#pragma unroll
                for (int r = 0; r < 4; ++r) gp[r] = a.gpre[((size_t)t * B + n) * T2V_G + r * T2V_H + U];
            }
            // ---- both cells finish with ctx(t-1): 4 + 4 MFMAs, partial tiles to LDS
            if (do_att) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    f32x4 acc = p16_mfma(wa[m][4], xc[0], pA[m]);
                    acc = p16_mfma(wa[m][5], xc[1], acc);
                    redp[((0 * 8 + wave) * 2 + m) * 64 + lane] = acc;
                }
            }
            if (do_dec) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    f32x4 acc = p16_mfma(wd[m][4], xc[0], pD[m]);
                    acc = p16_mfma(wd[m][5], xc[1], acc);
                    redp[((1 * 8 + wave) * 2 + m) * 64 + lane] = acc;
                }
            }
            P16_STAMP(wg == P16_L0 && wave == 0 && t == T / 2, 1);
            __syncthreads();
            if (flag[0] != 1) return;
            P16_STAMP(wg == P16_L0 && wave == 0 && t == T / 2, 2);
            P16_RT(1);
            // ---- cell update + publish (4 waves, one tile of one cell each)
            if (is_cell && (ccell == 0 ? do_att : do_dec)) {
                const f32x4* rp = redp + ((ccell * 8) * 2 + cm) * 64 + lane;       // wave stride: 2 * 64
                const f32x4 s4 = ((rp[0] + rp[128]) + (rp[256] + rp[384])) + ((rp[512] + rp[640]) + (rp[768] + rp[896]));
                P16_STAMP(wg == P16_L0 && wave == 0 && t == T / 2 && s4[0] != 123.f, 13);
                const int tt = ccell == 0 ? t : t - 1;
                float hd = 0.f, c = 0.f, gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f;
                if (cell_on) {
                    gi = sigmoidf_(s4[0] + (ccell == 0 ? gp[0] : bias[0]));
                    gf = sigmoidf_(s4[1] + (ccell == 0 ? gp[1] : bias[1]));
                    gg = tanhf_(s4[2] + (ccell == 0 ? gp[2] : bias[2]));
                    go = sigmoidf_(s4[3] + (ccell == 0 ? gp[3] : bias[3]));
                    c = gf * (cst * f_c) + gi * gg;
                    cst = c;
                    hd = go * tanhf_(c) * f_h;
                }
                P16_STAMP(wg == P16_L0 && wave == 0 && t == T / 2 && hd != 123.f, 14);
                // publish FIRST (the write-through store is what attention(t) waits for): lane b < B of this wave sends item b's
                // 4 units of the tile — 8 bytes of bf16 into the state row (+ 16 bytes of fp32 into HX for the attention slices);
                // the LDS round trip stays inside the wave (in-order), no barrier
                hs[wave * 64 + 4 * n + g] = hd;
                const float4 hv4 = *(const float4*)(hs + wave * 64 + 4 * (lane & 15));
                if (lane < B && (ccell == 0 || t < T)) {
                    const p16_u32x2 pk = {t2v_pack_bf16x2(hv4.x, hv4.y), t2v_pack_bf16x2(hv4.z, hv4.w)};
                    const unsigned kg = (unsigned)((ccell == 0 ? 0 : T2V_KATT) / 8 + j);
                    p16_st8(rG, grow + (kg * 16u + (unsigned)lane) * 16u + 8u * (unsigned)cm, pk);
                    if (ccell == 0)
                        p16_st16(rH, (unsigned)t * (unsigned)P16_HROW + (unsigned)(lane * T2V_H + u0 + 4 * cm) * 4u,
                                 p16_u32x4{__float_as_uint(hv4.x), __float_as_uint(hv4.y), __float_as_uint(hv4.z), __float_as_uint(hv4.w)});
                }
                P16_STAMP(wg == P16_L0 && wave == 0 && t == T / 2, 15);
                if (cell_on) {          // the saved activations follow (plain stores)
                    if (ccell == 0) a.CA[((size_t)(t + 1) * B + n) * T2V_H + U] = c;
                    else a.CD[((size_t)t * B + n) * T2V_H + U] = c;
                    float* gsv = ccell == 0 ? a.GA : a.GD;
                    if (gsv) {
                        float* gs = gsv + ((size_t)tt * B + n) * T2V_G + U;
                        gs[0] = gi; gs[T2V_H] = gf; gs[2 * T2V_H] = gg; gs[3 * T2V_H] = go;
                    }
                    a.XS[((size_t)(t + 1) * B + n) * T2V_XW + (ccell == 0 ? 0 : T2V_KATT) + U] = hd;
                }
            }
            if (t == T) break;
            P16_STAMP(wg == P16_L0 && wave == 0 && t == T / 2, 3);
            P16_RT(2);
            if (cell_on) {          // dropout factors of this lane's next cell evaluation: step tt' = t + 1 (attention_rnn) / t (decoder_rnn)
                const int ntt = ccell == 0 ? t + 1 : t;
                const uint32_t idx = (uint32_t)n * T2V_H + U;
                const float p = ccell == 0 ? a.p_att : a.p_dec;
                f_c = ntt > 0 ? t2v_drop_scale(seed, ccell == 0 ? T2V_RNG_ATT_C : T2V_RNG_DEC_C, ntt - 1, idx, p) : 1.0f;
                f_h = t2v_drop_scale(seed, ccell == 0 ? T2V_RNG_ATT_H : T2V_RNG_DEC_H, ntt, idx, p);
            }
            // ---- row t+1 in the shadow of attention(t): h_att(t) -> its share of attention_rnn(t+1) and decoder_rnn(t);
            // h_dec(t-1) -> decoder_rnn(t); ctx(t) last (the chain)
            {
                p16_u32x4 xh[4];
                const int rounds = p16_poll<4>(xh, rG, grow + off_h, live, nap_h, a.err, flag);
                nap_h = t2v_adapt_nap(nap_h, rounds);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accD = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        accA = p16_mfma(wa[m][i], xh[i], accA);
                        accD = p16_mfma(wd[m][i], xh[i], accD);
                    }
                    pA[m] = accA;
                    pD[m] = accD;
                }
            }
            P16_STAMP(wg == P16_L0 && wave == 0 && t == T / 2, 4);
            P16_RT(3);
            if (t >= 1) {
                p16_u32x4 xd[4];
                p16_poll<4>(xd, rG, grow + off_d, live, 0, a.err, flag);
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int i = 0; i < 4; ++i) pD[m] = p16_mfma(wd[m][6 + i], xd[i], pD[m]);
            }
            P16_STAMP(wg == P16_L0 && wave == 0 && t == T / 2, 5);
            P16_RT(4);
            {
                const int rounds = p16_poll<2>(xc, rG, grow + off_c, live, nap_c, a.err, flag);
                nap_c = t2v_adapt_nap(nap_c, rounds);
                if (a.prof && t == T / 2 && tid == 0) { a.prof[64 + wg * 8 + 6] = (unsigned long long)rounds; a.prof[64 + wg * 8 + 7] = (unsigned long long)nap_c; }
            }
            P16_RT(5);
            P16_WALL(wg == P16_L0 && wave == 0 && t == T / 2, 23);
        }
        return;
    }
#endif
#ifdef P16_ONLY_L
    return;
#else
    if (wg >= NT) return;

    // =============================================================================== T role: attention slice (b, s)
    const int ab = wg >> 3, as = wg & 7;
    constexpr int NTI = LONG ? P16_NTI_LONG : 2;         // position tiles per wave: tile jt = wave + 8 i
    constexpr int NPP = LONG ? 2 : 1;                    // positions per thread in the softmax: tid + 512 u
    const int TW = Tcap + 32;
    float* wq_s = lds;                                   // [16][1028]   (LONG: in registers)
    float* mem_s = wq_s + (LONG ? 0 : 16 * 1028);        // [Tcap][64]
    float* pm_s = mem_s + Tcap * 64;                     // [Tcap][16]   (LONG: in registers)
    float* win = pm_s + (LONG ? 0 : Tcap * 16);          // [2][TW]: alignment window, index x <-> position x - 15
    float* eall = win + 2 * TW;                          // [Tcap] (LONG: + T2V_CTX_PAD, zero from Tp on: t2v_ctx_partial)
    float* hx = eall + Tcap + (LONG ? T2V_CTX_PAD : 0);  // [1024] h_att(t) of this item
    float* qv = hx + T2V_H;                              // [16]
    float* qred = qv + 16;                               // [32][16]
    float* cred = qred + 32 * 16;                        // [8][64]
    float* rsm = cred + 8 * 64;                          // [32]
    float* rss = rsm + 32;                               // [32]
    float* cfin = rss + 32;                              // [64] finished context columns (wave 0)
    int* flag = (int*)(cfin + 64);
    const int g = lane >> 4, c16 = lane & 15;
    float4 wqr[LONG ? 8 : 1], pmr[LONG ? NTI : 1];
    if constexpr (LONG) {
        const float* wrow = a.wq + (size_t)(16 * as + (tid >> 5)) * 1024 + 4 * (tid & 31);
#pragma unroll
        for (int i = 0; i < 8; ++i) wqr[i] = *(const float4*)(wrow + 128 * i);
#pragma unroll
        for (int i = 0; i < NTI; ++i) {
            const int jp = 16 * (wave + 8 * i) + c16;
            pmr[i] = *(const float4*)(a.pm + ((size_t)ab * Tp + min(jp, Tp - 1)) * T2V_A + 16 * as + 4 * g);
        }
    } else {
        for (int i = tid; i < 16 * 1024; i += P16_THREADS) wq_s[(i >> 10) * 1028 + (i & 1023)] = a.wq[(size_t)(16 * as) * 1024 + i];
    }
    for (int i = tid; i < Tp * 64; i += P16_THREADS) mem_s[i] = a.memory[((size_t)ab * Tp + (i >> 6)) * T2V_E + 64 * as + (i & 63)];
    if constexpr (!LONG)
        for (int i = tid; i < Tp * 16; i += P16_THREADS) pm_s[i] = a.pm[((size_t)ab * Tp + (i >> 4)) * T2V_A + 16 * as + (i & 15)];
    for (int i = tid; i < 2 * TW; i += P16_THREADS) win[i] = 0.f;
    if constexpr (LONG)
        for (int i = Tp + tid; i < Tcap + T2V_CTX_PAD; i += P16_THREADS) eall[i] = 0.f;
    if (tid == 0) flag[0] = 1;
    float areg[16];
    {
        const float4* wp = (const float4*)(a.wcomb + (16 * as + c16) * 64 + 16 * g);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 w4 = wp[u];
            areg[4 * u] = w4.x; areg[4 * u + 1] = w4.y; areg[4 * u + 2] = w4.z; areg[4 * u + 3] = w4.w;
        }
    }
    const float4 vr = *(const float4*)(a.v + 16 * as + 4 * g);
    const int len = a.lengths ? a.lengths[ab] : Tp;
    __syncthreads();
    int h_nap = 0;

    for (int t = 0; t < T; ++t) {
        const unsigned grow = (unsigned)(t + 1) * (unsigned)P16_GROW;
        P16_STAMP(wg == 0 && wave == 0 && t == T / 2, 8);
        P16_RT(0);
        // ---- location features of this step's tiles (fused filter, K = 64): they depend on alpha(t-1) only
        f32x4 lacc[NTI];
#pragma unroll
        for (int i = 0; i < NTI; ++i) {
            const int jt = wave + 8 * i;
            lacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (16 * jt < Tp) {
                float bop[16];
#pragma unroll
                for (int st = 0; st < 16; ++st) {
                    const int kk = 4 * st + g;
                    bop[st] = win[(kk >> 5) * TW + 16 * jt + c16 + (kk & 31)];
                }
                f32x4 l0 = {0.f, 0.f, 0.f, 0.f}, l1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < 16; st += 2) {
                    l0 = mfma16x4(areg[st], bop[st], l0);
                    l1 = mfma16x4(areg[st + 1], bop[st + 1], l1);
                }
                lacc[i] = l0 + l1;
            }
        }
        // ---- h_att(t) of this item: 4 KB of fp32 in HX, 16 bytes per thread of waves 0..3; nap, then poll the payload
        if (tid < 256) {
            const unsigned s0 = (unsigned)t * (unsigned)P16_HROW + (unsigned)(ab * T2V_H + 4 * tid) * 4u;
            p16_u32x4 v;
            for (int i = 0; i < h_nap; i += 8) __builtin_amdgcn_s_sleep(8);
            int rounds = 0;
            for (;;) {
                v = p16_ld16(rH, s0);
                if (__all(p16_ok4(v))) break;
                __builtin_amdgcn_s_sleep(1);
                if (++rounds > (int)(P16_SPIN / 4) || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    flag[0] = 0;
                    break;
                }
            }
            h_nap = t2v_adapt_nap(h_nap, rounds);
            P16_WALL(wg == 0 && wave == 0 && t == T / 2, 21);
            P16_RT(1);
            if (a.prof && t == T / 2 && tid == 0) { a.prof[64 + wg * 8 + 6] = (unsigned long long)rounds; a.prof[64 + wg * 8 + 7] = (unsigned long long)h_nap; }
            *(float4*)(hx + 4 * tid) = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
        __syncthreads();
        if (flag[0] != 1) return;
        P16_STAMP(wg == 0 && wave == 0 && t == T / 2, 9);
        // ---- query slice: thread = (dim d = tid >> 5, k part kq = tid & 31)
        {
            const int d = tid >> 5, kq = tid & 31;
            const float* wrow = wq_s + d * 1028 + 4 * kq;
            const float* hp = hx + 4 * kq;
            float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 w4 = LONG ? wqr[LONG ? i : 0] : *(const float4*)(wrow + 128 * i);
                const float4 h4 = *(const float4*)(hp + 128 * i);
                acc0 = fmaf(w4.x, h4.x, acc0); acc1 = fmaf(w4.y, h4.y, acc1);
                acc0 = fmaf(w4.z, h4.z, acc0); acc1 = fmaf(w4.w, h4.w, acc1);
            }
            float q = row16_sum(acc0 + acc1);
            q += __shfl_xor(q, 16, 64);
            if (kq == 0) qv[d] = q;
        }
        __syncthreads();
        const float4 q4 = *(const float4*)(qv + 4 * g);
        // ---- partial energies of this slice
        const unsigned exw = (unsigned)(((t * B + ab) * 8 + as) * Tcap) * 4u;
#pragma unroll
        for (int i = 0; i < NTI; ++i) {
            const int jt = wave + 8 * i;
            if (16 * jt < Tp) {
                const f32x4 acc = lacc[i];
                const int jp = 16 * jt + c16;
                const float4 pm4 = LONG ? pmr[LONG ? i : 0] : *(const float4*)(pm_s + min(jp, Tp - 1) * 16 + 4 * g);
                float4 sv;
                sv.x = tanhf_(q4.x + acc[0] + pm4.x); sv.y = tanhf_(q4.y + acc[1] + pm4.y);
                sv.z = tanhf_(q4.z + acc[2] + pm4.z); sv.w = tanhf_(q4.w + acc[3] + pm4.w);
                float esum = vr.x * sv.x + vr.y * sv.y + vr.z * sv.z + vr.w * sv.w;
                esum += __shfl_xor(esum, 16, 64);
                esum += __shfl_xor(esum, 32, 64);
                if (g == 0 && jp < Tp) p16_st4(rE, exw + 4u * (unsigned)jp, esum);
                if (a.S && jp < Tp) *(float4*)(a.S + (((size_t)t * B + ab) * Tp + jp) * T2V_A + 16 * as + 4 * g) = sv;
            }
        }
        P16_STAMP(wg == 0 && wave == 0 && t == T / 2, 10);
        P16_RT(2);
        // (two copies of the softmax: the one-position form is kept word for word so that the short kernels keep their instruction
        // stream — round 6 checked the ISA of <.., false> against the previous build, identical)
        if constexpr (!LONG) {
            // ---- the 8 partials of every position (fixed order), masked softmax
            float ev0 = -INFINITY;
            if (tid < Tp) {
                const unsigned e0 = (unsigned)((t * B + ab) * 8 * Tcap + tid) * 4u;
                unsigned p[8];
                unsigned spins = 0;
                for (;;) {
                    bool ok = true;
    #pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        p[i] = p16_ld4(rE, e0 + (unsigned)(i * Tcap) * 4u);
                        ok = ok && p[i] != P16_SENT;
                    }
                    if (ok) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > P16_SPIN || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        flag[0] = 0;
                        break;
                    }
                }
                const float ev = ((__uint_as_float(p[0]) + __uint_as_float(p[1])) + (__uint_as_float(p[2]) + __uint_as_float(p[3]))) +
                                 ((__uint_as_float(p[4]) + __uint_as_float(p[5])) + (__uint_as_float(p[6]) + __uint_as_float(p[7])));
                ev0 = tid < len ? ev : -INFINITY;
            }
            P16_RT(3);
            {
                float mloc = ev0;
                mloc = T2V_DPP_MAX(mloc, 0xB1); mloc = T2V_DPP_MAX(mloc, 0x4E);
                mloc = T2V_DPP_MAX(mloc, 0x141); mloc = T2V_DPP_MAX(mloc, 0x140);
                mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
                mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
                if (lane == 0) rsm[wave] = mloc;
            }
            __syncthreads();
            if (flag[0] != 1) return;
            float m;
            {
                const float4 a0 = *(const float4*)rsm, a1 = *(const float4*)(rsm + 4);
                m = fmaxf(fmaxf(fmaxf(a0.x, a0.y), fmaxf(a0.z, a0.w)), fmaxf(fmaxf(a1.x, a1.y), fmaxf(a1.z, a1.w)));
            }
            const float e0v = tid < Tp ? expf(ev0 - m) : 0.f;
            {
                float sloc = row16_sum(e0v);
                sloc += __shfl_xor(sloc, 16, 64);
                sloc += __shfl_xor(sloc, 32, 64);
                if (lane == 0) rss[wave] = sloc;
            }
            __syncthreads();
            float ssum;
            {
                const float4 a0 = *(const float4*)rss, a1 = *(const float4*)(rss + 4);
                ssum = ((a0.x + a0.y) + (a0.z + a0.w)) + ((a1.x + a1.y) + (a1.z + a1.w));
            }
            const float al = e0v * (1.0f / ssum);
            if (tid < Tp) {
                eall[tid] = al;
                win[15 + tid] = al;                                        // previous weights of the next step
                const float cum = win[TW + 15 + tid] + al;                 // cumulative weights
                win[TW + 15 + tid] = cum;
                if (as == 0) {
                    a.AL[((size_t)(t + 1) * B + ab) * Tp + tid] = al;
                    a.ACUM[((size_t)(t + 1) * B + ab) * Tp + tid] = cum;
                }
            }
        } else {
            // ---- the 8 partials of every position (fixed order), masked softmax; thread -> positions tid + 512 u
            float ev0[NPP];
    #pragma unroll
            for (int u = 0; u < NPP; ++u) ev0[u] = -INFINITY;
            if (tid < Tp) {
                const unsigned e0 = (unsigned)((t * B + ab) * 8 * Tcap + tid) * 4u;
                unsigned p[NPP][8];
                unsigned spins = 0;
                for (;;) {
                    bool ok = true;
    #pragma unroll
                    for (int u = 0; u < NPP; ++u) {
                        // (a second position past the end re-reads the first one's words: no branch around the loads)
                        const unsigned eu = e0 + ((u > 0 && tid + P16_THREADS * u < Tp) ? (unsigned)(P16_THREADS * u) * 4u : 0u);
    #pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            p[u][i] = p16_ld4(rE, eu + (unsigned)(i * Tcap) * 4u);
                            ok = ok && p[u][i] != P16_SENT;
                        }
                    }
                    if (ok) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > P16_SPIN || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        flag[0] = 0;
                        break;
                    }
                }
    #pragma unroll
                for (int u = 0; u < NPP; ++u) {
                    const float ev = ((__uint_as_float(p[u][0]) + __uint_as_float(p[u][1])) + (__uint_as_float(p[u][2]) + __uint_as_float(p[u][3]))) +
                                     ((__uint_as_float(p[u][4]) + __uint_as_float(p[u][5])) + (__uint_as_float(p[u][6]) + __uint_as_float(p[u][7])));
                    ev0[u] = tid + P16_THREADS * u < len ? ev : -INFINITY;
                }
            }
            P16_RT(3);
            {
                float mloc = ev0[0];
    #pragma unroll
                for (int u = 1; u < NPP; ++u) mloc = fmaxf(mloc, ev0[u]);
                mloc = T2V_DPP_MAX(mloc, 0xB1); mloc = T2V_DPP_MAX(mloc, 0x4E);
                mloc = T2V_DPP_MAX(mloc, 0x141); mloc = T2V_DPP_MAX(mloc, 0x140);
                mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
                mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
                if (lane == 0) rsm[wave] = mloc;
            }
            __syncthreads();
            if (flag[0] != 1) return;
            float m;
            {
                const float4 a0 = *(const float4*)rsm, a1 = *(const float4*)(rsm + 4);
                m = fmaxf(fmaxf(fmaxf(a0.x, a0.y), fmaxf(a0.z, a0.w)), fmaxf(fmaxf(a1.x, a1.y), fmaxf(a1.z, a1.w)));
            }
            float e0v[NPP];
    #pragma unroll
            for (int u = 0; u < NPP; ++u) e0v[u] = tid + P16_THREADS * u < Tp ? expf(ev0[u] - m) : 0.f;
            {
                float sloc = e0v[0];
    #pragma unroll
                for (int u = 1; u < NPP; ++u) sloc += e0v[u];
                sloc = row16_sum(sloc);
                sloc += __shfl_xor(sloc, 16, 64);
                sloc += __shfl_xor(sloc, 32, 64);
                if (lane == 0) rss[wave] = sloc;
            }
            __syncthreads();
            float ssum;
            {
                const float4 a0 = *(const float4*)rss, a1 = *(const float4*)(rss + 4);
                ssum = ((a0.x + a0.y) + (a0.z + a0.w)) + ((a1.x + a1.y) + (a1.z + a1.w));
            }
            const float rinv = 1.0f / ssum;
    #pragma unroll
            for (int u = 0; u < NPP; ++u) {
                const int pos = tid + P16_THREADS * u;
                const float al = e0v[u] * rinv;
                if (pos < Tp) {
                    eall[pos] = al;
                    win[15 + pos] = al;                                        // previous weights of the next step
                    const float cum = win[TW + 15 + pos] + al;                 // cumulative weights
                    win[TW + 15 + pos] = cum;
                    if (as == 0) {
                        a.AL[((size_t)(t + 1) * B + ab) * Tp + pos] = al;
                        a.ACUM[((size_t)(t + 1) * B + ab) * Tp + pos] = cum;
                    }
                }
            }
        }
        __syncthreads();
        P16_STAMP(wg == 0 && wave == 0 && t == T / 2, 11);
        P16_RT(4);
        // ---- context columns [64 as, 64 as + 64): thread = (column c = tid & 63, part = tid >> 6)
        {
            const int c = tid & 63, part = tid >> 6;
            if constexpr (LONG) {
                // (up to 70 positions per thread: eight per round, reads first — 8 000 -> 6 500 cycles at 555 symbols; at <= 224 symbols
                // the plain loop is as fast and the short kernels keep their instruction stream)
                cred[part * 64 + c] = t2v_ctx_partial<8>(eall, mem_s, part, c, Tp);
            } else {
                float acc = 0.f;
                for (int jj = part; jj < Tp; jj += 8) acc = fmaf(eall[jj], mem_s[jj * 64 + c], acc);
                cred[part * 64 + c] = acc;
            }
        }
        __syncthreads();
        if (tid < 64) {
            float acc = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += cred[u * 64 + tid];
            cfin[tid] = acc;
            // publish: lanes 0..7 send 8 columns each as one 16-byte bf16 chunk (k-group 128 + 8 as + lane, item ab); the
            // LDS round trip stays inside this wave
            if (tid < 8) {
                const float4 c0 = *(const float4*)(cfin + 8 * tid), c1 = *(const float4*)(cfin + 8 * tid + 4);
                const uint4 pk = t2v_pack_bf16x8(c0, c1);
                const unsigned kg = (unsigned)(T2V_H / 8 + 8 * as + tid);
                p16_st16(rG, grow + (kg * 16u + (unsigned)ab) * 16u, p16_u32x4{pk.x, pk.y, pk.z, pk.w});
            }
            a.XS[((size_t)(t + 1) * B + ab) * T2V_XW + T2V_H + 64 * as + tid] = acc;       // (after the publish)
        }
        P16_WALL(wg == 0 && wave == 0 && t == T / 2, 22);
        P16_STAMP(wg == 0 && wave == 0 && t == T / 2, 12);
        P16_RT(5);
    }
#endif
}

// sentinel fill of the exchange buffers (16 bytes per thread and iteration)
__global__ __launch_bounds__(256) void k_p16_fill(uint4* p, size_t n16) {
    const uint4 s = {P16_SENT, P16_SENT, P16_SENT, P16_SENT};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = s;
}

static size_t p16_lds_bytes(int T_in) {
    const size_t Tcap = (size_t)((T_in + 15) / 16) * 16;
    const size_t lrole = 2 * 2 * 8 * 2 * 64 * 4 + 4 * 64 + 4;
    const size_t resident = T_in > P16_MAXT ? Tcap * 64 : 16 * 1028 + Tcap * 64 + Tcap * 16;     // LONG: W_q / processed memory in registers
    const size_t trole = resident + 2 * (Tcap + 32) + Tcap + (T_in > P16_MAXT ? T2V_CTX_PAD : 0) + T2V_H + 16 + 32 * 16 + 8 * 64 + 64 + 64 + 4;
    return sizeof(float) * (lrole > trole ? lrole : trole);
}
#define P16_LDS_MAX (160 * 1024)
static size_t p16_gh_floats(int T_out) { return (size_t)(T_out + 2) * (P16_GROW / 4); }
static size_t p16_hx_floats(int T_out) { return (size_t)T_out * (P16_HROW / 4); }
static size_t p16_ex_floats(int B, int T_in, int T_out) { return (size_t)T_out * B * 8 * t2v_tcap(T_in); }

static const void* p16_kernel(int T_in) {
    return T_in > P16_MAXT ? (const void*)k_dec_train_persist16<true> : (const void*)k_dec_train_persist16<false>;
}
static int p16_device_ok(int T_in, size_t lds) {
    static int cus = -1;
    if (cus < 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
        cus = prop.multiProcessorCount;
    }
    if (cus < T2V_NWG) return 0;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(p16_kernel(P16_MAXT), hipFuncAttributeMaxDynamicSharedMemorySize, P16_LDS_MAX) != hipSuccess ||
            hipFuncSetAttribute(p16_kernel(P16_MAXT + 1), hipFuncAttributeMaxDynamicSharedMemorySize, P16_LDS_MAX) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
        attr_set = true;
    }
    int nblk = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, p16_kernel(T_in), P16_THREADS, lds) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return nblk >= 1;
}

extern "C" int t2v_decoder_train_persist16_supported(int B, int T_in) {
    if (!(B >= 1 && B <= P16_MAXB && T_in >= 1 && T_in <= P16_MAXT_LONG && p16_lds_bytes(T_in) <= P16_LDS_MAX)) return 0;
    return p16_device_ok(T_in, p16_lds_bytes(T_in));
}
extern "C" long t2v_decoder_train_persist16_scratch_floats(int B, int T_in, int T_out) {
    if (B < 1 || B > P16_MAXB || T_in < 1 || T_out < 1) return 0;
    return (long)(p16_gh_floats(T_out) + p16_hx_floats(T_out) + p16_ex_floats(B, T_in, T_out));
}

extern "C" int t2v_decoder_train_fwd_persistent16(const t2v_dec_train_persist_weights* w, const t2v_dec_train_bufs* s, float* scratch,
                                                  int B, int T_in, int T_out, float p_att, float p_dec, uint64_t seed, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!w || !s || !scratch || T_out < 1 || !t2v_decoder_train_persist16_supported(B, T_in)) return T2V_ERR_ARG;
    if (!w->w_ih_att || !w->w_hh_att || !w->w_ih_dec || !w->w_hh_dec || !w->bias_dec || !w->wq || !w->wcomb || !w->v || !s->gpre ||
        !s->memory || !s->pm || !s->XS || !s->CA || !s->CD || !s->QP || !s->AL || !s->ACUM)
        return T2V_ERR_ARG;
    if ((uintptr_t)scratch & 15) return T2V_ERR_ARG;
    if (p16_gh_floats(T_out) * 4 >= 0x7fffffffull || p16_hx_floats(T_out) * 4 >= 0x7fffffffull ||
        p16_ex_floats(B, T_in, T_out) * 4 >= 0x7fffffffull)
        return T2V_ERR_ARG;                                  // 31-bit buffer offsets
    // per-pass resets: the sync / error words, the zero initial states of the arena (as t2v_decoder_train_fwd)
    unsigned* sync = (unsigned*)(s->QP + t2v_qp_sync_off(B));
    T2VZeroRegions z;
    z.add(sync, 64 * sizeof(uint32_t));
    z.add(s->XS, sizeof(float) * 2 * B * T2V_XW);
    z.add(s->CA, sizeof(float) * B * T2V_H);
    z.add(s->CD, sizeof(float) * B * T2V_H);
    z.add(s->AL, sizeof(float) * B * T_in);
    z.add(s->ACUM, sizeof(float) * B * T_in);
    t2v_zero_regions(z, stream);
    const size_t nfl = p16_gh_floats(T_out) + p16_hx_floats(T_out) + p16_ex_floats(B, T_in, T_out);
    k_p16_fill<<<1024, 256, 0, stream>>>((uint4*)scratch, nfl / 4);
    P16Args a;
    a.w_ih_att = w->w_ih_att; a.w_hh_att = w->w_hh_att; a.w_ih_dec = w->w_ih_dec; a.w_hh_dec = w->w_hh_dec;
    a.bias_dec = w->bias_dec; a.wq = w->wq; a.wcomb = w->wcomb; a.v = w->v;
    a.gpre = s->gpre; a.memory = s->memory; a.pm = s->pm; a.lengths = s->lengths;
    a.XS = s->XS; a.CA = s->CA; a.CD = s->CD; a.GA = s->GA; a.GD = s->GD; a.AL = s->AL; a.ACUM = s->ACUM; a.S = s->S;
    a.GH = scratch;
    a.HX = scratch + p16_gh_floats(T_out);
    a.EX = a.HX + p16_hx_floats(T_out);
    a.err = sync + 31;
    a.B = B; a.T_in = T_in; a.T_out = T_out; a.p_att = p_att; a.p_dec = p_dec; a.seed = seed;
    a.step = t2v_step_for(stream);
    a.prof = g_t2v_prof;
    if (T_in > P16_MAXT) k_dec_train_persist16<true><<<T2V_NWG, P16_THREADS, p16_lds_bytes(T_in), stream>>>(a);
    else k_dec_train_persist16<false><<<T2V_NWG, P16_THREADS, p16_lds_bytes(T_in), stream>>>(a);
    return t2v_check_launch();
}
