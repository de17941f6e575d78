// Hand-written BPTT of the teacher-forced decoder loop for hparams.bf16_run (BASELINE configs[4]: B = 16 per GPU) as ONE
// persistent launch (reference: autograd over Decoder.decode, model.py:346-389 / train.py:225; replaces the fp16 path of
// fp16_optimizer.py:51-382 on this loop).  Round 5.  Same arithmetic as the launch-per-step pair k_lstm_bwd256<true> +
// k_attn_cell_bwd: transposed LSTM weights and gate gradients rounded to bf16 (RNE) in front of every product, fp32
// accumulation, fp32 cell / attention backward, fp32 saved gradients (DGA, DGD, DCTX, dpre in S, dq partials).
//
// decoder_train_bwd_persist.hip (fp32, B <= 6) lets a workgroup own output COLUMNS of Wcat^T with all 4096 gate rows in
// registers, so every workgroup pulls the whole gate-gradient row per step.  At B = 16 that row is 128 KB per cell and a CU
// ingests ~11 B/cycle from beyond its L2: 5 us per step.  Here the product is cut along K as well:
//
//   G : 48 + 80 workgroups — Wcat_att^T (1536 columns) / Wcat_dec^T (2560 columns) as bf16 MFMA tiles.  Workgroup (c, q) owns
//       128 output columns x ONE QUARTER of the gate rows (1024 rows = 256 hidden units, unit-major: k = 4 * unit + gate):
//       8 tiles x 4 k-blocks x 4 registers = 128 weight registers per thread, K split again over its 8 waves.  Per step it
//       pulls 32 KB — its quarter of the row, polled by each wave straight into the MFMA's B operands (the row is laid out
//       [k / 8][item][8 k] like the forward's state rows) — and publishes fp32 PARTIAL column sums [item][column].
//   C : 16 + 16 workgroups — the LSTM cells of 64 hidden units each (attention_rnn / decoder_rnn): sum the four partials of
//       their columns, add what else flows into d h (projection gradient; W_q^T dq on the fp32 MFMA; E_h from the other
//       chain), run the cell backward, save the gate gradients and publish them as the next bf16 row.
//   T : B * S workgroups — attention(t) backward, split over encoder positions (the role of decoder_train_bwd_persist.hip;
//       its context gradient is now the sum of 4 + 4 partial rows + the projection's share).
// The decoder_rnn chain (C_d -> G_d -> C_d, two hand-offs per step) needs nothing from the attention chain and free-runs
// ahead; the attention chain per step is  C_a publishes dga(t+1) -> G_a -> partial ya -> T: d ctx(t) -> attention backward
// -> dq(t) -> C_a: W_q^T dq + cell -> dga(t).
// Hand-offs as everywhere in this library: every exchanged value is produced exactly once per pass, the exchange arrays are
// pre-filled with 0xFFFFFFFF and a word that is no longer the sentinel IS the data (sc1 write-through stores, sc1 loads).
#include <stdlib.h>
#include "t2v_common.h"
#include "t2v_kernels.h"

#define Q16_THREADS 512
#define Q16_MAXB 16
#define Q16_MAXT 224                    // (SMAX below: 16- / 32-position slices up to here)
#define Q16_MAXT_LONG 560               // 96-position slices on eight waves from 193 symbols on (six per item: 16 x 6 = 96 workgroups);
                                        // the range of the forward kernel, k_dec_train_persist16<true>
#define Q16_SPIN 400000
#define Q16_SENT 0xFFFFFFFFu
#define Q16_NGA 48                      // (1536 / 128) column groups x 4 row quarters
#define Q16_NGD 80                      // (2560 / 128) x 4
#define Q16_NCA 16                      // 1024 / 64 units
#define Q16_NCD 16
#define Q16_T32 192                     // 32-position slices up to here: 16 items x 6 slices fill the 96 attention workgroups
#define Q16_MAXTWG (T2V_NWG - Q16_NGA - Q16_NGD - Q16_NCA - Q16_NCD)       // 96 attention workgroups
#define Q16_ROW 131072u                 // bytes of one gate-gradient row: 4096 k x 16 items x bf16
#define Q16_NCOLA 1536                  // Wcat_att^T: [h_att recurrent 1024 | ctx 512]
#define Q16_NCOLD 2560                  // Wcat_dec^T: [h_dec recurrent 1024 | E_h 1024 | E_c 512]

struct Q16Args {
    const float* w_ih_att; const float* w_hh_att; const float* w_ih_dec; const float* w_hh_dec;
    const float* wq; const float* wcomb; const float* v;
    const float* memory; const float* XS; const float* CA; const float* CD; const float* GA; const float* GD; const float* AL;
    float* S;                   // (T,B,T_in,128) in: tanh outputs, out: dpre
    const float* dHC;           // (T,B,1536)
    float* DGA; float* DGD; float* DCTX; float* DV;
    void* GXA; void* GXD;       // T rows x 128 KB: bf16 gate-gradient rows of the two cells, MFMA-operand order
    float* PA;                  // (T, 4 quarters, 16 items, 1536) partial Wcat_att^T dga(t)
    float* PD;                  // (T, 4, 16, 2560) partial Wcat_dec^T dgd(t)
    float* DQX;                 // (T,B,S,128) partial dq rows per slice (the caller's DQP)
    float* GPX;                 // window partials of the attention slices
    float* DQT;                 // (T,16,128) dq(t) summed over the slices
    unsigned* err;
    int B, T_in, T, S_sl;
    float p_att, p_dec;
    uint64_t seed;
    const t2v_step_params* step;
    unsigned long long* prof;
};
// per-workgroup time line of TWO consecutive steps (t = T/2 + 1: slots 4..7, t = T/2: slots 0..3) on the chip-wide 100 MHz
// counter: prof[64 + workgroup * 8 + slot] (tools/dbg/persist16_bwd_prof.py follows one turn of the chain through the roles)
#define Q16_RT(SLOT) do { if (a.prof && (t == a.T / 2 || t == a.T / 2 + 1) && threadIdx.x == 0) \
        a.prof[64 + blockIdx.x * 8 + (SLOT) + (t == a.T / 2 ? 0 : 4)] = __builtin_amdgcn_s_memrealtime(); } while (0)

typedef unsigned q16_u32x4 __attribute__((ext_vector_type(4)));
#define Q16_SC1 16
__device__ __forceinline__ __amdgpu_buffer_rsrc_t q16_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ q16_u32x4 q16_ld16u(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, Q16_SC1);
}
__device__ __forceinline__ f32x4 q16_ld16(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, Q16_SC1));
}
__device__ __forceinline__ unsigned q16_ld4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, Q16_SC1);
}
__device__ __forceinline__ void q16_st16(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(q16_u32x4, v), r, (int)off, 0, Q16_SC1);
}
__device__ __forceinline__ void q16_st16u(__amdgpu_buffer_rsrc_t r, unsigned off, q16_u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)off, 0, Q16_SC1);
}
__device__ __forceinline__ void q16_st4(__amdgpu_buffer_rsrc_t r, unsigned off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)off, 0, Q16_SC1);
}
__device__ __forceinline__ bool q16_okf(float v) { return __float_as_uint(v) != Q16_SENT; }
__device__ __forceinline__ bool q16_ok4(f32x4 v) { return q16_okf(v[0]) && q16_okf(v[1]) && q16_okf(v[2]) && q16_okf(v[3]); }
__device__ __forceinline__ bool q16_ok4u(q16_u32x4 v) { return v[0] != Q16_SENT && v[1] != Q16_SENT && v[2] != Q16_SENT && v[3] != Q16_SENT; }
__device__ __forceinline__ bool q16_give_up(int& rounds, unsigned* err, int* flag) {
    if (++rounds > Q16_SPIN || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = 0;
        return true;
    }
    return false;
}
// NQ sixteen-byte words (stride `stride` bytes) polled until none carries the sentinel; per-thread loop (callers sync after it)
template <int NQ>
__device__ __forceinline__ void q16_poll_words(f32x4 (&x)[NQ], __amdgpu_buffer_rsrc_t r, unsigned off, unsigned stride, unsigned* err, int* flag) {
    int rounds = 0;
    for (;;) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) x[i] = q16_ld16(r, off + stride * (unsigned)i);
        bool ok = true;
#pragma unroll
        for (int i = 0; i < NQ; ++i) ok = ok && q16_ok4(x[i]);
        if (ok) break;
        __builtin_amdgcn_s_sleep(2);
        if (q16_give_up(rounds, err, flag)) break;
    }
}

// ======================================================================= G role: a (column group, row quarter) of Wcat^T
// DEC = false: Wcat_att^T — column c < 1024: W_hh_att[.][c] (d h_att), else W_ih_att[.][256 + c - 1024] (d ctx)
// DEC = true : Wcat_dec^T — c < 1024: W_hh_dec[.][c] (d h_dec), c < 2048: W_ih_dec[.][c - 1024] (E_h), else W_ih_dec[.][c - 1024] (E_c)
template <bool DEC>
__device__ __forceinline__ void q16_gemv_role(const Q16Args& a, float* lds, const int j) {
    constexpr int NCOL = DEC ? Q16_NCOLD : Q16_NCOLA;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int cg = j >> 2, q = j & 3;
    const int n = lane & 15, g = lane >> 4;
    const bool live = n < a.B;
    f32x4* red = (f32x4*)lds;                             // [parity 2][wave 8][tile 8][lane 64]
    int* flag = (int*)(lds + 2 * 8 * 8 * 64 * 4);
    const __amdgpu_buffer_rsrc_t rX = q16_rsrc(DEC ? a.GXD : a.GXA), rP = q16_rsrc(DEC ? a.PD : a.PA);
    // ---- weights: tile m, A row = output column col = 128 cg + 16 m + (lane & 15); this wave's k-blocks kb = 32 q + 4 wave + i,
    // 8 consecutive k = 32 kb + 8 g + e -> unit = k >> 2, gate = k & 3 -> gate row gate * 1024 + unit
    q16_u32x4 wreg[8][4];
    {
        // (32-bit buffer offsets, one tile's 32 values in flight at a time: with 64-bit addresses the compiler kept the 32 row offsets
        // of BOTH matrices live across the unrolled tiles, ran out of registers and spilled every loaded value on arrival — 256
        // serialised round trips per launch and 280 B/lane of scratch in the kernel's resource record)
        const __amdgpu_buffer_rsrc_t rHH = q16_rsrc(DEC ? a.w_hh_dec : a.w_hh_att), rIH = q16_rsrc(DEC ? a.w_ih_dec : a.w_ih_att);
        unsigned chain = 0;     // always 0, but data-dependent on the previous tile's last packed word: its address arithmetic cannot be hoisted
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int col = 128 * cg + 16 * m + n;
            const bool hh = 128 * cg + 16 * m < T2V_H;         // (a 16-column tile never straddles the two matrices)
            const unsigned ld = hh ? (unsigned)T2V_H : (DEC ? (unsigned)T2V_KATT : (unsigned)(T2V_PRE + T2V_E));
            const unsigned c0 = hh ? (unsigned)col : (DEC ? (unsigned)(col - T2V_H) : (unsigned)(T2V_PRE + col - T2V_H));
#pragma unroll
            for (int h = 0; h < 2; ++h) {               // two k-blocks = 16 values in flight, then the next two
                float wv[2][8];
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const int i = 2 * h + ii;
                    const unsigned unit0 = (unsigned)(32 * (32 * q + 4 * wave + i) + 8 * g) >> 2;     // k = 4 unit + gate; 8 k = 2 units x 4 gates
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const unsigned row = (unsigned)(e & 3) * T2V_H + unit0 + (unsigned)(e >> 2);
                        const unsigned off = (row * ld + c0) * 4u + chain;
                        wv[ii][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(hh ? rHH : rIH, (int)off, 0, 0));
                    }
                }
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const uint4 u = t2v_pack_bf16x8(make_float4(wv[ii][0], wv[ii][1], wv[ii][2], wv[ii][3]), make_float4(wv[ii][4], wv[ii][5], wv[ii][6], wv[ii][7]));
                    wreg[m][2 * h + ii] = q16_u32x4{u.x, u.y, u.z, u.w};
                }
                asm volatile("v_and_b32 %0, 0, %1" : "=v"(chain) : "v"(wreg[m][2 * h + 1].x));
            }
        }
    }
    if (tid == 0) flag[0] = 1;
    __syncthreads();
    const unsigned off_w = (unsigned)(32 * q + 4 * wave) * 1024u + 16u * (unsigned)lane;
    const int t_last = DEC ? 0 : 1;                       // Wcat_att^T dga(0) feeds nothing
    int nap = 0;
    for (int t = a.T - 1; t >= t_last; --t) {
        Q16_RT(0);
        // ---- this wave's 4 k-blocks of row t, straight into the B operands
        q16_u32x4 x[4];
        {
            for (int i = 0; i < nap; i += 8) __builtin_amdgcn_s_sleep(8);
            int rounds = 0;
            const unsigned off = (unsigned)t * Q16_ROW + off_w;
            for (;;) {
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = q16_ld16u(rX, off + 1024u * (unsigned)i);
                bool ok = true;
#pragma unroll
                for (int i = 0; i < 4; ++i) ok = ok && q16_ok4u(x[i]);
                if (__all(ok || !live)) break;
                __builtin_amdgcn_s_sleep(2);
                if (q16_give_up(rounds, a.err, flag)) break;
            }
            nap = t2v_adapt_nap(nap, rounds);
            if (!live) {
#pragma unroll
                for (int i = 0; i < 4; ++i) x[i] = q16_u32x4{0u, 0u, 0u, 0u};
            }
        }
        Q16_RT(1);
        f32x4* redp = red + (size_t)(t & 1) * (8 * 8 * 64);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(t2v_bf16x8, wreg[m][i]), __builtin_bit_cast(t2v_bf16x8, x[i]), acc, 0, 0, 0);
            redp[(wave * 8 + m) * 64 + lane] = acc;
        }
        __syncthreads();
        if (flag[0] != 1) return;
        // ---- thread (tile m = wave, lane): the 8 K-slices summed; lane (item n, g) holds columns 4 g .. 4 g + 3 of the tile
        {
            const f32x4* rp = redp + wave * 64 + lane;          // K-slice stride: 8 * 64
            const f32x4 s4 = ((rp[0] + rp[512]) + (rp[1024] + rp[1536])) + ((rp[2048] + rp[2560]) + (rp[3072] + rp[3584]));
            if (live) q16_st16(rP, (unsigned)(((t * 4 + q) * 16 + n) * NCOL + 128 * cg + 16 * wave + 4 * g) * 4u, s4);
        }
        Q16_RT(2);
    }
}

// cell backward of one (unit, item): in: d h (all contributions), the running cell gradient dcs; saved gates and cells
struct Q16Cell { float fh, fc, gtc, gf, e0, e1, e2, e3; };
__device__ __forceinline__ Q16Cell q16_cell_factors(float gi, float gf, float gg, float go, float cac, float cprev, uint64_t seed,
                                                    uint32_t st_h, uint32_t st_c, int t, uint32_t idx, float p) {
    Q16Cell c;
    c.fh = t2v_drop_scale(seed, st_h, t, idx, p);
    c.fc = t2v_drop_scale(seed, st_c, t, idx, p);
    if (t > 0) cprev *= t2v_drop_scale(seed, st_c, t - 1, idx, p);
    const float tc = tanhf_(cac);
    c.gtc = go * (1.0f - tc * tc);
    c.gf = gf;
    c.e0 = gg * gi * (1.0f - gi);
    c.e1 = cprev * gf * (1.0f - gf);
    c.e2 = gi * (1.0f - gg * gg);
    c.e3 = tc * go * (1.0f - go);
    return c;
}

// ======================================================================= C role: the cells of 64 hidden units
// thread tid < 256: item b = tid >> 4, units U0 + 4 (tid & 15) .. + 3.  ATT: attention_rnn (d h = E_h(t) + ya_h(t+1) + W_q^T dq(t)),
// else decoder_rnn (d h = dHC_h(t) + yd_rec(t+1)).
template <bool ATT>
__device__ __forceinline__ void q16_cell_role(const Q16Args& a, float* lds, const int j) {
    const uint64_t seed = t2v_step_seed(a.seed, a.step);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int B = a.B, T = a.T;
    float* dqs = lds;                                     // [128 dims][16 items]  dq(t)
    float* wqd = dqs + 128 * 16;                          // [2 K halves][64 units][16 items]  W_q^T dq
    int* flag = (int*)(wqd + 2 * 64 * 16);
    const int U0 = 64 * j;
    const int cb = tid >> 4, uq = tid & 15, U = U0 + 4 * uq;
    const bool cell_thr = tid < 256 && cb < B;
    const __amdgpu_buffer_rsrc_t rPA = q16_rsrc(a.PA), rPD = q16_rsrc(a.PD), rQT = q16_rsrc(a.DQT);
    const __amdgpu_buffer_rsrc_t rX = q16_rsrc(ATT ? a.GXA : a.GXD);
    const float* Gs = ATT ? a.GA : a.GD;
    const float* Cs = ATT ? a.CA : a.CD;
    float* DG = ATT ? a.DGA : a.DGD;
    const float p = ATT ? a.p_att : a.p_dec;
    const uint32_t st_h = ATT ? T2V_RNG_ATT_H : T2V_RNG_DEC_H, st_c = ATT ? T2V_RNG_ATT_C : T2V_RNG_DEC_C;
    // W_q^T tile of this wave (ATT): units U0 + 16 (wave & 3) + (lane & 15), dims 64 (wave >> 2) + 4 jj + (lane >> 4)
    float wqr[16];
    if (ATT) {
#pragma unroll
        for (int jj = 0; jj < 16; ++jj)
            wqr[jj] = a.wq[(size_t)(64 * (wave >> 2) + 4 * jj + (lane >> 4)) * T2V_H + U0 + 16 * (wave & 3) + (lane & 15)];
    }
    if (tid == 0) flag[0] = 1;
    float dcs[4] = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    for (int t = T - 1; t >= 0; --t) {
        Q16_RT(0);
        // ---- everything that is a function of the forward pass alone
        Q16Cell cf[4];
        f32x4 dh4 = {0.f, 0.f, 0.f, 0.f};
        if (cell_thr) {
            const float* gp = Gs + ((size_t)t * B + cb) * T2V_G + U;
            const float4 gi = *(const float4*)gp, gf = *(const float4*)(gp + T2V_H), gg = *(const float4*)(gp + 2 * T2V_H), go = *(const float4*)(gp + 3 * T2V_H);
            const float4 cac = *(const float4*)(Cs + ((size_t)(t + 1) * B + cb) * T2V_H + U);
            const float4 cpv = *(const float4*)(Cs + ((size_t)t * B + cb) * T2V_H + U);
            const uint32_t idx = (uint32_t)cb * T2V_H + U;
            cf[0] = q16_cell_factors(gi.x, gf.x, gg.x, go.x, cac.x, cpv.x, seed, st_h, st_c, t, idx, p);
            cf[1] = q16_cell_factors(gi.y, gf.y, gg.y, go.y, cac.y, cpv.y, seed, st_h, st_c, t, idx + 1, p);
            cf[2] = q16_cell_factors(gi.z, gf.z, gg.z, go.z, cac.z, cpv.z, seed, st_h, st_c, t, idx + 2, p);
            cf[3] = q16_cell_factors(gi.w, gf.w, gg.w, go.w, cac.w, cpv.w, seed, st_h, st_c, t, idx + 3, p);
            if (!ATT) {
                const float4 d = *(const float4*)(a.dHC + ((size_t)t * B + cb) * (T2V_H + T2V_E) + U);
                dh4 = f32x4{d.x, d.y, d.z, d.w};
            }
        }
        // ---- the partial column sums of the G workgroups (4 row quarters each)
        if (cell_thr) {
            if (ATT) {
                f32x4 e[4];            // E_h(t) = W_ih_dec[:, :1024]^T dgd(t): the decoder_rnn chain runs ahead, these are there
                q16_poll_words<4>(e, rPD, (unsigned)(((t * 4) * 16 + cb) * Q16_NCOLD + T2V_H + U) * 4u, 16u * Q16_NCOLD * 4u, a.err, flag);
                dh4 = (e[0] + e[1]) + (e[2] + e[3]);
            }
            if (t < T - 1) {
                f32x4 y[4];
                if (ATT) q16_poll_words<4>(y, rPA, (unsigned)((((t + 1) * 4) * 16 + cb) * Q16_NCOLA + U) * 4u, 16u * Q16_NCOLA * 4u, a.err, flag);
                else q16_poll_words<4>(y, rPD, (unsigned)((((t + 1) * 4) * 16 + cb) * Q16_NCOLD + U) * 4u, 16u * Q16_NCOLD * 4u, a.err, flag);
                dh4 = dh4 + ((y[0] + y[1]) + (y[2] + y[3]));
            }
        }
        Q16_RT(1);
        if (ATT) {
            // ---- dq(t) of all items (summed over the position slices by slice 0 of each item) -> LDS [dim][item]
            {
                const int b = tid >> 5, d4 = tid & 31;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (b < B) {
                    f32x4 vv[1];
                    q16_poll_words<1>(vv, rQT, (unsigned)((t * 16 + b) * T2V_A + 4 * d4) * 4u, 0u, a.err, flag);
                    v = vv[0];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) dqs[(4 * d4 + e) * 16 + b] = v[e];
            }
            Q16_RT(2);
            __syncthreads();
            if (flag[0] != 1) return;
            // ---- W_q^T dq on the fp32 MFMA: wave = (unit tile m = wave & 3, K half), 16 steps of K = 4
            {
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                const float* bp = dqs + (64 * (wave >> 2) + (lane >> 4)) * 16 + (lane & 15);
#pragma unroll
                for (int jj = 0; jj < 16; jj += 2) {
                    acc0 = mfma16x4(wqr[jj], bp[(4 * jj) * 16], acc0);
                    acc1 = mfma16x4(wqr[jj + 1], bp[(4 * jj + 4) * 16], acc1);
                }
                const f32x4 acc = acc0 + acc1;
                float* o = wqd + (wave >> 2) * 1024 + (16 * (wave & 3) + 4 * (lane >> 4)) * 16 + (lane & 15);
                o[0] = acc[0]; o[16] = acc[1]; o[32] = acc[2]; o[48] = acc[3];
            }
            __syncthreads();
            if (cell_thr) {
#pragma unroll
                for (int e = 0; e < 4; ++e) dh4[e] += wqd[(4 * uq + e) * 16 + cb] + wqd[1024 + (4 * uq + e) * 16 + cb];
            }
        } else {
            __syncthreads();
            if (flag[0] != 1) return;
        }
        // ---- cell backward of the 4 units, publish (two 16-byte chunks: unit pairs), save
        if (cell_thr) {
            float dg[4][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float dht = dh4[e] * cf[e].fh;
                const float dct = dcs[e] * cf[e].fc + dht * cf[e].gtc;
                dg[e][0] = dct * cf[e].e0; dg[e][1] = dct * cf[e].e1; dg[e][2] = dct * cf[e].e2; dg[e][3] = dht * cf[e].e3;
                dcs[e] = dct * cf[e].gf;
            }
            if (t >= (ATT ? 1 : 0)) {
                // k = 4 * unit + gate; chunk (k / 8, item) = 8 consecutive k = units (2 p, 2 p + 1) x 4 gates
                const unsigned kg = (unsigned)(U >> 1);             // = k / 8 of unit U, gate 0
                const unsigned o = (unsigned)t * Q16_ROW + (kg * 16u + (unsigned)cb) * 16u;
                const uint4 c0 = t2v_pack_bf16x8(make_float4(dg[0][0], dg[0][1], dg[0][2], dg[0][3]), make_float4(dg[1][0], dg[1][1], dg[1][2], dg[1][3]));
                const uint4 c1 = t2v_pack_bf16x8(make_float4(dg[2][0], dg[2][1], dg[2][2], dg[2][3]), make_float4(dg[3][0], dg[3][1], dg[3][2], dg[3][3]));
                q16_st16u(rX, o, q16_u32x4{c0.x, c0.y, c0.z, c0.w});
                q16_st16u(rX, o + 256u, q16_u32x4{c1.x, c1.y, c1.z, c1.w});
            }
            float* o = DG + ((size_t)t * B + cb) * T2V_G + U;
#pragma unroll
            for (int r = 0; r < 4; ++r) *(float4*)(o + r * T2V_H) = make_float4(dg[0][r], dg[1][r], dg[2][r], dg[3][r]);
        }
        Q16_RT(3);
    }
}

// ======================================================================= T role: attention(t) backward of (item b, slice s)
// The position-split body of decoder_train_bwd_persist.hip (softmax / tanh / fused-location-filter backward; operands that do
// not change over the pass resident in registers, the cumulative-weights gradient in LDS), with the context gradient taken
// from the partial rows of the G workgroups: d ctx(t) = sum_q PA(t+1)[q][ctx] + sum_q PD(t)[q][E_c] + dHC(t)[ctx].
// NWV: waves that compute — 4 (16- / 32-position slices) or 8 (96-position slices: the form decoder_train_bwd_persist.hip
// built in round 4; W_comb^T operand tile in LDS, dpT row stride 17 mod 32, window-partial rows of 128 floats per channel).
template <int JS, int NWV = 4>
__device__ __forceinline__ void q16_attention_role(const Q16Args& a, float* lds, const int b, const int s) {
    constexpr int NJT = JS / 16;
    constexpr int PW = JS + 30;
    constexpr int GPW = PW <= 64 ? 64 : 128;
    constexpr int NRG = 2 * NWV;                     // row groups of 32 lanes in the dpre loop
    constexpr int DPS = JS == 96 ? JS + 17 : JS + 1;
    static_assert(JS % NRG == 0 && JS % NWV == 0 && PW <= GPW, "slice geometry");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool act = tid < 64 * NWV;
    const int g = lane >> 4, c16 = lane & 15;
    const int B = a.B, Tp = a.T_in, T = a.T, S = a.S_sl, j0 = s * JS;
    const int Tcap = (Tp + 15) & ~15;
    const int nown = min(JS, Tp - j0);
    // ---- LDS carve
    float* gfull0 = lds;                      // [Tcap]
    float* gfull1 = gfull0 + Tcap;            // [Tcap]
    float* alf = gfull1 + Tcap;               // [Tcap]
    float* gcum = alf + Tcap;                 // [Tcap] running cumulative-weights gradient (this workgroup's copy)
    float* dctx = gcum + Tcap;                // [512]
    float* de = dctx + T2V_E;                 // [JS]
    float* red = de + JS;                     // [1 + JS/NWV][4 NWV]
    float* dpT = red + (1 + JS / NWV) * 4 * NWV;     // [128][JS+1]
    float* Tl = dpT + T2V_A * DPS;            // [64][JS+1]
    float* rq = Tl + 64 * (JS + 1);           // [NRG][128] (also: the 32 row partials of the dot product)
    float* rv = rq + NRG * T2V_A;             // [NRG][128]
    int* flag = (int*)(rv + NRG * T2V_A);
    constexpr bool AREG_LDS = NWV == 8;
    float* wcs = (float*)(flag + 40);         // [64 rows (c,k)][132] when AREG_LDS
    const __amdgpu_buffer_rsrc_t rPA = q16_rsrc(a.PA), rPD = q16_rsrc(a.PD), rQ = q16_rsrc(a.DQX), rP = q16_rsrc(a.GPX), rQT = q16_rsrc(a.DQT);
    // ---- operands resident for the whole pass
    float4 m0[JS / NWV], m1[JS / NWV];
    float areg[AREG_LDS ? 1 : 32];
    float4 vd4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act) {
        const int d4 = tid & 31;
#pragma unroll
        for (int r = 0; r < JS / NWV; ++r) {
            const int jl = wave + NWV * r;
            const float* mrow = a.memory + ((size_t)b * Tp + j0 + (jl < nown ? jl : 0)) * T2V_E + lane * 4;
            m0[r] = *(const float4*)mrow;
            m1[r] = *(const float4*)(mrow + 256);
        }
        if constexpr (!AREG_LDS) {
            const float4* wp = (const float4*)(a.wcomb + T2V_A * 64 + (16 * (wave & 3) + c16) * 128 + 32 * g);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float4 w4 = wp[u];
                areg[4 * u + 0] = w4.x; areg[4 * u + 1] = w4.y; areg[4 * u + 2] = w4.z; areg[4 * u + 3] = w4.w;
            }
        }
        vd4 = *(const float4*)(a.v + 4 * d4);
    }
    if (AREG_LDS)
        for (int i = tid; i < 64 * 128; i += Q16_THREADS)          // row stride 132, 33 floats per k-group: conflict-free operand reads
            wcs[(i >> 7) * 132 + ((i & 127) >> 5) * 33 + (i & 31)] = a.wcomb[T2V_A * 64 + i];
    for (int j = tid; j < Tcap; j += Q16_THREADS) gcum[j] = 0.f;
    if (tid == 0) flag[0] = 1;
    float dvacc = 0.f;                          // tid < 128: running dv[tid] of this slice
    int nap = 0;
    __syncthreads();

    for (int t = T - 1; t >= 0; --t) {
        // (thread-derived indices are recomputed per step from an opaque copy: hoisted, they are spilled next to the registers of
        // memory rows, and every reload is a drain of the wave's memory queue)
        int tid_op = threadIdx.x;
        asm volatile("" : "+v"(tid_op));
        const int tid = tid_op, lane = tid & 63, wave = tid >> 6;
        const bool act = tid < 64 * NWV;
        const int g = lane >> 4, c16 = lane & 15;
        const int d4 = tid & 31, rg = (tid >> 5) & (NRG - 1);
        Q16_RT(0);
        // ---- operands that do not wait for the context gradient: tanh outputs, alpha(t), ctx(t), window partials of step t+1,
        // and the early part of the context gradient (decoder_rnn's E_c(t) — that chain runs ahead — + the projection's share)
        float4 sreg[JS / NRG];
        float2 ctx2 = make_float2(0.f, 0.f);
        f32x4 dc_early = {0.f, 0.f, 0.f, 0.f};
        if (act) {
            const float* sp = a.S + (((size_t)t * B + b) * Tp + j0) * T2V_A + 4 * d4;
#pragma unroll
            for (int i = 0; i < JS / NRG; ++i) {
                const int jl = rg + NRG * i;
                sreg[i] = *(const float4*)(sp + (size_t)min(jl, nown - 1) * T2V_A);
            }
            if (tid < 256) ctx2 = *(const float2*)(a.XS + ((size_t)(t + 1) * B + b) * T2V_XW + T2V_H + 2 * tid);
        }
        if (tid < 128) {
            const float4 hc = *(const float4*)(a.dHC + ((size_t)t * B + b) * (T2V_H + T2V_E) + T2V_H + 4 * tid);
            f32x4 e[4];
            q16_poll_words<4>(e, rPD, (unsigned)(((t * 4) * 16 + b) * Q16_NCOLD + 2 * T2V_H + 4 * tid) * 4u, 16u * Q16_NCOLD * 4u, a.err, flag);
            dc_early = ((e[0] + e[1]) + (e[2] + e[3])) + f32x4{hc.x, hc.y, hc.z, hc.w};
        }
        float dot_g = 0.f;
        for (int j = tid; j < Tp; j += Q16_THREADS) {
            float gp = 0.f, gc = gcum[j];
            if (t < T - 1) {
                const int lo = max(0, (j + 15 - PW + JS) / JS), hi = min(S - 1, (j + 15) / JS);
                for (int sp2 = lo; sp2 <= hi; ++sp2) {
                    const int jj = j - sp2 * JS + 15;
                    if (jj < 0 || jj >= PW) continue;
                    const unsigned off = (unsigned)((((t + 1) * B + b) * S + sp2) * (2 * GPW) + jj) * 4u;
                    unsigned x0, x1;
                    int spins = 0;
                    for (;;) {          // published at the end of the previous reverse step: almost always there
                        x0 = q16_ld4(rP, off);
                        x1 = q16_ld4(rP, off + 4u * GPW);
                        if (x0 != Q16_SENT && x1 != Q16_SENT) break;
                        __builtin_amdgcn_s_sleep(1);
                        if (q16_give_up(spins, a.err, flag)) break;
                    }
                    gp += __uint_as_float(x0);
                    gc += __uint_as_float(x1);
                }
            }
            gcum[j] = gc;
            gfull0[j] = gp;
            gfull1[j] = gc;
            const float al = a.AL[((size_t)(t + 1) * B + b) * Tp + j];
            alf[j] = al;
            dot_g = fmaf(al, gp + gc, dot_g);
        }
        // ---- the context gradient of this item: the four partial rows of Wcat_att^T dga(t+1) (16 bytes x 4 per thread of waves
        // 0 and 1), nap first — this hand-off is on the chain of every reverse step
        if (tid < 128) {
            f32x4 dc = dc_early;
            if (t < T - 1) {
                for (int i = 0; i < nap; i += 8) __builtin_amdgcn_s_sleep(8);
                const unsigned off = (unsigned)((((t + 1) * 4) * 16 + b) * Q16_NCOLA + T2V_H + 4 * tid) * 4u;
                f32x4 y[4];
                int rounds = 0;
                for (;;) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) y[i] = q16_ld16(rPA, off + 16u * Q16_NCOLA * 4u * (unsigned)i);
                    bool ok = true;
#pragma unroll
                    for (int i = 0; i < 4; ++i) ok = ok && q16_ok4(y[i]);
                    if (__all(ok)) break;
                    __builtin_amdgcn_s_sleep(2);
                    if (q16_give_up(rounds, a.err, flag)) break;
                }
                nap = t2v_adapt_nap(nap, rounds);
                dc = dc + ((y[0] + y[1]) + (y[2] + y[3]));
            }
            *(f32x4*)(dctx + 4 * tid) = dc;
            if (s == 0) *(f32x4*)(a.DCTX + ((size_t)t * B + b) * T2V_E + 4 * tid) = dc;      // saved copy for the d_memory product
        }
        __syncthreads();
        if (flag[0] != 1) return;
        Q16_RT(1);
        // ---- dot = dctx·ctx_t + sum_j alpha_j (Gprev_j + Gcum_j); dalpha of the own positions = dctx·memory_j + G_j
        {
            float dotp = dot_g;
            if (tid < 256) dotp += dctx[2 * tid] * ctx2.x + dctx[2 * tid + 1] * ctx2.y;
            dotp = row16_sum(dotp);
            // 32 row partials (8 waves x 4 rows): waves 4..7 carry only their share of dot_g
            if (c16 == 0) rq[4 * wave + g] = dotp;
            if (act) {
                const float4 d0 = *(const float4*)(dctx + lane * 4), d1 = *(const float4*)(dctx + 256 + lane * 4);
#pragma unroll
                for (int r = 0; r < JS / NWV; ++r) {
                    float acc = m0[r].x * d0.x;
                    acc = fmaf(m0[r].y, d0.y, acc); acc = fmaf(m0[r].z, d0.z, acc); acc = fmaf(m0[r].w, d0.w, acc);
                    acc = fmaf(m1[r].x, d1.x, acc); acc = fmaf(m1[r].y, d1.y, acc);
                    acc = fmaf(m1[r].z, d1.z, acc); acc = fmaf(m1[r].w, d1.w, acc);
                    acc = row16_sum(acc);
                    if (c16 == 0) red[(1 + r) * 4 * NWV + 4 * wave + g] = acc;
                }
            }
        }
        __syncthreads();
        if (tid < JS) {
            float dsum = 0.f;
#pragma unroll
            for (int u = 0; u < 32; ++u) dsum += rq[u];
            const int wv = tid % NWV, r = tid / NWV;              // position tid = wv + NWV r
            const float* rr = red + (1 + r) * 4 * NWV + 4 * wv;
            const float dalv = ((rr[0] + rr[1]) + (rr[2] + rr[3])) + gfull0[j0 + min(tid, nown - 1)] + gfull1[j0 + min(tid, nown - 1)];
            de[tid] = tid < nown ? alf[j0 + tid] * (dalv - dsum) : 0.f;
        }
        __syncthreads();
        // ---- through v·tanh(.): dpre, partial dq / dv
        if (act) {
            float4 dq = make_float4(0.f, 0.f, 0.f, 0.f), dv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < JS / NRG; ++i) {
                const int jl = rg + NRG * i;
                const float dej = de[jl];
                const float4 sv = sreg[i];
                float4 dp;
                dp.x = dej * vd4.x * (1.0f - sv.x * sv.x); dp.y = dej * vd4.y * (1.0f - sv.y * sv.y);
                dp.z = dej * vd4.z * (1.0f - sv.z * sv.z); dp.w = dej * vd4.w * (1.0f - sv.w * sv.w);
                sreg[i] = dp;           // the saved copy (operand of the d W_comb / d memory_layer products) leaves AFTER the hand-off
                dq.x += dp.x; dq.y += dp.y; dq.z += dp.z; dq.w += dp.w;
                dv.x = fmaf(dej, sv.x, dv.x); dv.y = fmaf(dej, sv.y, dv.y); dv.z = fmaf(dej, sv.z, dv.z); dv.w = fmaf(dej, sv.w, dv.w);
                dpT[(4 * d4 + 0) * DPS + jl] = dp.x; dpT[(4 * d4 + 1) * DPS + jl] = dp.y;
                dpT[(4 * d4 + 2) * DPS + jl] = dp.z; dpT[(4 * d4 + 3) * DPS + jl] = dp.w;
            }
            *(float4*)&rq[rg * T2V_A + 4 * d4] = dq;
            *(float4*)&rv[rg * T2V_A + 4 * d4] = dv;
        }
        __syncthreads();
        if (tid < T2V_A) {
            const float* p = rq + tid;
            float q = ((p[0] + p[T2V_A]) + (p[2 * T2V_A] + p[3 * T2V_A])) + ((p[4 * T2V_A] + p[5 * T2V_A]) + (p[6 * T2V_A] + p[7 * T2V_A]));
            const float* p2 = rv + tid;
            float vv = ((p2[0] + p2[T2V_A]) + (p2[2 * T2V_A] + p2[3 * T2V_A])) + ((p2[4 * T2V_A] + p2[5 * T2V_A]) + (p2[6 * T2V_A] + p2[7 * T2V_A]));
            if (NRG > 8) {
                p += 8 * T2V_A; p2 += 8 * T2V_A;
                q += ((p[0] + p[T2V_A]) + (p[2 * T2V_A] + p[3 * T2V_A])) + ((p[4 * T2V_A] + p[5 * T2V_A]) + (p[6 * T2V_A] + p[7 * T2V_A]));
                vv += ((p2[0] + p2[T2V_A]) + (p2[2 * T2V_A] + p2[3 * T2V_A])) + ((p2[4 * T2V_A] + p2[5 * T2V_A]) + (p2[6 * T2V_A] + p2[7 * T2V_A]));
            }
            q16_st4(rQ, (unsigned)(((t * B + b) * S + s) * T2V_A + tid) * 4u, q);       // partial row (the d W_q GEMM reads them later)
            dvacc += vv;
            if (s == 0) {
                // slice 0 of an item sums the S partial rows in slice order and publishes ONE row per item (all partial rows are
                // requested in ONE round: a round trip per slice would cost 0.45 us each)
                constexpr int SMAX = Q16_MAXT / 16;             // 14 slices at most
                const unsigned off0 = (unsigned)(((t * B + b) * S) * T2V_A + tid) * 4u;
                unsigned x[SMAX];
                int spins = 0;
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int s2 = 1; s2 < SMAX; ++s2) x[s2] = q16_ld4(rQ, off0 + (unsigned)(min(s2, S - 1) * T2V_A) * 4u);
#pragma unroll
                    for (int s2 = 1; s2 < SMAX; ++s2) ok = ok && (s2 >= S || x[s2] != Q16_SENT);
                    if (__all(ok)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (q16_give_up(spins, a.err, flag)) break;
                }
                float tot = q;
#pragma unroll
                for (int s2 = 1; s2 < SMAX; ++s2) tot += s2 < S ? __uint_as_float(x[s2]) : 0.f;
                q16_st4(rQT, (unsigned)((t * 16 + b) * T2V_A + tid) * 4u, tot);          // the attention_rnn cell workgroups wait for this
            }
        }
        Q16_RT(2);
        if (act) {      // dpre rows: 8 KB of stores that must not sit in this CU's memory pipe in front of the dq words above
            float* sp = a.S + (((size_t)t * B + b) * Tp + j0) * T2V_A + 4 * d4;
#pragma unroll
            for (int i = 0; i < JS / NRG; ++i) {
                const int jl = rg + NRG * i;
                if (jl < nown) *(float4*)(sp + (size_t)jl * T2V_A) = sreg[i];
            }
        }
        // ---- through the fused location filter on MFMA: T[(c,k)][jl] = sum_d W_comb[d][(c,k)] dpre[jl][d], K = 128
        if (act) {
#pragma unroll
            for (int jt = (NWV == 8 ? (wave >> 2) : 0); jt < NJT; jt += NWV / 4) {     // (8 waves: waves 4..7 take the odd position tiles)
                f32x4 ac4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) ac4[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < 32; ++st) {
                    const float av = AREG_LDS ? wcs[(16 * (wave & 3) + c16) * 132 + 33 * g + st] : areg[AREG_LDS ? 0 : st];
                    ac4[st & 3] = mfma16x4(av, dpT[(4 * st + g) * DPS + 16 * jt + c16], ac4[st & 3]);
                }
                const f32x4 acc = (ac4[0] + ac4[1]) + (ac4[2] + ac4[3]);
#pragma unroll
                for (int r = 0; r < 4; ++r) Tl[(16 * (wave & 3) + 4 * g + r) * (JS + 1) + 16 * jt + c16] = acc[r];
            }
        }
        __syncthreads();
        // ---- gradient wrt the alignment window of this slice -> the slices of step t-1 (their window partials)
        if (tid < 2 * GPW && t > 0) {
            const int c = tid / GPW, jj = tid % GPW;
            if (jj < PW) {
                float tt[T2V_KS];
#pragma unroll
                for (int k = 0; k < T2V_KS; ++k) {
                    const int jl = jj - k;
                    const float tv = Tl[(32 * c + k) * (JS + 1) + min(max(jl, 0), JS - 1)];
                    tt[k] = (jl >= 0 && jl < JS) ? tv : 0.f;
                }
                float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
#pragma unroll
                for (int k = 0; k + 3 < T2V_KS; k += 4) { acc0 += tt[k]; acc1 += tt[k + 1]; acc2 += tt[k + 2]; acc3 += tt[k + 3]; }
                acc0 += tt[28]; acc1 += tt[29]; acc2 += tt[30];
                q16_st4(rP, (unsigned)(((t * B + b) * S + s) * (2 * GPW) + c * GPW + jj) * 4u, (acc0 + acc1) + (acc2 + acc3));
            }
        }
        __syncthreads();
        Q16_RT(3);
    }
    if (tid < T2V_A) a.DV[((size_t)b * S + s) * T2V_A + tid] = dvacc;
}

template <bool LONG>          // the attention role on 96-position slices (T_in > Q16_T32); the other roles are the same
__global__ __launch_bounds__(Q16_THREADS) void k_bwd_persist16(Q16Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wg = blockIdx.x;
    const int S = a.S_sl, NT = a.B * S;
#if defined(Q16_ONLY_T)
    if (wg < NT) {
        if (LONG) q16_attention_role<96, 8>(a, lds, wg / S, wg % S);
        else if (a.T_in <= 96) q16_attention_role<16>(a, lds, wg / S, wg % S);
        else q16_attention_role<32>(a, lds, wg / S, wg % S);
    }
#elif defined(Q16_ONLY_G)
    if (wg < 48) q16_gemv_role<false>(a, lds, wg); else q16_gemv_role<true>(a, lds, wg - 48);
#elif defined(Q16_ONLY_C)
    if (wg < 16) q16_cell_role<true>(a, lds, wg); else q16_cell_role<false>(a, lds, wg - 16);
#else
    if (wg < Q16_MAXTWG) {
        if (wg >= NT) return;
        if (LONG) q16_attention_role<96, 8>(a, lds, wg / S, wg % S);
        else if (a.T_in <= 96) q16_attention_role<16>(a, lds, wg / S, wg % S);
        else q16_attention_role<32>(a, lds, wg / S, wg % S);
        return;
    }
    const int j = wg - Q16_MAXTWG;
    if (j < Q16_NGA) q16_gemv_role<false>(a, lds, j);
    else if (j < Q16_NGA + Q16_NGD) q16_gemv_role<true>(a, lds, j - Q16_NGA);
    else if (j < Q16_NGA + Q16_NGD + Q16_NCA) q16_cell_role<true>(a, lds, j - Q16_NGA - Q16_NGD);
    else q16_cell_role<false>(a, lds, j - Q16_NGA - Q16_NGD - Q16_NCA);
#endif
}

// sentinel fill (16 bytes per thread and iteration)
__global__ __launch_bounds__(256) void k_q16_fill(uint4* p, size_t n16) {
    const uint4 s = {Q16_SENT, Q16_SENT, Q16_SENT, Q16_SENT};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = s;
}

// slices of the attention role: 16 positions up to 96 symbols, 32 up to 192, else 96 (B * S attention workgroups must fit into
// 96: with at most six slices per item B = 16 always does)
static inline int q16_js(int T_in) { return T_in <= 96 ? 16 : T_in <= Q16_T32 ? 32 : 96; }
static inline int q16_slices(int T_in) { const int js = q16_js(T_in); return (T_in + js - 1) / js; }
extern "C" int t2v_decoder_bwd_persist16_slices(int T_in) { return T_in < 1 ? 0 : q16_slices(T_in); }

#define Q16_LDS_MAX (160 * 1024)
static size_t q16_lds_bytes(int T_in) {
    const size_t grole = 2 * 8 * 8 * 64 * 4 + 4;
    const size_t crole = 128 * 16 + 2 * 64 * 16 + 4;
    const size_t Tcap = (size_t)((T_in + 15) / 16) * 16, JS = (size_t)q16_js(T_in), NWV = JS == 96 ? 8 : 4;
    const size_t trole = 4 * Tcap + T2V_E + JS + (1 + JS / NWV) * 4 * NWV + T2V_A * (JS == 96 ? JS + 17 : JS + 1) + 64 * (JS + 1) +
                         2 * 2 * NWV * T2V_A + 40 + (NWV == 8 ? 64 * 132 : 0);
    size_t m = grole > crole ? grole : crole;
    m = m > trole ? m : trole;
    return sizeof(float) * m;
}
static int q16_device_ok(size_t lds, bool lng) {
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
        if (hipFuncSetAttribute((const void*)k_bwd_persist16<false>, hipFuncAttributeMaxDynamicSharedMemorySize, Q16_LDS_MAX) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_bwd_persist16<true>, hipFuncAttributeMaxDynamicSharedMemorySize, Q16_LDS_MAX) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
        attr_set = true;
    }
    int nblk = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, lng ? (const void*)k_bwd_persist16<true> : (const void*)k_bwd_persist16<false>,
                                                     Q16_THREADS, lds) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return nblk >= 1;
}
extern "C" int t2v_decoder_bwd_persist16_supported(int B, int T_in) {
    if (!(B >= 1 && B <= Q16_MAXB && T_in >= 1 && T_in <= Q16_MAXT_LONG)) return 0;
    if (B * q16_slices(T_in) > Q16_MAXTWG || q16_lds_bytes(T_in) > Q16_LDS_MAX) return 0;
    return q16_device_ok(q16_lds_bytes(T_in), T_in > Q16_T32);
}
// layout of `scratch` (floats): GXA | GXD | PA | PD | GPX | DQT
static void q16_layout(int B, int T_in, int T_out, size_t (&n)[6]) {
    const size_t S = (size_t)q16_slices(T_in);
    n[0] = (size_t)T_out * (Q16_ROW / 4);
    n[1] = n[0];
    n[2] = (size_t)T_out * 4 * 16 * Q16_NCOLA;
    n[3] = (size_t)T_out * 4 * 16 * Q16_NCOLD;
    n[4] = (size_t)T_out * B * S * 2 * (q16_js(T_in) + 30 <= 64 ? 64 : 128);
    n[5] = (size_t)T_out * 16 * T2V_A;
}
extern "C" long t2v_decoder_bwd_persist16_scratch_floats(int B, int T_in, int T_out) {
    if (B < 1 || B > Q16_MAXB || T_in < 1 || T_in > Q16_MAXT_LONG || T_out < 1) return 0;
    size_t n[6];
    q16_layout(B, T_in, T_out, n);
    return (long)(n[0] + n[1] + n[2] + n[3] + n[4] + n[5]);
}
// float offset, inside `scratch`, of dq(t) summed over the position slices — (T_out, 16, 128), rows of items >= B unused
extern "C" long t2v_decoder_bwd_persist16_dq_offset(int B, int T_in, int T_out) {
    if (B < 1 || B > Q16_MAXB || T_in < 1 || T_in > Q16_MAXT_LONG || T_out < 1) return -1;
    size_t n[6];
    q16_layout(B, T_in, T_out, n);
    return (long)(n[0] + n[1] + n[2] + n[3] + n[4]);
}

// 1 when the pass can run at this T_out as well: every array of the layout above (and DQP) is addressed with 31-bit buffer offsets.
// What q16_run checks — exported so that a caller can choose the launch-per-step pass BEFORE the forward commits to this one
// (the PD array reaches 2 GB at T_out = 3 277).
static bool q16_offsets_ok(int B, int T_in, int T_out) {
    size_t n[6];
    q16_layout(B, T_in, T_out, n);
    for (int i = 0; i < 6; ++i)
        if (n[i] * 4 >= 0x7fffffffull) return false;
    return (size_t)T_out * B * q16_slices(T_in) * 128 * 4 < 0x7fffffffull;
}
extern "C" int t2v_decoder_bwd_persist16_fits(int B, int T_in, int T_out) {
    if (T_out < 1 || !t2v_decoder_bwd_persist16_supported(B, T_in)) return 0;
    return q16_offsets_ok(B, T_in, T_out) ? 1 : 0;
}

// The preparation of the pass (error word, sentinel fills: 1.4 MB per time step) as a call of its own: it needs nothing but the
// buffers, so a training step issues it on a side stream right behind the decoder forward, next to the Postnet
extern "C" int t2v_decoder_bwd_persistent16_prepare(float* DQP, float* scratch, uint32_t* err_word, int B, int T_in, int T_out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!DQP || !scratch || !err_word || !t2v_decoder_bwd_persist16_fits(B, T_in, T_out)) return T2V_ERR_ARG;
    if (((uintptr_t)scratch & 15) || ((uintptr_t)DQP & 15)) return T2V_ERR_ARG;
    size_t n[6];
    q16_layout(B, T_in, T_out, n);
    const size_t n_dq = (size_t)T_out * B * q16_slices(T_in) * 128;
    (void)hipMemsetAsync(err_word, 0, sizeof(uint32_t), stream);
    k_q16_fill<<<2048, 256, 0, stream>>>((uint4*)scratch, (n[0] + n[1] + n[2] + n[3] + n[4] + n[5]) / 4);
    k_q16_fill<<<256, 256, 0, stream>>>((uint4*)DQP, n_dq / 4);
    return t2v_check_launch();
}

static int q16_run(const t2v_dec_train_persist_weights* w, const t2v_dec_train_bufs* s, const float* dHC,
                   float* DGA, float* DGD, float* DCTX, float* DV, float* DQP, float* scratch,
                   uint32_t* err_word, int B, int T_in, int T_out, float p_att, float p_dec, uint64_t seed,
                   void* stream_, bool prepare) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!w || !s || !dHC || !DGA || !DGD || !DCTX || !DV || !DQP || !scratch || !err_word) return T2V_ERR_ARG;
    if (!t2v_decoder_bwd_persist16_supported(B, T_in) || T_out < 1) return T2V_ERR_ARG;
    if (!w->w_ih_att || !w->w_hh_att || !w->w_ih_dec || !w->w_hh_dec || !w->wq || !w->wcomb || !w->v) return T2V_ERR_ARG;
    if (!s->memory || !s->XS || !s->CA || !s->CD || !s->GA || !s->GD || !s->AL || !s->S) return T2V_ERR_ARG;
    size_t n[6];
    q16_layout(B, T_in, T_out, n);
    const int S = q16_slices(T_in);
    if (((uintptr_t)scratch & 15) || ((uintptr_t)DQP & 15)) return T2V_ERR_ARG;
    if (!q16_offsets_ok(B, T_in, T_out)) return T2V_ERR_ARG;              // 31-bit buffer offsets
    if (prepare) {
        const int rc = t2v_decoder_bwd_persistent16_prepare(DQP, scratch, err_word, B, T_in, T_out, stream_);
        if (rc != T2V_OK) return rc;
    }
    Q16Args a;
    a.w_ih_att = w->w_ih_att; a.w_hh_att = w->w_hh_att; a.w_ih_dec = w->w_ih_dec; a.w_hh_dec = w->w_hh_dec;
    a.wq = w->wq; a.wcomb = w->wcomb; a.v = w->v;
    a.memory = s->memory; a.XS = s->XS; a.CA = s->CA; a.CD = s->CD; a.GA = s->GA; a.GD = s->GD; a.AL = s->AL; a.S = s->S;
    a.dHC = dHC; a.DGA = DGA; a.DGD = DGD; a.DCTX = DCTX; a.DV = DV;
    float* p = scratch;
    a.GXA = p; p += n[0];
    a.GXD = p; p += n[1];
    a.PA = p; p += n[2];
    a.PD = p; p += n[3];
    a.GPX = p; p += n[4];
    a.DQT = p;
    a.DQX = DQP;
    a.err = err_word;
    a.B = B; a.T_in = T_in; a.T = T_out; a.S_sl = S; a.p_att = p_att; a.p_dec = p_dec; a.seed = seed;
    a.step = t2v_step_for(stream);
    a.prof = g_t2v_prof;
    if (T_in > Q16_T32) k_bwd_persist16<true><<<T2V_NWG, Q16_THREADS, q16_lds_bytes(T_in), stream>>>(a);
    else k_bwd_persist16<false><<<T2V_NWG, Q16_THREADS, q16_lds_bytes(T_in), stream>>>(a);
    return t2v_check_launch();
}

extern "C" int t2v_decoder_bwd_persistent16(const t2v_dec_train_persist_weights* w, const t2v_dec_train_bufs* s, const float* dHC,
                                            float* DGA, float* DGD, float* DCTX, float* DV, float* DQP, float* scratch,
                                            uint32_t* err_word, int B, int T_in, int T_out, float p_att, float p_dec, uint64_t seed,
                                            void* stream_) {
    return q16_run(w, s, dHC, DGA, DGD, DCTX, DV, DQP, scratch, err_word, B, T_in, T_out, p_att, p_dec, seed, stream_, true);
}
// ... the pass without its preparation (scratch / DQP / err_word as handed to t2v_decoder_bwd_persistent16_prepare)
extern "C" int t2v_decoder_bwd_persistent16_prepared(const t2v_dec_train_persist_weights* w, const t2v_dec_train_bufs* s, const float* dHC,
                                                     float* DGA, float* DGD, float* DCTX, float* DV, float* DQP, float* scratch,
                                                     uint32_t* err_word, int B, int T_in, int T_out, float p_att, float p_dec, uint64_t seed,
                                                     void* stream_) {
    return q16_run(w, s, dHC, DGA, DGD, DCTX, DV, DQP, scratch, err_word, B, T_in, T_out, p_att, p_dec, seed, stream_, false);
}
