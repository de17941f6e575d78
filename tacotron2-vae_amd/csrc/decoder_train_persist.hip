// Teacher-forced decoder recurrence, forward, as ONE persistent launch (reference Decoder.forward's time loop
// model.py:415-421 -> Decoder.decode 346-389 -> Attention.forward 67-88): 256 workgroups x 512 threads stay resident for
// all T_out steps, the 67 MB of LSTM weights live in registers, nothing is streamed per step.  The launch-per-step path
// (decoder_fwd.hip + attn_fwd.hip: 2 launches and a 67 MB weight stream per step) stays the general path; this kernel
// serves B <= 6 and T_in <= 224 — the headline training shape (B = 6 per GPU, T_in = 84).
//
// Roles (one workgroup per CU, all co-resident):
//   T  : workgroups [0, 8B)   — location-sensitive attention of item b = wg / 8, attention dims [16s, 16s + 16) and context
//        columns [64s, 64s + 64), s = wg % 8.  W_q slice, memory / processed-memory slices and the alignment window stay
//        in LDS for all steps (as in decoder_persist.hip).
//   L  : workgroups [8B, 256) — LSTM rows of BOTH cells: workgroup j owns hidden units [j*1024/NL, (j+1)*1024/NL) (4 or 5
//        units = 16 or 20 gate rows per cell), weights in VGPRs (thread = (gate, 1/128 of K): 60 + 100 registers).
// Teacher forcing takes decoder_rnn off the critical path: attention_rnn(t+1) needs only h_att(t) and ctx(t) (the Prenet
// term gpre is precomputed for all steps), so the per-step chain is
//     A-gemv(t) -> h_att hop -> attention(t) -> ctx hop -> A-gemv(t+1)
// while every L workgroup runs D-gemv(t-1) and its own gathers in the shadow of attention(t).
//
// Hand-off = the saved activations themselves.  Every recurrent value is produced exactly once per pass, so it needs no
// tag and no flag: the exchange buffer G (one row per step, laid out [plane][k][4 items] so that a consumer's 16-byte
// load is an LDS-ready operand) is pre-filled with a NaN sentinel (0xFFFFFFFF, never produced by arithmetic); producers
// store with sc1 (write-through), consumers poll their own words with sc1 loads until none is the sentinel.  4 bytes per
// value on the wire instead of an 8-byte {value, tag} granule, 16-byte loads, no memory ordering needed (MI355X guide
// G16 form R2 with an implicit tag).  The partial energies of the 8 attention slices of an item travel the same way (EX).
// The backward pass reads the usual arena (XS, CA, CD, GA, GD, AL, ACUM, S), written here with plain stores.
#include "t2v_common.h"
#include "t2v_kernels.h"

#define PT_THREADS 512
#define PT_MAXB 6
#define PT_MAXT 224
#define PT_MAXU 5
#define PT_SPIN 1500000u
#define PT_SENT 0xFFFFFFFFu
#define PT_JA (T2V_KATT / 128)       // 12 k per thread, attention_rnn  [h_att | ctx]
#define PT_JD (T2V_XW / 128)         // 20 k per thread, decoder_rnn    [h_att | ctx | h_dec]

struct PTArgs {
    const float* w_ih_att; const float* w_hh_att; const float* w_ih_dec; const float* w_hh_dec;
    const float* bias_dec; const float* wq; const float* wcomb; const float* v;
    const float* gpre; const float* memory; const float* pm; const int32_t* lengths;
    float* XS; float* CA; float* CD; float* GA; float* GD; float* AL; float* ACUM; float* S;
    float* G;                 // (T+2) rows x NP planes x 2560 k x 4 items, sentinel-filled
    float* EX;                // T x B x 8 x Tcap partial energies, sentinel-filled
    unsigned* err;
    int B, T_in, T_out;
    float p_att, p_dec;
    uint64_t seed;
    const t2v_step_params* step;
    unsigned long long* prof;   // optional: stamps of step T_out/2 (L workgroup 8B: slots 0..7, T workgroup 0: slots 8..15)
};
#define PT_STAMP(COND, I) do { if (a.prof && (COND) && tid == 0) a.prof[(I)] = __builtin_readcyclecounter(); } while (0)

// 16-byte / 4-byte accesses at agent scope (bypass the CU's vector L1; stores are write-through)
__device__ __forceinline__ void pt_ld16_issue(f32x4& v, const float* p) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
}
__device__ __forceinline__ void pt_wait_loads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void pt_st16(float* p, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void pt_st4(float* p, float v) {
    __hip_atomic_store((unsigned*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned pt_ld4(const float* p) {
    return __hip_atomic_load((const unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool pt_valid(f32x4 v, int nw) {      // the first nw words are not the sentinel
    bool ok = __float_as_uint(v[0]) != PT_SENT;
    ok = ok && (nw < 2 || __float_as_uint(v[1]) != PT_SENT);
    ok = ok && (nw < 3 || __float_as_uint(v[2]) != PT_SENT);
    ok = ok && (nw < 4 || __float_as_uint(v[3]) != PT_SENT);
    return ok;
}

// Gather nk columns [k0, k0 + nk) of all planes of one G row into the LDS state planes.  PER chunks per thread
// (NP * nk <= PER * 512).  A chunk is valid when the words of the items it carries are all written.  Returns through
// *flag (LDS, stays 1 unless a spin timed out); the caller syncs before reading X.
template <int PER, int NP>
__device__ __forceinline__ void pt_gather(f32x4* X, const float* grow, int k0, int nk, int B, unsigned* err, int* flag) {
    const int tid = threadIdx.x;
    const float* src[PER];
    int dsti[PER], nw[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        int c = tid + PT_THREADS * u;
        const bool on = c < NP * nk;
        c = on ? c : 0;
        const int pl = c / nk, kk = k0 + (c - pl * nk);
        src[u] = grow + ((size_t)pl * T2V_XW + kk) * 4;
        dsti[u] = on ? pl * T2V_XW + kk : -1;
        nw[u] = min(4, B - 4 * pl);
    }
    f32x4 v[PER];
    unsigned spins = 0;
    for (;;) {
#pragma unroll
        for (int u = 0; u < PER; ++u) pt_ld16_issue(v[u], src[u]);
        pt_wait_loads();
        bool ok = true;
#pragma unroll
        for (int u = 0; u < PER; ++u) ok = ok && pt_valid(v[u], nw[u]);
        if (ok) break;
        __builtin_amdgcn_s_sleep(2);
        if (++spins > PT_SPIN || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = 0;
            break;
        }
    }
#pragma unroll
    for (int u = 0; u < PER; ++u)
        if (dsti[u] >= 0) X[dsti[u]] = v[u];
}

// one cell's gate pre-activations for the NI items of one plane: acc[u][i] = sum_j w[u][j] * x[kp + 128 j][i], then the
// 16-lane row sums; lane (lane & 15) == (idx & 15) of every row writes value idx = u * NB + B0 + i, so the 8 row partials
// of a gate (2 waves x 4 rows) land in red[gate][partial][idx].  One plane at a time keeps 20 (not 30) accumulators live
// next to the 160 weight registers.
template <int NJ, int NI, int NB, int B0>
__device__ __forceinline__ void pt_gemv_plane(const float (&w)[PT_MAXU][NJ], const f32x4* Xp, int kp, float* red) {
    float acc[PT_MAXU][NI];
#pragma unroll
    for (int u = 0; u < PT_MAXU; ++u)
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[u][i] = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int k = kp + 128 * j;
        float x[4];
        if (NI == 4) {
            const f32x4 x4 = Xp[k];
            x[0] = x4[0]; x[1] = x4[1]; x[2] = x4[2]; x[3] = x4[3];
        } else {
            const float2 x2 = *(const float2*)&Xp[k];
            x[0] = x2.x; x[1] = x2.y; x[2] = 0.f; x[3] = 0.f;
        }
#pragma unroll
        for (int u = 0; u < PT_MAXU; ++u)
#pragma unroll
            for (int i = 0; i < NI; ++i) acc[u][i] = fmaf(w[u][j], x[i], acc[u][i]);
        if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // keep the operand loads close to their use
    }
    const int tid = threadIdx.x, lane = tid & 63, g = tid >> 7;
    const int part = ((tid >> 6) & 1) * 4 + (lane >> 4);
    float* dst = red + (g * 8 + part) * 32;
#pragma unroll
    for (int u = 0; u < PT_MAXU; ++u)
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int idx = u * NB + B0 + i;
            const float sm = row16_sum(acc[u][i]);
            if ((lane & 15) == (idx & 15)) dst[idx] = sm;
        }
}
template <int NJ, int NB>
__device__ __forceinline__ void pt_gemv_all(const float (&w)[PT_MAXU][NJ], const f32x4* X, int kp, float* red) {
    pt_gemv_plane<NJ, 4, NB, 0>(w, X, kp, red);
    if constexpr (NB > 4) pt_gemv_plane<NJ, 2, NB, 4>(w, X + T2V_XW, kp, red);
}

template <int NB>      // 4: B <= 4 (one item plane), 6: B = 5, 6 (two planes)
__global__ __launch_bounds__(PT_THREADS) void k_dec_train_persist(PTArgs a) {
    constexpr int NP = NB > 4 ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const uint64_t seed = t2v_step_seed(a.seed, a.step);
    const int wg = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int B = a.B, Tp = a.T_in, T = a.T_out;
    const int NT = 8 * B, NL = T2V_NWG - NT;
    const size_t grow_f = (size_t)NP * T2V_XW * 4;            // floats per G row
    const int Tcap = (Tp + 15) & ~15;

    if (wg >= NT) {
        // =========================================================================== L role: LSTM rows of both cells
        f32x4* X = (f32x4*)lds;                              // [NP][2560] state planes: X[pl*2560 + k] = items 4pl..4pl+3 of column k
        float* red = lds + (size_t)NP * T2V_XW * 4;          // [4 gates][8 partials][32]
        float* cst = red + 4 * 8 * 32;                       // [2 cells][32] cell states, [4][32] decoder_rnn biases
        int* flag = (int*)(cst + 6 * 32);
        const int j = wg - NT;
        const int u0 = (j * T2V_H) / NL, nu = ((j + 1) * T2V_H) / NL - u0;      // 4 or 5 units
        const int g = tid >> 7, kp = tid & 127;
        float wa[PT_MAXU][PT_JA], wd[PT_MAXU][PT_JD];
#pragma unroll
        for (int u = 0; u < PT_MAXU; ++u) {
            const bool on = u < nu;
            const size_t row = (size_t)g * T2V_H + u0 + (on ? u : 0);
#pragma unroll
            for (int jj = 0; jj < PT_JA; ++jj) {          // [h_att | ctx]: weight_hh, then weight_ih columns 256..767
                const int k = kp + 128 * jj;
                const float w = jj < 8 ? a.w_hh_att[row * T2V_H + k] : a.w_ih_att[row * (T2V_PRE + T2V_E) + T2V_PRE + (k - T2V_H)];
                wa[u][jj] = on ? w : 0.f;
            }
#pragma unroll
            for (int jj = 0; jj < PT_JD; ++jj) {          // [h_att | ctx | h_dec]: weight_ih, then weight_hh
                const int k = kp + 128 * jj;
                const float w = jj < 12 ? a.w_ih_dec[row * T2V_KATT + k] : a.w_hh_dec[row * T2V_H + (k - T2V_KATT)];
                wd[u][jj] = on ? w : 0.f;
            }
        }
        for (int i = tid; i < NP * T2V_XW; i += PT_THREADS) X[i] = f32x4{0.f, 0.f, 0.f, 0.f};      // row 0: zero initial states
        if (tid == 0) flag[0] = 1;
        // cell threads: tid = u * NB + b (wave 0); each keeps its two cell states in registers for the whole pass
        const int cu = tid / NB, cb = tid - cu * NB;
        const bool cell_on = tid < PT_MAXU * NB && cu < nu && cb < B;
        const int U = u0 + (cu < nu ? cu : 0);
        if (tid < 64) { cst[tid] = 0.f; }
        if (cell_on) {
#pragma unroll
            for (int r = 0; r < 4; ++r) cst[64 + 32 * r + tid] = a.bias_dec[r * T2V_H + U];
        }
        __syncthreads();

        for (int t = 0; t <= T; ++t) {
            const bool do_att = t < T, do_dec = t >= 1;
            float* grow = a.G + (size_t)(t + 1) * grow_f;          // row t+1 = [h_att(t) | ctx(t) | h_dec(t-1)]
            PT_STAMP(wg == NT && t == T / 2, 0);
            // ---- attention_rnn(t): X holds row t = [h_att(t-1) | ctx(t-1) | h_dec(t-2)]
            if (do_att) {
                float gp[4] = {0.f, 0.f, 0.f, 0.f};
                if (cell_on) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) gp[r] = a.gpre[((size_t)t * B + cb) * T2V_G + r * T2V_H + U];
                }
                pt_gemv_all<PT_JA, NB>(wa, X, kp, red);
                __syncthreads();
                if (wave == 0) {
                    float hd = 0.f;
                    if (cell_on) {
                        float s[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float* rp = red + (r * 8) * 32 + cu * NB + cb;
                            s[r] = ((rp[0] + rp[32]) + (rp[64] + rp[96])) + ((rp[128] + rp[160]) + (rp[192] + rp[224])) + gp[r];
                        }
                        const float gi = sigmoidf_(s[0]), gf = sigmoidf_(s[1]), gg = tanhf_(s[2]), go = sigmoidf_(s[3]);
                        const uint32_t idx = (uint32_t)cb * T2V_H + U;
                        float cprev = cst[tid];
                        if (t > 0) cprev *= t2v_drop_scale(seed, T2V_RNG_ATT_C, t - 1, idx, a.p_att);
                        const float c = gf * cprev + gi * gg;
                        cst[tid] = c;
                        a.CA[((size_t)(t + 1) * B + cb) * T2V_H + U] = c;
                        if (a.GA) {
                            float* gs = a.GA + ((size_t)t * B + cb) * T2V_G + U;
                            gs[0] = gi; gs[T2V_H] = gf; gs[2 * T2V_H] = gg; gs[3 * T2V_H] = go;
                        }
                        hd = go * tanhf_(c) * t2v_drop_scale(seed, T2V_RNG_ATT_H, t, idx, a.p_att);
                        a.XS[((size_t)(t + 1) * B + cb) * T2V_XW + U] = hd;
                    }
                    // publish: lane (u, plane) sends the 4 items of its plane as one 16-byte write-through store
                    const int pu = lane / NP, pp = lane - pu * NP;
                    f32x4 v4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int bb = 4 * pp + i;
                        const float x = __shfl(hd, min(pu, PT_MAXU - 1) * NB + min(bb, NB - 1), 64);
                        v4[i] = bb < B ? x : 0.f;
                    }
                    if (lane < PT_MAXU * NP && pu < nu) pt_st16(grow + ((size_t)pp * T2V_XW + u0 + pu) * 4, v4);
                }
            }
            PT_STAMP(wg == NT && t == T / 2, 1);
            // ---- decoder_rnn(t-1): same row (its h_dec(t-2) columns included)
            if (do_dec) {
                __syncthreads();               // attention_rnn's cell threads are done with red
                pt_gemv_all<PT_JD, NB>(wd, X, kp, red);
                __syncthreads();
                if (wave == 0) {
                    float hd = 0.f;
                    if (cell_on) {
                        float s[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float* rp = red + (r * 8) * 32 + cu * NB + cb;
                            s[r] = ((rp[0] + rp[32]) + (rp[64] + rp[96])) + ((rp[128] + rp[160]) + (rp[192] + rp[224])) + cst[64 + 32 * r + tid];
                        }
                        const float gi = sigmoidf_(s[0]), gf = sigmoidf_(s[1]), gg = tanhf_(s[2]), go = sigmoidf_(s[3]);
                        const uint32_t idx = (uint32_t)cb * T2V_H + U;
                        const int tt = t - 1;
                        float cprev = cst[32 + tid];
                        if (tt > 0) cprev *= t2v_drop_scale(seed, T2V_RNG_DEC_C, tt - 1, idx, a.p_dec);
                        const float c = gf * cprev + gi * gg;
                        cst[32 + tid] = c;
                        a.CD[((size_t)t * B + cb) * T2V_H + U] = c;
                        if (a.GD) {
                            float* gs = a.GD + ((size_t)tt * B + cb) * T2V_G + U;
                            gs[0] = gi; gs[T2V_H] = gf; gs[2 * T2V_H] = gg; gs[3 * T2V_H] = go;
                        }
                        hd = go * tanhf_(c) * t2v_drop_scale(seed, T2V_RNG_DEC_H, tt, idx, a.p_dec);
                        a.XS[((size_t)(t + 1) * B + cb) * T2V_XW + T2V_KATT + U] = hd;
                    }
                    const int pu = lane / NP, pp = lane - pu * NP;
                    f32x4 v4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int bb = 4 * pp + i;
                        const float x = __shfl(hd, min(pu, PT_MAXU - 1) * NB + min(bb, NB - 1), 64);
                        v4[i] = bb < B ? x : 0.f;
                    }
                    if (t < T && lane < PT_MAXU * NP && pu < nu) pt_st16(grow + ((size_t)pp * T2V_XW + T2V_KATT + u0 + pu) * 4, v4);
                }
            } else if (wave == 0) {
                // t = 0: h_dec(-1) = 0 (XS row 1 was cleared by the reset launch)
                const int pu = lane / NP, pp = lane - pu * NP;
                if (lane < PT_MAXU * NP && pu < nu)
                    pt_st16(grow + ((size_t)pp * T2V_XW + T2V_KATT + u0 + pu) * 4, f32x4{0.f, 0.f, 0.f, 0.f});
            }
            if (t == T) break;
            PT_STAMP(wg == NT && t == T / 2, 2);
            __syncthreads();                   // every wave is done reading X (row t)
            // ---- row t+1 into X: h_att(t) is there (or about to be), h_dec(t-1) follows, ctx(t) comes last
            pt_gather<(NP * T2V_H + PT_THREADS - 1) / PT_THREADS, NP>(X, grow, 0, T2V_H, B, a.err, flag);
            PT_STAMP(wg == NT && t == T / 2, 3);
            pt_gather<(NP * T2V_H + PT_THREADS - 1) / PT_THREADS, NP>(X, grow, T2V_KATT, T2V_H, B, a.err, flag);
            PT_STAMP(wg == NT && t == T / 2, 4);
            pt_gather<(NP * T2V_E + PT_THREADS - 1) / PT_THREADS, NP>(X, grow, T2V_H, T2V_E, B, a.err, flag);
            __syncthreads();
            if (flag[0] != 1) return;
            PT_STAMP(wg == NT && t == T / 2, 5);
        }
        return;
    }

    // =============================================================================== T role: attention slice (b, s)
    const int ab = wg >> 3, as = wg & 7;
    const int TW = Tcap + 32;
    float* wq_s = lds;                                   // [16][1028]
    float* mem_s = wq_s + 16 * 1028;                     // [Tcap][64]
    float* pm_s = mem_s + Tcap * 64;                     // [Tcap][16]
    float* win = pm_s + Tcap * 16;                       // [2][TW]: alignment window, index x <-> position x - 15
    float* eall = win + 2 * TW;                          // [Tcap]
    float* hx = eall + Tcap;                             // [1024] h_att(t) of this item
    float* qv = hx + T2V_H;                              // [16]
    float* qred = qv + 16;                               // [32][16]
    float* cred = qred + 32 * 16;                        // [8][64]
    float* rsm = cred + 8 * 64;                          // [32]
    float* rss = rsm + 32;                               // [32]
    int* flag = (int*)(rss + 32);
    const int g = lane >> 4, c16 = lane & 15;
    for (int i = tid; i < 16 * 1024; i += PT_THREADS) wq_s[(i >> 10) * 1028 + (i & 1023)] = a.wq[(size_t)(16 * as) * 1024 + i];
    for (int i = tid; i < Tp * 64; i += PT_THREADS) mem_s[i] = a.memory[((size_t)ab * Tp + (i >> 6)) * T2V_E + 64 * as + (i & 63)];
    for (int i = tid; i < Tp * 16; i += PT_THREADS) pm_s[i] = a.pm[((size_t)ab * Tp + (i >> 4)) * T2V_A + 16 * as + (i & 15)];
    for (int i = tid; i < 2 * TW; i += PT_THREADS) win[i] = 0.f;
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
    const int mypl = ab >> 2, myw = ab & 3;
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        float* grow = a.G + (size_t)(t + 1) * grow_f;
        PT_STAMP(wg == 0 && t == T / 2, 8);
        // ---- h_att(t) of this item: word myw of 1024 chunks of plane mypl (two per thread)
        {
            const float* s0 = grow + ((size_t)mypl * T2V_XW + tid) * 4;
            f32x4 v0, v1;
            unsigned spins = 0;
            for (;;) {
                pt_ld16_issue(v0, s0);
                pt_ld16_issue(v1, s0 + PT_THREADS * 4);
                pt_wait_loads();
                if (__float_as_uint(v0[myw]) != PT_SENT && __float_as_uint(v1[myw]) != PT_SENT) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > PT_SPIN || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    flag[0] = 0;
                    break;
                }
            }
            hx[tid] = v0[myw];
            hx[tid + PT_THREADS] = v1[myw];
        }
        __syncthreads();
        if (flag[0] != 1) return;
        PT_STAMP(wg == 0 && t == T / 2, 9);
        // ---- query slice: thread = (dim d = tid & 15, k part kp = tid >> 4 of 32 k's)
        {
            const int d = tid & 15, kq = tid >> 4;
            const float* wrow = wq_s + d * 1028 + 32 * kq;
            const float* hp = hx + 32 * kq;
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) acc = fmaf(wrow[i], hp[i], acc);
            qred[kq * 16 + d] = acc;
        }
        __syncthreads();
        if (tid < 16) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) acc += qred[i * 16 + tid];
            qv[tid] = acc;
        }
        __syncthreads();
        const float4 q4 = make_float4(qv[4 * g], qv[4 * g + 1], qv[4 * g + 2], qv[4 * g + 3]);
        // ---- location features (fused filter, K = 64) + partial energies of this slice: wave -> tiles wave, wave + 8
        float* exw = a.EX + (((size_t)t * B + ab) * 8 + as) * Tcap;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int jt = wave + 8 * i;
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
                const f32x4 acc = l0 + l1;
                const int jp = 16 * jt + c16;
                const float4 pm4 = *(const float4*)(pm_s + min(jp, Tp - 1) * 16 + 4 * g);
                float4 sv;
                sv.x = tanhf_(q4.x + acc[0] + pm4.x); sv.y = tanhf_(q4.y + acc[1] + pm4.y);
                sv.z = tanhf_(q4.z + acc[2] + pm4.z); sv.w = tanhf_(q4.w + acc[3] + pm4.w);
                if (a.S && jp < Tp) *(float4*)(a.S + (((size_t)t * B + ab) * Tp + jp) * T2V_A + 16 * as + 4 * g) = sv;
                float esum = vr.x * sv.x + vr.y * sv.y + vr.z * sv.z + vr.w * sv.w;
                esum += __shfl_xor(esum, 16, 64);
                esum += __shfl_xor(esum, 32, 64);
                if (g == 0 && jp < Tp) pt_st4(exw + jp, esum);
            }
        }
        PT_STAMP(wg == 0 && t == T / 2, 10);
        // ---- the 8 partials of every position (fixed order), masked softmax
        float ev0 = -INFINITY;
        if (tid < Tp) {
            const float* e0 = a.EX + ((size_t)t * B + ab) * 8 * Tcap + tid;
            unsigned p[8];
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    p[i] = pt_ld4(e0 + (size_t)i * Tcap);
                    ok = ok && p[i] != PT_SENT;
                }
                if (ok) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > PT_SPIN || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    flag[0] = 0;
                    break;
                }
            }
            const float ev = ((__uint_as_float(p[0]) + __uint_as_float(p[1])) + (__uint_as_float(p[2]) + __uint_as_float(p[3]))) +
                             ((__uint_as_float(p[4]) + __uint_as_float(p[5])) + (__uint_as_float(p[6]) + __uint_as_float(p[7])));
            ev0 = tid < len ? ev : -INFINITY;
        }
        float mloc = ev0;
        mloc = T2V_DPP_MAX(mloc, 0xB1); mloc = T2V_DPP_MAX(mloc, 0x4E);
        mloc = T2V_DPP_MAX(mloc, 0x141); mloc = T2V_DPP_MAX(mloc, 0x140);
        if ((lane & 15) == 0) rsm[tid >> 4] = mloc;
        __syncthreads();
        if (flag[0] != 1) return;
        float m = rsm[0];
#pragma unroll
        for (int u = 1; u < 32; ++u) m = fmaxf(m, rsm[u]);
        const float e0v = tid < Tp ? expf(ev0 - m) : 0.f;
        const float sloc = row16_sum(e0v);
        if ((lane & 15) == 0) rss[tid >> 4] = sloc;
        __syncthreads();
        float ssum = 0.f;
        {
            float sr[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) sr[u] = rss[u];
#pragma unroll
            for (int w = 16; w >= 1; w >>= 1)
#pragma unroll
                for (int u = 0; u < w; ++u) sr[u] += sr[u + w];
            ssum = sr[0];
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
        __syncthreads();
        PT_STAMP(wg == 0 && t == T / 2, 11);
        // ---- context columns [64 as, 64 as + 64): thread = (column c = tid & 63, part = tid >> 6)
        {
            const int c = tid & 63, part = tid >> 6;
            float acc = 0.f;
            for (int jj = part; jj < Tp; jj += 8) acc = fmaf(eall[jj], mem_s[jj * 64 + c], acc);
            cred[part * 64 + c] = acc;
        }
        __syncthreads();
        if (tid < 64) {
            float acc = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += cred[u * 64 + tid];
            pt_st4(grow + ((size_t)mypl * T2V_XW + T2V_H + 64 * as + tid) * 4 + myw, acc);
            a.XS[((size_t)(t + 1) * B + ab) * T2V_XW + T2V_H + 64 * as + tid] = acc;
        }
        PT_STAMP(wg == 0 && t == T / 2, 12);
    }
}

// sentinel fill of the exchange buffers (16 bytes per thread and iteration)
__global__ __launch_bounds__(256) void k_pt_fill(uint4* p, size_t n16) {
    const uint4 s = {PT_SENT, PT_SENT, PT_SENT, PT_SENT};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = s;
}

static size_t pt_lds_bytes(int B, int T_in) {
    const size_t Tcap = (size_t)((T_in + 15) / 16) * 16;
    const size_t np = B > 4 ? 2 : 1;
    const size_t lrole = np * T2V_XW * 4 + 4 * 8 * 32 + 6 * 32 + 4;
    const size_t trole = 16 * 1028 + Tcap * 64 + Tcap * 16 + 2 * (Tcap + 32) + Tcap + T2V_H + 16 + 32 * 16 + 8 * 64 + 64 + 4;
    return sizeof(float) * (lrole > trole ? lrole : trole);
}
#define PT_LDS_MAX (160 * 1024)
static size_t pt_g_floats(int B, int T_out) { return (size_t)(T_out + 2) * (B > 4 ? 2 : 1) * T2V_XW * 4; }
static size_t pt_ex_floats(int B, int T_in, int T_out) { return (size_t)T_out * B * 8 * t2v_tcap(T_in); }

static int pt_device_ok(int B, size_t lds) {
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
        if (hipFuncSetAttribute((const void*)k_dec_train_persist<4>, hipFuncAttributeMaxDynamicSharedMemorySize, PT_LDS_MAX) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_dec_train_persist<6>, hipFuncAttributeMaxDynamicSharedMemorySize, PT_LDS_MAX) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
        attr_set = true;
    }
    int nblk = 0;
    const void* fn = B > 4 ? (const void*)k_dec_train_persist<6> : (const void*)k_dec_train_persist<4>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, fn, PT_THREADS, lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return nblk >= 1;
}

extern "C" int t2v_decoder_train_persist_supported(int B, int T_in) {
    if (!(B >= 1 && B <= PT_MAXB && T_in >= 1 && T_in <= PT_MAXT && pt_lds_bytes(B, T_in) <= PT_LDS_MAX)) return 0;
    return pt_device_ok(B, pt_lds_bytes(B, T_in));
}
extern "C" long t2v_decoder_train_persist_scratch_floats(int B, int T_in, int T_out) {
    if (B < 1 || B > PT_MAXB || T_in < 1 || T_out < 1) return 0;
    return (long)(pt_g_floats(B, T_out) + pt_ex_floats(B, T_in, T_out));
}

extern "C" int t2v_decoder_train_fwd_persistent(const t2v_dec_train_persist_weights* w, const t2v_dec_train_bufs* s, float* scratch,
                                                int B, int T_in, int T_out, float p_att, float p_dec, uint64_t seed, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!w || !s || !scratch || T_out < 1 || !t2v_decoder_train_persist_supported(B, T_in)) return T2V_ERR_ARG;
    if (!w->w_ih_att || !w->w_hh_att || !w->w_ih_dec || !w->w_hh_dec || !w->bias_dec || !w->wq || !w->wcomb || !w->v || !s->gpre ||
        !s->memory || !s->pm || !s->XS || !s->CA || !s->CD || !s->QP || !s->AL || !s->ACUM)
        return T2V_ERR_ARG;
    if ((uintptr_t)scratch & 15) return T2V_ERR_ARG;
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
    const size_t nfl = pt_g_floats(B, T_out) + pt_ex_floats(B, T_in, T_out);
    k_pt_fill<<<1024, 256, 0, stream>>>((uint4*)scratch, nfl / 4);
    PTArgs a;
    a.w_ih_att = w->w_ih_att; a.w_hh_att = w->w_hh_att; a.w_ih_dec = w->w_ih_dec; a.w_hh_dec = w->w_hh_dec;
    a.bias_dec = w->bias_dec; a.wq = w->wq; a.wcomb = w->wcomb; a.v = w->v;
    a.gpre = s->gpre; a.memory = s->memory; a.pm = s->pm; a.lengths = s->lengths;
    a.XS = s->XS; a.CA = s->CA; a.CD = s->CD; a.GA = s->GA; a.GD = s->GD; a.AL = s->AL; a.ACUM = s->ACUM; a.S = s->S;
    a.G = scratch;
    a.EX = scratch + pt_g_floats(B, T_out);
    a.err = sync + 31;
    a.B = B; a.T_in = T_in; a.T_out = T_out; a.p_att = p_att; a.p_dec = p_dec; a.seed = seed;
    a.step = t2v_step_for(stream);
    a.prof = g_t2v_prof;
    const size_t lds = pt_lds_bytes(B, T_in);
    if (B > 4) k_dec_train_persist<6><<<T2V_NWG, PT_THREADS, lds, stream>>>(a);
    else k_dec_train_persist<4><<<T2V_NWG, PT_THREADS, lds, stream>>>(a);
    return t2v_check_launch();
}
