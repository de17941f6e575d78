// Decoder recurrence, backward (hand-written BPTT; replaces autograd over Decoder.decode,
// reference model.py:346-389 / train.py:225).  Per reverse step t three kernels:
//   k_lstm_bwd : data-gradient GEMVs  YD = Wcat_dec^T·dgd_t,  YA = Wcat_att^T·dga_{t+1}   (MFMA)
//   k_attn_bwd : attention(t) backward (context, softmax, tanh/v, location dense + conv)
//   k_cell_bwd : LSTM cell backward for attention_rnn(t) and decoder_rnn(t-1)
// Weight gradients are NOT accumulated here: the saved per-step pre-activation gradients
// (DGA, DGD, DQ, dpre, DC, DCTX) feed time-batched GEMMs after the loop.
#include <type_traits>
#include "t2v_common.h"
#include "t2v_kernels.h"

#define MFMA4(ACC, WV, XV)                  \
    ACC = mfma16x4((WV).x, (XV).x, ACC);    \
    ACC = mfma16x4((WV).y, (XV).y, ACC);    \
    ACC = mfma16x4((WV).z, (XV).z, ACC);    \
    ACC = mfma16x4((WV).w, (XV).w, ACC)

// Four-wave version of k_lstm_bwd (same finding as for the forward stream: ~64 KB in flight per CU beats 16 waves
// with everything in flight): wave v walks k-blocks [64v, 64v+64) of its tile in 8 rounds of 8 (W, k) float4
// pairs, two rounds in flight, alternating direction per launch.
template <bool WBF>       // WBF: bf16 packs (uint2 per lane and k-block), one 16x16x16 bf16 MFMA per block
__global__ __launch_bounds__(256) void k_lstm_bwd256(LstmBwdArgs a) {
    typedef typename std::conditional<WBF, uint2, float4>::type wt_t;
    const int w = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = lane & 15, g = lane >> 4;
    const bool dec = w < T2V_XW / 16;
    const int wt = dec ? w : w - T2V_XW / 16;
    const float* kv = dec ? a.dgd_t : a.dga_n;
    if (!kv) return;   // block-uniform
    const bool bvalid = b < a.B;
    __shared__ f32x4 red[4][64];
    const wt_t* p = (const wt_t*)(dec ? a.packBD : a.packBA) + (size_t)wt * 256 * 64 + lane;   // tile-major: + kb * 64
    const float* xrow = kv + (size_t)(bvalid ? b : 0) * T2V_G + 4 * g;        // lanes b>=B read row 0 (unused D columns)
    const int kb0 = 64 * wave, flip = a.flip;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    wt_t wv[2][8];
    float4 xv[2][8];
#define B256_LOAD(H)                                                                          \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                           \
        const int kb = flip ? kb0 + 63 - (8 * (H) + i) : kb0 + 8 * (H) + i;                   \
        wv[(H) & 1][i] = p[(size_t)kb * 64];                                                  \
        xv[(H) & 1][i] = *(const float4*)(xrow + 16 * kb);                                    \
    }
    if constexpr (WBF) {
        // bf16 packs: 128 blocks of 32 rows per tile, wave v walks [32v, 32v+32) in 8 rounds of 4, two rounds in flight
        const uint4* p8 = (const uint4*)(dec ? a.packBD : a.packBA) + (size_t)wt * 128 * 64 + lane;
        const float* xrow8 = kv + (size_t)(bvalid ? b : 0) * T2V_G + 8 * g;
        uint4 w8[2][4];
        float4 x8[2][8];
#define B8_LOAD(H)                                                                            \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
        const int jb = flip ? 32 * wave + 31 - (4 * (H) + i) : 32 * wave + 4 * (H) + i;       \
        w8[(H) & 1][i] = p8[(size_t)jb * 64];                                                 \
        x8[(H) & 1][2 * i] = *(const float4*)(xrow8 + 32 * jb);                               \
        x8[(H) & 1][2 * i + 1] = *(const float4*)(xrow8 + 32 * jb + 4);                       \
    }
        B8_LOAD(0)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            if (h + 1 < 8) { B8_LOAD(h + 1) }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = mfma16x32_bf16(w8[h & 1][i], x8[h & 1][2 * i], x8[h & 1][2 * i + 1], acc);
            __builtin_amdgcn_sched_barrier(0);
        }
#undef B8_LOAD
    } else {
    B256_LOAD(0)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        if (h + 1 < 8) { B256_LOAD(h + 1) }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) mfma_block(acc, wv[h & 1][i], xv[h & 1][i]);
        __builtin_amdgcn_sched_barrier(0);
    }
    }
#undef B256_LOAD
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && bvalid) {
        const f32x4 s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        float* y = dec ? a.YD + (size_t)b * T2V_XW : a.YA + (size_t)b * T2V_KATT;
        *(float4*)(y + 16 * wt + 4 * g) = make_float4(s[0], s[1], s[2], s[3]);
    }
}

// attention(t) backward, split over encoder positions: workgroup (b, s) owns positions [s*JS, s*JS + JS), JS = 16
// (T_in <= 128) or 32, S = ceil(T_in / JS) slices per item (any T_in).  The softmax backward needs
// dot = sum_j alpha_j dalpha_j over ALL positions; since dalpha_j = dctx·memory_j + G_j and
// sum_j alpha_j memory_j = ctx_t (saved), every workgroup gets it as dot = dctx·ctx_t + sum_j alpha_j G_j without
// talking to the others.  The location layer is ONE linear map of the alignment window (fused filter bank W_comb,
// attn_fwd.hip), so its backward is one K = 128 MFMA contraction T[(c,k)][j] = sum_d W_comb[d][c,k] dpre[j][d] and
// a diagonal sum; the gradients that flow to the previous step (15-wide halo) are written as per-slice partial rows
// (parity double-buffered) and re-assembled by every workgroup of step t-1; partial dq rows go to the cell
// workgroups of the SAME launch as 8-byte {value, tag} granules (the data is the flag); partial dv rows are summed
// by the host side.  No atomics, fixed summation order.
#define ATB_THREADS 256
#define ATB_SPIN_LIMIT 4000000u

template <int JS>
__device__ __forceinline__ void attn_bwd_body(const AttnBwdArgs& a, const int b, const int s, const int S) {
    constexpr int NJT = JS / 16;           // 16-position MFMA tiles per slice
    constexpr int PW = JS + 30;            // width of a partial dcat row
    extern __shared__ __attribute__((aligned(16))) float dyn[];     // gfull[2][Tcap] | alf[Tcap]
    __shared__ __attribute__((aligned(16))) float dctx[T2V_E];
    __shared__ float de[JS];
    __shared__ float red[1 + JS / 4][16];                       // per (wave, 16-lane row) partial sums: dot, then one row per dalpha
    __shared__ float dpT[T2V_A][JS + 1];
    __shared__ float Tl[64][JS + 1];
    __shared__ __attribute__((aligned(16))) float rq[8][T2V_A], rv[8][T2V_A];     // per row-group partial dq / dv
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    const int Tp = a.T_in, j0 = s * JS;
    const int Tcap = (Tp + 15) & ~15;
    float* gfull0 = dyn;
    float* gfull1 = dyn + Tcap;
    float* alf = dyn + 2 * Tcap;
    const int nown = min(JS, Tp - j0);     // > 0 by construction of S

    T2V_STAMP(a, 0);
    // ---- entry loads (wide, unconditional, in the order they are needed)
    // tanh phase mapping: thread = (dim quad d4 = tid & 31, row group rg = tid >> 5), rows rg, rg + 8, ..
    const int d4 = tid & 31, rg = tid >> 5;
    float4 sreg[JS / 8];
    {
        const float* sp = a.S_t + ((size_t)b * Tp + j0) * T2V_A + 4 * d4;
#pragma unroll
        for (int i = 0; i < JS / 8; ++i) {
            const int jl = rg + 8 * i;
            sreg[i] = *(const float4*)(sp + (size_t)min(jl, nown - 1) * T2V_A);       // rows >= nown: masked below
        }
    }
    float4 m0[JS / 4], m1[JS / 4];                         // this wave's memory rows (wave w: rows w, w+4, ..)
#pragma unroll
    for (int r = 0; r < JS / 4; ++r) {
        const int jl = wave + 4 * r;
        const float* mrow = a.memory + ((size_t)b * Tp + j0 + (jl < nown ? jl : 0)) * T2V_E + lane * 4;
        m0[r] = *(const float4*)mrow;
        m1[r] = *(const float4*)(mrow + 256);
    }
    const float4 vd4 = *(const float4*)(a.v + 4 * d4);
    {
        const float2 h2 = *(const float2*)(a.dHC_t + (size_t)b * (T2V_H + T2V_E) + T2V_H + 2 * tid);
        const float2 y2 = *(const float2*)(a.YD + (size_t)b * T2V_XW + T2V_H + 2 * tid);
        const float2 z2 = *(const float2*)(a.YA + (size_t)b * T2V_KATT + T2V_H + 2 * tid);
        const float2 v2 = make_float2(h2.x + y2.x + z2.x, h2.y + y2.y + z2.y);
        *(float2*)(dctx + 2 * tid) = v2;
        if (s == 0) *(float2*)(a.DCTX_t + (size_t)b * T2V_E + 2 * tid) = v2;
    }
    const float2 ctx2 = *(const float2*)(a.ctx_t + (size_t)b * T2V_XW + 2 * tid);
    // assemble G over all positions from the previous reverse step's per-slice partial rows: position j receives
    // from the slices sp2 with 0 <= j - sp2*JS + 15 < PW (at most three), summed in ascending slice order
    float dot_g = 0.f;
    for (int j = tid; j < Tp; j += ATB_THREADS) {
        float gp = 0.f, gc = a.GC[((size_t)b * S + s) * Tcap + j];
        const int lo = max(0, (j + 15 - PW + JS) / JS), hi = min(S - 1, (j + 15) / JS);
        float pv[3][2];
#pragma unroll
        for (int u = 0; u < 3; ++u) {      // all loads independent
            const int sp2 = lo + u;
            const int jj = j - sp2 * JS + 15;
            const bool in = sp2 <= hi && jj >= 0 && jj < PW;
            const float* row = a.GP_in + (((size_t)b * S + (in ? sp2 : 0)) * 2) * 64 + (in ? jj : 0);
            const float z = in ? 1.f : 0.f;
            pv[u][0] = row[0] * z;
            pv[u][1] = row[64] * z;
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) { gp += pv[u][0]; gc += pv[u][1]; }
        a.GC[((size_t)b * S + s) * Tcap + j] = gc;
        gfull0[j] = gp;
        gfull1[j] = gc;
        const float al = a.al_cur[(size_t)b * Tp + j];
        alf[j] = al;
        dot_g = fmaf(al, gp + gc, dot_g);
    }
    // fused location filter, transposed, as the MFMA A operand of the location backward: wave w owns the (c,k)
    // tile [16w, 16w+16): A[m = c16][kd = 4st + g] = W_comb[4st + g][16w + c16] (read from the transposed copy)
    float areg[32];
    {
        const float4* wp = (const float4*)(a.wcomb + T2V_A * 64 + (16 * wave + c16) * 128 + 32 * g);   // backward copy [kk][g][st]
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 w4 = wp[u];
            areg[4 * u + 0] = w4.x; areg[4 * u + 1] = w4.y; areg[4 * u + 2] = w4.z; areg[4 * u + 3] = w4.w;
        }
    }
    __syncthreads();

    T2V_STAMP(a, 1);
    // ---- dot = dctx·ctx_t + sum_j alpha_j (Gprev_j + Gcum_j); dalpha of the own positions = dctx·memory_j + G_j.
    // Sums over a 16-lane row with DPP, the 16 row partials of each quantity through LDS (no LDS-crossbar shuffles)
    {
        float dotp = dctx[2 * tid] * ctx2.x + dctx[2 * tid + 1] * ctx2.y + dot_g;
        dotp = row16_sum(dotp);
        if (c16 == 0) red[0][4 * wave + g] = dotp;
        const float4 d0 = *(const float4*)(dctx + lane * 4), d1 = *(const float4*)(dctx + 256 + lane * 4);
#pragma unroll
        for (int r = 0; r < JS / 4; ++r) {
            float acc = m0[r].x * d0.x;
            acc = fmaf(m0[r].y, d0.y, acc); acc = fmaf(m0[r].z, d0.z, acc); acc = fmaf(m0[r].w, d0.w, acc);
            acc = fmaf(m1[r].x, d1.x, acc); acc = fmaf(m1[r].y, d1.y, acc);
            acc = fmaf(m1[r].z, d1.z, acc); acc = fmaf(m1[r].w, d1.w, acc);
            acc = row16_sum(acc);
            if (c16 == 0) red[1 + r][4 * wave + g] = acc;
        }
    }
    __syncthreads();
    if (tid < JS) {
        float dsum[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) dsum[u] = red[0][u];
#pragma unroll
        for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
            for (int u = 0; u < w; ++u) dsum[u] += dsum[u + w];
        const int wv = tid & 3, r = tid >> 2;            // row jl = tid was summed by wave wv as its r-th row
        const float dalv = ((red[1 + r][4 * wv] + red[1 + r][4 * wv + 1]) + (red[1 + r][4 * wv + 2] + red[1 + r][4 * wv + 3])) +
                           gfull0[j0 + min(tid, nown - 1)] + gfull1[j0 + min(tid, nown - 1)];
        de[tid] = tid < nown ? alf[j0 + tid] * (dalv - dsum[0]) : 0.f;
    }
    __syncthreads();

    T2V_STAMP(a, 2);
    // ---- through v·tanh(.): dpre, partial dq / dv
    {
        float* sp = a.S_t + ((size_t)b * Tp + j0) * T2V_A + 4 * d4;
        float4 dq = make_float4(0.f, 0.f, 0.f, 0.f), dv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < JS / 8; ++i) {
            const int jl = rg + 8 * i;
            const float dej = de[jl];                    // 0 for rows >= nown
            const float4 sv = sreg[i];
            float4 dp;
            dp.x = dej * vd4.x * (1.0f - sv.x * sv.x); dp.y = dej * vd4.y * (1.0f - sv.y * sv.y);
            dp.z = dej * vd4.z * (1.0f - sv.z * sv.z); dp.w = dej * vd4.w * (1.0f - sv.w * sv.w);
            if (jl < nown) *(float4*)(sp + (size_t)jl * T2V_A) = dp;
            dq.x += dp.x; dq.y += dp.y; dq.z += dp.z; dq.w += dp.w;
            dv.x = fmaf(dej, sv.x, dv.x); dv.y = fmaf(dej, sv.y, dv.y); dv.z = fmaf(dej, sv.z, dv.z); dv.w = fmaf(dej, sv.w, dv.w);
            dpT[4 * d4 + 0][jl] = dp.x; dpT[4 * d4 + 1][jl] = dp.y; dpT[4 * d4 + 2][jl] = dp.z; dpT[4 * d4 + 3][jl] = dp.w;
        }
        *(float4*)&rq[rg][4 * d4] = dq;
        *(float4*)&rv[rg][4 * d4] = dv;
        __syncthreads();
        if (tid < T2V_A) {
            // granule: the cell-backward workgroups of the SAME launch poll it (tag 1; the buffer is zeroed per pass)
            const float dqs = ((rq[0][tid] + rq[1][tid]) + (rq[2][tid] + rq[3][tid])) + ((rq[4][tid] + rq[5][tid]) + (rq[6][tid] + rq[7][tid]));
            __hip_atomic_store(a.DQ_t + ((size_t)b * S + s) * T2V_A + tid,
                               ((t2v_u64)1u << 32) | (t2v_u64)__float_as_uint(dqs), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a.DV[((size_t)b * S + s) * T2V_A + tid] += ((rv[0][tid] + rv[1][tid]) + (rv[2][tid] + rv[3][tid])) +
                                                        ((rv[4][tid] + rv[5][tid]) + (rv[6][tid] + rv[7][tid]));
        }
    }

    T2V_STAMP(a, 3);
    // ---- through the fused location filter on MFMA: T[(c,k)][jl] = sum_d W_comb[d][(c,k)] dpre[jl][d], K = 128
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) {
        f32x4 ac4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) ac4[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 32; ++st) ac4[st & 3] = mfma16x4(areg[st], dpT[4 * st + g][16 * jt + c16], ac4[st & 3]);
        const f32x4 acc = (ac4[0] + ac4[1]) + (ac4[2] + ac4[3]);
#pragma unroll
        for (int r = 0; r < 4; ++r) Tl[16 * wave + 4 * g + r][16 * jt + c16] = acc[r];
    }
    __syncthreads();

    T2V_STAMP(a, 4);
    // ---- gradient wrt the alignment window of this slice: position p = j0 - 15 + jj receives
    //      part[c][jj] = sum_k T[(c,k)][jj - k]   (loc[j] reads a_c[j + k - 15])
    if (tid < 2 * 64) {
        const int c = tid >> 6, jj = tid & 63;
        if (jj < PW) {
            float t[T2V_KS];
#pragma unroll
            for (int k = 0; k < T2V_KS; ++k) {           // unconditional LDS reads (clamped address) + select
                const int jl = jj - k;
                const float tv = Tl[32 * c + k][min(max(jl, 0), JS - 1)];
                t[k] = (jl >= 0 && jl < JS) ? tv : 0.f;
            }
            float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
#pragma unroll
            for (int k = 0; k + 3 < T2V_KS; k += 4) { acc0 += t[k]; acc1 += t[k + 1]; acc2 += t[k + 2]; acc3 += t[k + 3]; }
            acc0 += t[28]; acc1 += t[29]; acc2 += t[30];
            a.GP_out[(((size_t)b * S + s) * 2 + c) * 64 + jj] = (acc0 + acc1) + (acc2 + acc3);
        }
    }
    T2V_STAMP(a, 5);
}

// 64 workgroups x 256 threads; thread = (unit U, item b).  Inside the merged launch the decoder_rnn(t-1) part and
// all operand fetches run while the attention workgroups are still busy; only the W_q^T·dq term waits: the partial
// dq rows arrive as tagged granules and are polled directly (bounded).
template <int NPP>
__device__ __forceinline__ void cell_bwd_body(const CellBwdArgs& a, const int cblk) {
    __shared__ __attribute__((aligned(16))) float wqs[16][T2V_A + 4];
    __shared__ __attribute__((aligned(16))) float dqs[16][T2V_A + 4];
    __shared__ int cell_ok;
    const int tid = threadIdx.x;
    const int U = cblk * 16 + (tid >> 4), b = tid & 15;
    const bool bv = b < a.B;
    const uint32_t idx = (uint32_t)b * T2V_H + U;
    const size_t bu = (size_t)b * T2V_H + U;
    const uint64_t seed = t2v_step_seed(a.seed, a.step);
    if (tid == 0) cell_ok = 1;
#define CELL_STAMP(I) do { if (a.prof && tid == 0 && cblk == 0) a.prof[(I)] = __builtin_readcyclecounter(); } while (0)
    CELL_STAMP(0);
    // everything this thread needs from global memory is requested up front (one latency round)
    float yd0 = 0.f, ya0 = 0.f, ga[4] = {0, 0, 0, 0}, cac = 0.f, cap = 0.f, dca = 0.f;
    float hcp = 0.f, yd1 = 0.f, gd[4] = {0, 0, 0, 0}, cdc = 0.f, cdp = 0.f, dcd = 0.f;
    if (a.do_att) {
        // stage W_q^T rows of this block's 16 units
        for (int i = tid; i < 16 * T2V_A / 4; i += 256) {
            const int u = i >> 5, c4 = i & 31;
            *(float4*)&wqs[u][4 * c4] = *(const float4*)(a.wqT + (size_t)(cblk * 16 + u) * T2V_A + 4 * c4);
        }
        if (bv) {
            yd0 = a.YD[(size_t)b * T2V_XW + U];
            ya0 = a.YA[(size_t)b * T2V_KATT + U];
            const float* gp = a.GA_t + (size_t)b * T2V_G + U;
            ga[0] = gp[0]; ga[1] = gp[T2V_H]; ga[2] = gp[2 * T2V_H]; ga[3] = gp[3 * T2V_H];
            cac = a.CA_cur[bu]; cap = a.CA_prev[bu]; dca = a.DCA[bu];
        }
    }
    if (a.do_dec && bv) {
        hcp = a.dHC_prev[(size_t)b * (T2V_H + T2V_E) + U];
        yd1 = a.YD[(size_t)b * T2V_XW + T2V_KATT + U];
        const float* gp = a.GD_p + (size_t)b * T2V_G + U;
        gd[0] = gp[0]; gd[1] = gp[T2V_H]; gd[2] = gp[2 * T2V_H]; gd[3] = gp[3 * T2V_H];
        cdc = a.CD_cur[bu]; cdp = a.CD_prev[bu]; dcd = a.DCD[bu];
    }
    // ---- decoder_rnn(t-1): independent of the attention backward -> done first
    if (a.do_dec && bv) {
        const int td = a.t - 1;
        const float dh = hcp + yd1;
        const float fh = t2v_drop_scale(seed, T2V_RNG_DEC_H, td, idx, a.p_dec);
        const float fc = t2v_drop_scale(seed, T2V_RNG_DEC_C, td, idx, a.p_dec);
        const float gi = gd[0], gf = gd[1], gg = gd[2], go = gd[3];
        const float tc = tanhf_(cdc);
        const float dht = dh * fh;
        const float dct = dcd * fc + dht * go * (1.0f - tc * tc);
        float cprev = cdp;
        if (td > 0) cprev *= t2v_drop_scale(seed, T2V_RNG_DEC_C, td - 1, idx, a.p_dec);
        float* o = a.DGD_p + (size_t)b * T2V_G + U;
        o[0] = dct * gg * gi * (1.0f - gi);
        o[T2V_H] = dct * cprev * gf * (1.0f - gf);
        o[2 * T2V_H] = dct * gi * (1.0f - gg * gg);
        o[3 * T2V_H] = dht * tc * go * (1.0f - go);
        a.DCD[bu] = dct * gf;
    }
    if (!a.do_att) return;
    // the three state-dropout factors of attention_rnn's cell backward (64-bit counter hashes of (seed, step, unit), ~300 cycles
    // each): evaluated HERE, in front of the nap + poll for the dq rows, not behind W_q^T dq at the very end of the launch (round 6)
    float fh_att = 1.0f, fc_att = 1.0f;
    if (bv) {
        fh_att = t2v_drop_scale(seed, T2V_RNG_ATT_H, a.t, idx, a.p_att);
        fc_att = t2v_drop_scale(seed, T2V_RNG_ATT_C, a.t, idx, a.p_att);
        if (a.t > 0) cap *= t2v_drop_scale(seed, T2V_RNG_ATT_C, a.t - 1, idx, a.p_att);
    }
    __syncthreads();       // cell_ok initialised, wqs staged
    CELL_STAMP(1);
    // ---- gather the attention workgroups' partial dq rows (granules, polled until tagged), fixed-order sum
    // ONE wave polls one sentinel granule per publishing wave (dims 0 and 64 of every slice row) with s_sleep between
    // rounds; the other 255 threads of every cell workgroup stay off the memory system until the rows have landed
    // (hundreds of pollers next to the latency-bound attention workgroups slowed the whole launch by ~2 us)
    // the partial dq rows cannot be there before the attention workgroups have done their entry loads, softmax backward
    // and tanh backward (>= 3.5 us after the launch): nap first, then poll the rows themselves (the data is the flag) with
    // short naps in between — a separate sentinel poll ahead of the gather cost one more memory round trip (+0.35 us),
    // polling without the naps slows the attention workgroups' own loads
    __builtin_amdgcn_s_sleep(40);
    CELL_STAMP(2);
    {
        // thread -> (item, dim) pairs i = tid + 256 p, p < NPP (NPP = pairs per thread rounded up to 1/2/4/8), eight
        // slices per round; every load is unconditional (clamped address) so that all NPP x 8 are in flight at once —
        // loads inside divergent branches get a vmcnt(0) each and ran one round trip at a time
        const int npair = a.B * T2V_A;
        float tot[NPP];
#pragma unroll
        for (int p = 0; p < NPP; ++p) tot[p] = 0.f;
        for (int s0 = 0; s0 < a.S; s0 += 8) {
            float pv[NPP][8];
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int p = 0; p < NPP; ++p) {
                    const int i = min(tid + 256 * p, npair - 1);
                    const t2v_u64* gq = a.DQ_t + (size_t)(i >> 7) * a.S * T2V_A + (i & (T2V_A - 1));
#pragma unroll
                    for (int sl = 0; sl < 8; ++sl) {
                        const bool live = s0 + sl < a.S;
                        const t2v_u64 x = __hip_atomic_load(gq + (size_t)(live ? s0 + sl : a.S - 1) * T2V_A, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        pv[p][sl] = live ? __uint_as_float((unsigned)x) : 0.f;
                        ok = ok && (unsigned)(x >> 32) == 1u;
                    }
                }
                if (ok) break;
                __builtin_amdgcn_s_sleep(6);
                if (++spins > ATB_SPIN_LIMIT || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    cell_ok = 0;
                    break;
                }
            }
#pragma unroll
            for (int p = 0; p < NPP; ++p)
                tot[p] += ((pv[p][0] + pv[p][1]) + (pv[p][2] + pv[p][3])) + ((pv[p][4] + pv[p][5]) + (pv[p][6] + pv[p][7]));
        }
#pragma unroll
        for (int p = 0; p < NPP; ++p) {
            const int i = tid + 256 * p;
            if (i < npair) dqs[i >> 7][i & (T2V_A - 1)] = tot[p];
        }
    }
    __syncthreads();
    CELL_STAMP(3);
    if (!cell_ok) return;
    // W_q^T·dq for this block's 16 units x 16 items on MFMA: D[m = unit][n = item] = sum_d wqs[m][d] dqs[n][d]; the four
    // waves split K = 128 (8 k-steps each), partial tiles through LDS (fixed order).  Rows of dqs for items >= B hold
    // stale LDS words: they only reach D columns nobody reads.
    float wq_dq;
    {
        __shared__ f32x4 dred[4][64];
        const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const int k = 4 * (8 * wave + st) + g;
            acc = mfma16x4(wqs[c16][k], dqs[c16][k], acc);
        }
        dred[wave][lane] = acc;
        __syncthreads();
        const int m = tid >> 4, n = tid & 15, src = n + 16 * (m >> 2), r = m & 3;
        wq_dq = (dred[0][src][r] + dred[1][src][r]) + (dred[2][src][r] + dred[3][src][r]);
    }
    if (!bv) return;
    {
        const float dh = yd0 + ya0 + wq_dq;
        const float fh = fh_att, fc = fc_att;
        const float gi = ga[0], gf = ga[1], gg = ga[2], go = ga[3];
        const float tc = tanhf_(cac);
        const float dht = dh * fh;
        const float dct = dca * fc + dht * go * (1.0f - tc * tc);
        const float cprev = cap;              // (already scaled by its dropout factor, above)
        float* o = a.DGA_t + (size_t)b * T2V_G + U;
        o[0] = dct * gg * gi * (1.0f - gi);
        o[T2V_H] = dct * cprev * gf * (1.0f - gf);
        o[2 * T2V_H] = dct * gi * (1.0f - gg * gg);
        o[3 * T2V_H] = dht * tc * go * (1.0f - go);
        a.DCA[bu] = dct * gf;
    }
    CELL_STAMP(4);
}

// One launch per reverse step: workgroups [0, B*S) = attention backward slices, then 64 cell-backward workgroups.
template <int JS, int NPP>
__global__ __launch_bounds__(256) void k_attn_cell_bwd(AttnBwdArgs a, CellBwdArgs c, int nattn, int S) {
    const int blk = blockIdx.x;
    if (blk < nattn) attn_bwd_body<JS>(a, blk / S, blk % S, S);
    else cell_bwd_body<NPP>(c, blk - nattn);
}
template <int JS>
static void launch_attn_cell_bwd(const AttnBwdArgs& fa, const CellBwdArgs& c, int nattn, int S, int B, size_t lds, hipStream_t stream) {
    const dim3 grid(nattn + T2V_H / 16);
    if (B <= 2) k_attn_cell_bwd<JS, 1><<<grid, 256, lds, stream>>>(fa, c, nattn, S);
    else if (B <= 4) k_attn_cell_bwd<JS, 2><<<grid, 256, lds, stream>>>(fa, c, nattn, S);
    else if (B <= 8) k_attn_cell_bwd<JS, 4><<<grid, 256, lds, stream>>>(fa, c, nattn, S);
    else k_attn_cell_bwd<JS, 8><<<grid, 256, lds, stream>>>(fa, c, nattn, S);
}

extern "C" int t2v_attn_bwd_slices(int T_in) { return T_in < 1 ? 0 : t2v_attn_bwd_slices_(T_in); }

// mask bits: 1 = k_lstm_bwd256, 2 = k_attn_cell_bwd (3 = the reverse pass; single bits = measurement replays)
static int launch_train_bwd(const t2v_dec_weights* w, const t2v_dec_train_bufs* s,
                            const t2v_dec_bwd_bufs* g, int B, int T_in, int T_out,
                            float p_att, float p_dec, uint64_t seed, void* stream_, int mask) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!w || !s || !g || B < 1 || B > 16 || T_in < 1 || T_in > T2V_MAX_T_IN || T_out < 1) return T2V_ERR_ARG;
    if (!w->packB_att || !w->packB_dec || !w->wcomb) return T2V_ERR_ARG;
    const int JS = t2v_attn_bwd_js(T_in), S = t2v_attn_bwd_slices_(T_in);
    const size_t Tcap = t2v_tcap(T_in);
    T2VZeroRegions z;
    z.add(g->YD, sizeof(float) * B * T2V_XW);
    z.add(g->YA, sizeof(float) * B * T2V_KATT);
    z.add(g->DCA, sizeof(float) * B * T2V_H);
    z.add(g->DCD, sizeof(float) * B * T2V_H);
    z.add(g->GPREV, sizeof(float) * 2 * B * S * 2 * 64);                    // partial dcat rows x parity
    z.add(g->GCUM, sizeof(float) * ((size_t)B * S * Tcap + 64));            // per-workgroup Gcum copies + sync words
    z.add(g->DV, sizeof(float) * B * S * T2V_A);
    z.add(g->DQ, sizeof(t2v_u64) * (size_t)T_out * B * S * T2V_A);          // granule tags
    t2v_zero_regions(z, stream);

    const size_t HC = T2V_H + T2V_E;
    unsigned* sync = (unsigned*)(g->GCUM + (size_t)B * S * Tcap);     // [1] error word
    const size_t lds = sizeof(float) * 3 * Tcap;
    const t2v_step_params* step_rec = t2v_step_for(stream);
    for (int t = T_out; t >= 0; --t) {
        bool have_attn = false;
        AttnBwdArgs fa = {};
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
            if (mask & 1) {
                if (w->packs_bf16) k_lstm_bwd256<true><<<T2V_NWG, 256, 0, stream>>>(l);
                else k_lstm_bwd256<false><<<T2V_NWG, 256, 0, stream>>>(l);
            }

            AttnBwdArgs f;
            f.dHC_t = g->dHC + (size_t)t * B * HC;
            f.YD = g->YD;
            f.YA = g->YA;
            f.al_cur = s->AL + (size_t)(t + 1) * B * T_in;
            f.memory = s->memory;
            f.wcomb = w->wcomb;
            f.v = w->v;
            f.S_t = s->S + (size_t)t * B * T_in * T2V_A;
            f.DQ_t = (t2v_u64*)g->DQ + (size_t)t * B * S * T2V_A;
            f.ctx_t = s->XS + (size_t)(t + 1) * B * T2V_XW + T2V_H;
            f.GP_in = g->GPREV + (size_t)((t + 1) & 1) * B * S * 2 * 64;
            f.GP_out = g->GPREV + (size_t)(t & 1) * B * S * 2 * 64;
            f.GC = g->GCUM;
            f.DCTX_t = g->DCTX + (size_t)t * B * T2V_E;
            f.DV = g->DV;
            f.T_in = T_in;
            f.prof = g_t2v_prof ? g_t2v_prof + 16 : nullptr;
            have_attn = true;
            fa = f;
        }
        CellBwdArgs c;
        c.YD = g->YD;
        c.YA = g->YA;
        c.DQ_t = t < T_out ? (const t2v_u64*)g->DQ + (size_t)t * B * S * T2V_A : nullptr;
        c.S = S;
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
        c.step = step_rec;
        c.err = sync + 1;
        c.prof = g_t2v_prof ? g_t2v_prof + 24 : nullptr;
        const int nattn = have_attn ? B * S : 0;
        if (!(mask & 2)) continue;
        if (JS == 16) launch_attn_cell_bwd<16>(fa, c, nattn, S, B, lds, stream);
        else launch_attn_cell_bwd<32>(fa, c, nattn, S, B, lds, stream);
    }
    return t2v_check_launch();
}

extern "C" int t2v_decoder_train_bwd(const t2v_dec_weights* w, const t2v_dec_train_bufs* s,
                                     const t2v_dec_bwd_bufs* g, int B, int T_in, int T_out,
                                     float p_att, float p_dec, uint64_t seed, void* stream_) {
    return launch_train_bwd(w, s, g, B, T_in, T_out, p_att, p_dec, seed, stream_, 3);
}

extern "C" int t2v_decoder_replay_bwd_kernels(const t2v_dec_weights* w, const t2v_dec_train_bufs* s,
                                              const t2v_dec_bwd_bufs* g, int B, int T_in, int T_out,
                                              float p_att, float p_dec, uint64_t seed, int kernel_mask, void* stream_) {
    return launch_train_bwd(w, s, g, B, T_in, T_out, p_att, p_dec, seed, stream_, kernel_mask & 3);
}
