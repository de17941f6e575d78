// Location-sensitive attention, forward step (reference Attention.forward model.py:67-88 with
// LocationLayer 12-28), split over encoder positions: grid = (B, S), 256 threads; workgroup (b, s) owns
// positions [s*JS, s*JS+JS), JS = 16 or 32.
//   1. q = sum of the 256 per-workgroup query partials written by k_lstm_fwd (every workgroup, fixed order)
//   2. location conv for the own positions as an MFMA GEMM (K = 2*31 taps padded to 64)
//   3. location_dense on MFMA + tanh + v-dot  -> energies of the own positions
//   4. the S workgroups of an item exchange their energies (write-through stores, one arrival counter per
//      item, bounded spin, sc1 loads) and each computes the masked softmax over all positions
//   5. context: workgroup s produces 64-column chunks s, s+S, ... of ctx = alpha·memory
// Saved for the backward (training): tanh outputs S, conv outputs, alpha, cumulative alpha.
#include "t2v_common.h"
#include "t2v_kernels.h"

#define AF_THREADS 256
#define AF_MAXS 8
#define AF_SPIN_LIMIT 4000000

__device__ __forceinline__ void af_st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float af_ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int JS>
__global__ __launch_bounds__(AF_THREADS) void k_attn_fwd(AttnFwdArgs a) {
    constexpr int NJT = JS / 16;
    constexpr int APW = JS + 30;                // prev/cum weights window incl. the 15-wide halos
    __shared__ float q[T2V_A];
    __shared__ float ap[2][APW + 2];
    __shared__ float cs[T2V_F][JS + 1];
    __shared__ float wcl[T2V_F * 63];
    __shared__ float ep[4][JS];
    __shared__ float eall[256];
    __shared__ float scr[8 * T2V_A];      // 8 groups x 128 query partial sums; reused by the context reduction
    __shared__ int ok_flag;
    const int b = blockIdx.x, s = blockIdx.y, S = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    const int Tp = a.T_in, j0 = s * JS;
    const int nown = min(JS, Tp - j0);
    const int len = a.lengths ? a.lengths[b] : Tp;

    T2V_STAMP(a, 0);
    // ---- entry: every global read is issued here
    float4 qpart[32];                       // thread = (d4 = tid&31 -> d = 4*d4.., wg = tid>>5 -> partials 32wg..32wg+31)
    {
        const float4* p = (const float4*)(a.qp + ((size_t)b * T2V_NWG + 32 * (tid >> 5)) * T2V_A) + (tid & 31);
#pragma unroll
        for (int i = 0; i < 32; ++i) qpart[i] = p[(size_t)i * (T2V_A / 4)];
    }
    // location_dense as the MFMA A operand of phase 3: rows d = 16*dt + c16, k = f = 4st + g; wave owns dt = 2w, 2w+1
    float dreg[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int st = 0; st < 8; ++st) dreg[h][st] = a.loc_dense[(16 * (2 * wave + h) + c16) * T2V_F + 4 * st + g];
    // pm / v in the phase-3 output layout: lane (g, c16) holds d = 16*dt + 4g + r for position j0 + 16jt + c16
    float4 pmr[2][NJT], vr[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int d0 = 16 * (2 * wave + h) + 4 * g;
        vr[h] = *(const float4*)(a.v + d0);
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) {
            const int jl = 16 * jt + c16;
            pmr[h][jt] = jl < nown ? *(const float4*)(a.pm + ((size_t)b * Tp + j0 + jl) * T2V_A + d0)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    for (int q4 = tid; q4 < T2V_F * 62 / 4; q4 += AF_THREADS) {
        const float4 w4 = ((const float4*)a.loc_conv)[q4];
        const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) { const int i = 4 * q4 + c; wcl[(i / 62) * 63 + (i % 62)] = wv[c]; }
    }
    if (tid < 2 * APW) {
        const int ch = tid / APW, x = tid - ch * APW;          // window index -> position j0 - 15 + x
        const int j = j0 - 15 + x;
        float v = 0.f;
        if (j >= 0 && j < Tp) v = (ch == 0 ? a.al_prev : a.acum_prev)[(size_t)b * Tp + j];
        ap[ch][x] = v;
    }
    // context operands: 64-column chunk(s) of memory, thread = (col = tid&63, part = tid>>6); part walks j = part, part+4, ..
    // ---- 1. processed query: 8 groups of 32 partials, then 8 -> 1 through LDS (fixed order)
    {
        float4 s0 = qpart[0], s1 = qpart[1];
#pragma unroll
        for (int i = 2; i < 32; i += 2) {
            s0.x += qpart[i].x; s0.y += qpart[i].y; s0.z += qpart[i].z; s0.w += qpart[i].w;
            s1.x += qpart[i + 1].x; s1.y += qpart[i + 1].y; s1.z += qpart[i + 1].z; s1.w += qpart[i + 1].w;
        }
        float* dst = scr + (tid >> 5) * T2V_A + 4 * (tid & 31);
        dst[0] = s0.x + s1.x; dst[1] = s0.y + s1.y; dst[2] = s0.z + s1.z; dst[3] = s0.w + s1.w;
    }
    __syncthreads();
    if (tid < T2V_A) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += scr[i * T2V_A + tid];
        q[tid] = acc;
    }
    // memory rows of this workgroup's context chunk(s), prefetched before the barrier
    constexpr int MR = JS == 16 ? 32 : 64;                      // rows per thread: T_in <= 128 / 256
    float memr[2][MR];
    const int nchunk = min(2, (T2V_E / 64 - s + S - 1) / S);   // chunks s, s+S are prefetched
    {
        const int part = tid >> 6;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float* mb = a.memory + (size_t)b * Tp * T2V_E + 64 * (s + c * S) + (tid & 63);
#pragma unroll
            for (int i = 0; i < MR; ++i) {
                const int j = part + 4 * i;
                memr[c][i] = (c < nchunk && j < len) ? mb[(size_t)j * T2V_E] : 0.f;
            }
        }
    }

    T2V_STAMP(a, 1);
    // ---- 2. location conv of the own positions: tile = 16 filters x 16 positions, K = 64 (kk = 32*ch + k)
    if (wave < 2 * NJT) {
        const int f0 = 16 * (wave & 1), jt = wave >> 1;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const int kk = 4 * st + g, ch = kk >> 5, k = kk & 31;
            const float av = k < T2V_KS ? wcl[(f0 + c16) * 63 + ch * T2V_KS + k] : 0.f;
            const float bv = ap[ch][16 * jt + c16 + (k < T2V_KS ? k : T2V_KS - 1)];
            acc = mfma16x4(av, bv, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) cs[f0 + 4 * g + r][16 * jt + c16] = acc[r];
    }
    __syncthreads();
    if (a.conv_save) {
        for (int i = tid; i < T2V_F * JS; i += AF_THREADS) {
            const int f = i / JS, jl = i - f * JS;
            if (jl < nown) a.conv_save[((size_t)b * T2V_F + f) * Tp + j0 + jl] = cs[f][jl];
        }
    }

    T2V_STAMP(a, 2);
    // ---- 3. energies of the own positions
    {
#pragma unroll
        for (int jt = 0; jt < NJT; ++jt) {
            const int jl = 16 * jt + c16;
            float esum = 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int d0 = 16 * (2 * wave + h) + 4 * g;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < 8; ++st) acc = mfma16x4(dreg[h][st], cs[4 * st + g][jl], acc);
                const float4 pm4 = pmr[h][jt];
                float4 sv;
                sv.x = tanhf_(q[d0 + 0] + acc[0] + pm4.x);
                sv.y = tanhf_(q[d0 + 1] + acc[1] + pm4.y);
                sv.z = tanhf_(q[d0 + 2] + acc[2] + pm4.z);
                sv.w = tanhf_(q[d0 + 3] + acc[3] + pm4.w);
                if (a.s_save && jl < nown) *(float4*)(a.s_save + ((size_t)b * Tp + j0 + jl) * T2V_A + d0) = sv;
                esum += vr[h].x * sv.x + vr[h].y * sv.y + vr[h].z * sv.z + vr[h].w * sv.w;
            }
            esum += __shfl_xor(esum, 16, 64);
            esum += __shfl_xor(esum, 32, 64);
            if (g == 0) ep[wave][jl] = esum;
        }
    }
    __syncthreads();

    T2V_STAMP(a, 3);
    // ---- 4. exchange energies among the item's workgroups, masked softmax over all positions
    float* exb = a.ex + (size_t)b * AF_MAXS * 32;
    if (tid < nown) af_st_sc1(exb + s * 32 + tid, (ep[0][tid] + ep[1][tid]) + (ep[2][tid] + ep[3][tid]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        unsigned* cnt = a.sync + b;
        unsigned* err = a.sync + 31;
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (unsigned)S * (unsigned)a.epoch;
        int good = 1;
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > AF_SPIN_LIMIT || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = 0;
                break;
            }
        }
        ok_flag = good;
    }
    __syncthreads();
    if (!ok_flag) return;
    if (tid < Tp) {
        const int sl = tid / JS;
        const float ev = af_ld_sc1(exb + sl * 32 + (tid - sl * JS));
        eall[tid] = tid < len ? ev : -INFINITY;
    }
    __syncthreads();
    {
        float m = -INFINITY;
        for (int j = lane; j < Tp; j += 64) m = fmaxf(m, eall[j]);
        m = wave_max(m);
        float sum = 0.f;
        for (int j = lane; j < Tp; j += 64) sum += expf(eall[j] - m);
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        __syncthreads();
        if (tid < Tp) {
            const float al = expf(eall[tid] - m) * inv;
            eall[tid] = al;
            const int jl = tid - j0;
            if (jl >= 0 && jl < nown) {
                a.al_cur[(size_t)b * Tp + tid] = al;
                a.acum_cur[(size_t)b * Tp + tid] = ap[1][15 + jl] + al;
            }
        }
    }
    __syncthreads();

    T2V_STAMP(a, 4);
    // ---- 5. context chunks (operands already in registers)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        if (c < nchunk) {
            const int part = tid >> 6;
            float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
            for (int i = 0; i < MR; i += 2) {
                const int ja = part + 4 * i, jb = ja + 4;
                acc0 = fmaf(ja < Tp ? eall[ja] : 0.f, memr[c][i], acc0);
                acc1 = fmaf(jb < Tp ? eall[jb] : 0.f, memr[c][i + 1], acc1);
            }
            __syncthreads();
            scr[tid] = acc0 + acc1;
            __syncthreads();
            if (tid < 64)
                a.xs_next[(size_t)b * T2V_XW + T2V_H + 64 * (s + c * S) + tid] =
                    (scr[tid] + scr[64 + tid]) + (scr[128 + tid] + scr[192 + tid]);
        }
    }
    for (int chunk = s + 2 * S; chunk < T2V_E / 64; chunk += S) {     // only when S < 4 (very short texts)
        const int part = tid >> 6;
        const float* mb = a.memory + (size_t)b * Tp * T2V_E + 64 * chunk + (tid & 63);
        float acc = 0.f;
        for (int j = part; j < len; j += 4) acc = fmaf(eall[j], mb[(size_t)j * T2V_E], acc);
        __syncthreads();
        scr[tid] = acc;
        __syncthreads();
        if (tid < 64)
            a.xs_next[(size_t)b * T2V_XW + T2V_H + 64 * chunk + tid] = (scr[tid] + scr[64 + tid]) + (scr[128 + tid] + scr[192 + tid]);
    }
    T2V_STAMP(a, 5);
}

size_t t2v_attn_fwd_lds(int T_in) { (void)T_in; return 0; }
static inline int attn_fwd_js(int T_in) { return 16 * ((T_in + 127) / 128); }

// f.ex / f.sync / f.epoch must be set by the caller (scratch tail of the QP buffer, see t2vae.h)
void t2v_launch_attn_fwd(const AttnFwdArgs& f, int B, int T_in, hipStream_t stream) {
    const int JS = attn_fwd_js(T_in), S = (T_in + JS - 1) / JS;
    if (JS == 16) k_attn_fwd<16><<<dim3(B, S), AF_THREADS, 0, stream>>>(f);
    else k_attn_fwd<32><<<dim3(B, S), AF_THREADS, 0, stream>>>(f);
}
