// Location-sensitive attention, forward step (reference Attention.forward model.py:67-88 with
// LocationLayer 12-28), split over the attention dimension: grid = (B, 8), 256 threads; workgroup (b, s) owns
// attention dims [16s, 16s+16) and context columns [64s, 64s+64), so it pulls only ~50 KB per step through its
// CU (a CU gets ~25-40 GB/s of non-local data; the 128 KB of query partials per item was the bottleneck).
//   1. q[16s..] = sum of the 256 per-workgroup query partials of this slice (fixed order)
//   2. location conv for all positions as an MFMA GEMM (K = 2*31 taps padded to 64; redundant per slice)
//   3. location_dense on MFMA + tanh + v-dot over the slice -> PARTIAL energies of all positions
//   4. the 8 workgroups of an item exchange partial energies (write-through stores, one arrival counter per
//      item, bounded spin, sc1 loads); each sums them in a fixed order and does the masked softmax
//   5. context columns [64s, 64s+64) of ctx = alpha·memory
// Saved for the backward (training): tanh outputs S, conv outputs, alpha, cumulative alpha.
#include "t2v_common.h"
#include "t2v_kernels.h"

#define AF_THREADS 256
#define AF_MAXS 8
#define AF_SPIN_LIMIT 4000000

__device__ __forceinline__ void af_st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float af_ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int NJT>      // 16-position tiles covering T_in (6 / 8 / 16)
__global__ __launch_bounds__(AF_THREADS) void k_attn_fwd(AttnFwdArgs a) {
    constexpr int TPAD = 16 * NJT;
    __shared__ float q[16];
    __shared__ float ap[2][TPAD + 32];
    __shared__ float cs[T2V_F][TPAD + 1];
    __shared__ float wcl[T2V_F * 63];
    __shared__ float eall[256];
    __shared__ float scr[64 * 16];          // 64 groups x 16 partial query sums; reused by the context reduction
    __shared__ int ok_flag;
    const int b = blockIdx.x, s = blockIdx.y;          // s = attention-dim slice [16s,16s+16) and context chunk
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    const int Tp = a.T_in;
    const int len = a.lengths ? a.lengths[b] : Tp;

    T2V_STAMP(a, 0);
    // ---- entry: every global read of the kernel is issued here (about 50 KB per workgroup)
    // query partials of this slice: thread = (dq = tid&3 -> 4 consecutive d, wq = tid>>2 -> 4 source workgroups)
    float4 qpart[4];
    {
        const float4* p = (const float4*)(a.qp + ((size_t)b * T2V_NWG + 4 * (tid >> 2)) * T2V_A + 16 * s) + (tid & 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) qpart[i] = p[(size_t)i * (T2V_A / 4)];
    }
    // location_dense rows of this slice as the MFMA A operand: A[d = 16s + c16][k = f = 4st + g]
    float dreg[8];
#pragma unroll
    for (int st = 0; st < 8; ++st) dreg[st] = a.loc_dense[(16 * s + c16) * T2V_F + 4 * st + g];
    // pm / v in the energy-phase output layout: lane (g, c16) <-> d = 16s + 4g + r, position 16jt + c16;
    // wave w handles tiles jt = w, w+4, ..
    float4 pmr[(NJT + 3) / 4];
    const float4 vr = *(const float4*)(a.v + 16 * s + 4 * g);
#pragma unroll
    for (int i = 0; i < (NJT + 3) / 4; ++i) {
        const int j = 16 * (wave + 4 * i) + c16;
        pmr[i] = j < Tp ? *(const float4*)(a.pm + ((size_t)b * Tp + j) * T2V_A + 16 * s + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // context operands: column chunk s (64 columns), thread = (col = tid&63, part = tid>>6), rows part, part+4, ..
    constexpr int MR = 4 * NJT;
    float memr[MR];
    {
        const float* mb = a.memory + (size_t)b * Tp * T2V_E + 64 * s + (tid & 63);
#pragma unroll
        for (int i = 0; i < MR; ++i) {
            const int j = (tid >> 6) + 4 * i;
            memr[i] = j < len ? mb[(size_t)j * T2V_E] : 0.f;
        }
    }
    for (int q4 = tid; q4 < T2V_F * 62 / 4; q4 += AF_THREADS) {
        const float4 w4 = ((const float4*)a.loc_conv)[q4];
        const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) { const int i = 4 * q4 + c; wcl[(i / 62) * 63 + (i % 62)] = wv[c]; }
    }
    for (int i = tid; i < 2 * (TPAD + 32); i += AF_THREADS) {
        const int ch = i / (TPAD + 32), x = i - ch * (TPAD + 32);      // window index x <-> position x - 15
        const int j = x - 15;
        float v = 0.f;
        if (j >= 0 && j < Tp) v = (ch == 0 ? a.al_prev : a.acum_prev)[(size_t)b * Tp + j];
        ap[ch][x] = v;
    }
    // ---- 1. processed query slice: 64 groups of 4 partials, then 64 -> 1 through LDS (fixed order)
    {
        float4 s4 = qpart[0];
#pragma unroll
        for (int i = 1; i < 4; ++i) { s4.x += qpart[i].x; s4.y += qpart[i].y; s4.z += qpart[i].z; s4.w += qpart[i].w; }
        float* dst = scr + (tid >> 2) * 16 + 4 * (tid & 3);
        dst[0] = s4.x; dst[1] = s4.y; dst[2] = s4.z; dst[3] = s4.w;
    }
    __syncthreads();
    {   // 64 -> 16 -> 1 in a fixed order
        const int dd = tid & 15, grp = tid >> 4;
        const float v4 = (scr[(4 * grp) * 16 + dd] + scr[(4 * grp + 1) * 16 + dd]) +
                         (scr[(4 * grp + 2) * 16 + dd] + scr[(4 * grp + 3) * 16 + dd]);
        __syncthreads();
        scr[grp * 16 + dd] = v4;
        __syncthreads();
        if (tid < 16) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += scr[i * 16 + tid];
            q[tid] = acc;
        }
    }

    T2V_STAMP(a, 1);
    // ---- 2. location conv over all positions (redundant in the 8 workgroups of an item, ~2 MFMA tiles per
    //         wave): tile = 16 filters x 16 positions, K = 64 (kk = 32*ch + k)
    {   // wave w owns filter tile f0 = 16*(w&1) and position tiles jt = (w>>1), (w>>1)+2, ..: the weight
        // operands are loaded once and the position tiles run as independent accumulator chains
        const int f0 = 16 * (wave & 1);
        float av[16];
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const int kk = 4 * st + g, ch = kk >> 5, k = kk & 31;
            av[st] = k < T2V_KS ? wcl[(f0 + c16) * 63 + ch * T2V_KS + k] : 0.f;
        }
        static_assert(NJT % 2 == 0, "position tiles are split evenly over wave pairs");
        constexpr int NT = NJT / 2;
        f32x4 acc[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 16; ++st) {
            const int kk = 4 * st + g, ch = kk >> 5, k = kk & 31;
            const float* row = &ap[ch][c16 + (k < T2V_KS ? k : T2V_KS - 1)];
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int jt = (wave >> 1) + 2 * i;
                acc[i] = mfma16x4(av[st], row[16 * jt], acc[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int jt = (wave >> 1) + 2 * i;
#pragma unroll
            for (int r = 0; r < 4; ++r) cs[f0 + 4 * g + r][16 * jt + c16] = acc[i][r];
        }
    }
    __syncthreads();
    if (s == 0 && a.conv_save) {
        for (int i = tid; i < T2V_F * Tp; i += AF_THREADS) {
            const int f = i / Tp, j = i - f * Tp;
            a.conv_save[((size_t)b * T2V_F + f) * Tp + j] = cs[f][j];
        }
    }

    T2V_STAMP(a, 2);
    // ---- 3. partial energies of this d-slice for all positions
    {
        float* exb = a.ex + ((size_t)b * AF_MAXS + s) * 256;
#pragma unroll
        for (int i = 0; i < (NJT + 3) / 4; ++i) {
            const int jt = wave + 4 * i;
            if (jt < NJT) {
                const int j = 16 * jt + c16;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st = 0; st < 8; ++st) acc = mfma16x4(dreg[st], cs[4 * st + g][j], acc);
                const float4 pm4 = pmr[i];
                float4 sv;
                sv.x = tanhf_(q[4 * g + 0] + acc[0] + pm4.x);
                sv.y = tanhf_(q[4 * g + 1] + acc[1] + pm4.y);
                sv.z = tanhf_(q[4 * g + 2] + acc[2] + pm4.z);
                sv.w = tanhf_(q[4 * g + 3] + acc[3] + pm4.w);
                if (a.s_save && j < Tp) *(float4*)(a.s_save + ((size_t)b * Tp + j) * T2V_A + 16 * s + 4 * g) = sv;
                float esum = vr.x * sv.x + vr.y * sv.y + vr.z * sv.z + vr.w * sv.w;
                esum += __shfl_xor(esum, 16, 64);
                esum += __shfl_xor(esum, 32, 64);
                if (g == 0 && j < Tp) af_st_sc1(exb + j, esum);
            }
        }
    }

    T2V_STAMP(a, 3);
    // ---- 4. the 8 workgroups of the item exchange their partial energies; masked softmax over all positions
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        unsigned* cnt = a.sync + b;
        unsigned* err = a.sync + 31;
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (unsigned)AF_MAXS * (unsigned)a.epoch;
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
        const float* ex0 = a.ex + (size_t)b * AF_MAXS * 256 + tid;
        float p[AF_MAXS];
#pragma unroll
        for (int i = 0; i < AF_MAXS; ++i) p[i] = af_ld_sc1(ex0 + i * 256);
        const float ev = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
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
            if (s == 0) {
                a.al_cur[(size_t)b * Tp + tid] = al;
                a.acum_cur[(size_t)b * Tp + tid] = ap[1][15 + tid] + al;
            }
        }
    }
    __syncthreads();

    T2V_STAMP(a, 4);
    // ---- 5. context chunk s (operands already in registers)
    {
        const int part = tid >> 6;
        float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
        for (int i = 0; i < MR; i += 2) {
            const int ja = part + 4 * i, jb = ja + 4;
            acc0 = fmaf(ja < Tp ? eall[ja] : 0.f, memr[i], acc0);
            acc1 = fmaf(jb < Tp ? eall[jb] : 0.f, memr[i + 1], acc1);
        }
        __syncthreads();
        scr[tid] = acc0 + acc1;
        __syncthreads();
        if (tid < 64)
            a.xs_next[(size_t)b * T2V_XW + T2V_H + 64 * s + tid] = (scr[tid] + scr[64 + tid]) + (scr[128 + tid] + scr[192 + tid]);
    }
    T2V_STAMP(a, 5);
}

size_t t2v_attn_fwd_lds(int T_in) { (void)T_in; return 0; }

// f.ex / f.sync / f.epoch must be set by the caller (scratch tail of the QP buffer, see t2vae.h)
void t2v_launch_attn_fwd(const AttnFwdArgs& f, int B, int T_in, hipStream_t stream) {
    const dim3 grid(B, AF_MAXS);
    if (T_in <= 96) k_attn_fwd<6><<<grid, AF_THREADS, 0, stream>>>(f);
    else if (T_in <= 128) k_attn_fwd<8><<<grid, AF_THREADS, 0, stream>>>(f);
    else k_attn_fwd<16><<<grid, AF_THREADS, 0, stream>>>(f);
}
