// Location-sensitive attention, forward step (reference Attention.forward model.py:67-88 with
// LocationLayer 12-28), split over the attention dimension: grid = (B, 8), 256 threads; workgroup (b, s) owns
// attention dims [16s, 16s+16) and context columns [64s, 64s+64), so it pulls only ~45 KB per step through its
// CU (a CU gets ~25-40 GB/s of non-local data; the 128 KB of query partials per item was the bottleneck).
//   1. q[16s..] = sum of the 256 per-workgroup query partials of this slice (fixed order)
//   2. location features + energies in ONE MFMA GEMM per 16-position tile: LocationLayer is conv (no bias, no
//      activation) followed by a linear layer, i.e. a single linear map of the 2x31 alignment window; the fused
//      filter bank W_comb[d][c,k] = sum_f dense[d][f] conv[f][c][k] (k_loc_fuse, once per pass) makes
//      loc[j][d] = sum_{c,k} W_comb[d][c,k] a_c[j+k-15] a K = 64 contraction, then tanh / v-dot over the slice
//      -> PARTIAL energies of all positions
//   3. the 8 workgroups of an item exchange partial energies as 8-byte {value, epoch tag} granules (one sc1 store
//      each, the data is the flag: MI355X guide G16 form R2 — no counter, no fence, no drain); each sums them in a
//      fixed order and does the masked softmax
//   4. context columns [64s, 64s+64) of ctx = alpha·memory
// Any T_in: NJT = 6 / 8 / 16 position tiles cover T_in <= 256 with every operand register-resident; longer
// inputs (koemo reaches 555 symbols) take the BIG variant, which walks 256-position chunks.
// Saved for the backward (training): tanh outputs S, alpha, cumulative alpha.
#include "t2v_common.h"
#include "t2v_kernels.h"

#define AF_THREADS 256
#define AF_NS 8
#define AF_SPIN_LIMIT 4000000

__device__ __forceinline__ void af_put(t2v_u64* p, float v, unsigned tag) {
    __hip_atomic_store(p, ((t2v_u64)tag << 32) | (t2v_u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ t2v_u64 af_get(const t2v_u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// W_comb (128, 64): column kk = 32c + k for channel c (0 = previous weights, 1 = cumulative weights), tap k < 31;
// columns 31 and 63 are zero padding so that K = 64 = 16 MFMA k-steps.  Stored twice, each in the order the
// consuming lanes read it as float4 runs (a lane's MFMA operands are kk = 4st + g resp. d = 4st + g):
//   wc[0    .. 8192): forward  copy  F[d][g][st]  = W_comb[d][4st + g]        (g < 4, st < 16)
//   wc[8192 .. 16384): backward copy R[kk][g][st] = W_comb[4st + g][kk]       (g < 4, st < 32)
__global__ void k_loc_fuse(const float* __restrict__ conv, const float* __restrict__ dense, float* __restrict__ wc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // 128 * 64
    if (i >= T2V_A * 64) return;
    const int d = i >> 6, n = i & 63, c = n >> 5, k = n & 31;
    float acc = 0.f;
    if (k < T2V_KS)
        for (int f = 0; f < T2V_F; ++f) acc = fmaf(dense[d * T2V_F + f], conv[(f * 2 + c) * T2V_KS + k], acc);
    wc[d * 64 + (n & 3) * 16 + (n >> 2)] = acc;
    wc[T2V_A * 64 + n * 128 + (d & 3) * 32 + (d >> 2)] = acc;
}

extern "C" int t2v_fuse_location_weights(const float* loc_conv, const float* loc_dense, float* wcomb, void* stream_) {
    if (!loc_conv || !loc_dense || !wcomb) return T2V_ERR_ARG;
    k_loc_fuse<<<(T2V_A * 64 + 255) / 256, 256, 0, (hipStream_t)stream_>>>(loc_conv, loc_dense, wcomb);
    return t2v_check_launch();
}

template <int NJT, bool BIG>      // NJT 16-position tiles per chunk; !BIG: one chunk covers T_in
__global__ __launch_bounds__(AF_THREADS) void k_attn_fwd(AttnFwdArgs a) {
    constexpr int TPAD = 16 * NJT;
    constexpr int NI = (NJT + 3) / 4;           // tiles per wave
    extern __shared__ __attribute__((aligned(16))) float eall[];     // Tcap energies -> attention weights
    __shared__ float ap[2][TPAD + 32];
    __shared__ __attribute__((aligned(16))) float scr[AF_THREADS];   // per-wave query sums; reused by the context reduction
    __shared__ int ok_flag;
    const int b = blockIdx.x, s = blockIdx.y;          // s = attention-dim slice [16s,16s+16) and context chunk
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    const int Tp = a.T_in;
    const int Tcap = (Tp + 15) & ~15;
    const int len = a.lengths ? a.lengths[b] : Tp;

    T2V_STAMP(a, 0);
    // ---- entry: every global read that does not depend on the exchange is issued here, in the order the results are
    // needed (VMEM returns in order): alignment window and query partials first, the context operands last
    if (tid == 0) ok_flag = 1;
    float apv[(2 * (TPAD + 32) + AF_THREADS - 1) / AF_THREADS];
#pragma unroll
    for (int u = 0; u < (2 * (TPAD + 32) + AF_THREADS - 1) / AF_THREADS; ++u) {      // window index x <-> position x - 15
        const int i = tid + AF_THREADS * u;
        const int ch = i >= TPAD + 32 ? 1 : 0, x = i - ch * (TPAD + 32);
        const int j = x - 15;
        // unconditional load from a clamped address + select: a load inside a divergent branch costs a vmcnt(0)
        const float av = (ch == 0 ? a.al_prev : a.acum_prev)[(size_t)b * Tp + min(max(j, 0), Tp - 1)];
        apv[u] = av * ((i < 2 * (TPAD + 32) && j >= 0 && j < Tp) ? 1.f : 0.f);    // (a select would let the compiler sink the load back under the branch)
    }
    // fused location filter rows of this slice as the MFMA A operand: A[d = 16s + c16][kk = 4st + g]
    float areg[16];
    {
        const float4* wp = (const float4*)(a.wcomb + (16 * s + c16) * 64 + 16 * g);      // forward copy: [d][g][st]
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 w4 = wp[u];
            areg[4 * u + 0] = w4.x; areg[4 * u + 1] = w4.y; areg[4 * u + 2] = w4.z; areg[4 * u + 3] = w4.w;
        }
    }
    const float4 vr = *(const float4*)(a.v + 16 * s + 4 * g);
    // query partials of this slice: thread = (dq = tid&3 -> 4 consecutive d, wq = tid>>2 -> 4 source workgroups)
    float4 qpart[4];
    {
        const float4* p = (const float4*)(a.qp + ((size_t)b * T2V_NWG + 4 * (tid >> 2)) * T2V_A + 16 * s) + (tid & 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) qpart[i] = p[(size_t)i * (T2V_A / 4)];
    }
    // pm in the energy-phase output layout: lane (g, c16) <-> d = 16s + 4g + r, position 16jt + c16;
    // wave w handles tiles jt = w, w+4, ..
    float4 pmr[NI];
    float4 memr[BIG ? 1 : NJT];
    if constexpr (!BIG) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int j = 16 * (wave + 4 * i) + c16;
            pmr[i] = *(const float4*)(a.pm + ((size_t)b * Tp + min(j, Tp - 1)) * T2V_A + 16 * s + 4 * g);   // j >= T_in: unused
        }
        // context operands: column chunk s (64 columns), thread = (column quad cq = tid&15, row group rg = tid>>4),
        // rows rg, rg+16, ..: one float4 per row instead of four dword loads
        const float4* mb = (const float4*)(a.memory + (size_t)b * Tp * T2V_E + 64 * s) + (tid & 15);
#pragma unroll
        for (int i = 0; i < NJT; ++i) {
            const int j = (tid >> 4) + 16 * i;
            memr[i] = mb[(size_t)min(j, Tp - 1) * (T2V_E / 4)];      // rows >= len meet alpha = 0, rows >= T_in a zero factor
        }
    }
    __builtin_amdgcn_sched_barrier(0);      // keep the loads above in this order, ahead of everything below
    T2V_STAMP(a, 6);
#pragma unroll
    for (int u = 0; u < (2 * (TPAD + 32) + AF_THREADS - 1) / AF_THREADS; ++u) {
        const int i = tid + AF_THREADS * u;
        if (i < 2 * (TPAD + 32)) (&ap[0][0])[i] = apv[u];
    }
    __syncthreads();                      // alignment window visible
    T2V_STAMP(a, 7);
    // location features of this wave's tiles (independent of the query)
    f32x4 lacc[NI];
    if constexpr (!BIG) {
        // every LDS operand is requested before the first MFMA (left to itself the compiler waits for each read right
        // before its MFMA: one LDS latency per MFMA)
        float bop[NI][16];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int jt = min(wave + 4 * i, NJT - 1);
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                const int kk = 4 * st + g;
                bop[i][st] = ap[kk >> 5][16 * jt + c16 + (kk & 31)];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            f32x4 l0 = {0.f, 0.f, 0.f, 0.f}, l1 = {0.f, 0.f, 0.f, 0.f};          // two independent accumulator chains per tile
#pragma unroll
            for (int st = 0; st < 16; st += 2) {
                l0 = mfma16x4(areg[st], bop[i][st], l0);
                l1 = mfma16x4(areg[st + 1], bop[i][st + 1], l1);
            }
            lacc[i] = l0 + l1;
        }
    }
    T2V_STAMP(a, 8);
    // ---- 1. processed query slice: 4 source workgroups per thread; the four threads of a 16-lane row that share a
    // d-quad are summed with two DPP row rotations (no LDS crossbar), the 16 row sums go through LDS (fixed order)
    {
        float4 s4 = qpart[0];
#pragma unroll
        for (int i = 1; i < 4; ++i) { s4.x += qpart[i].x; s4.y += qpart[i].y; s4.z += qpart[i].z; s4.w += qpart[i].w; }
        s4.x = T2V_DPP_ADD(s4.x, 0x124); s4.y = T2V_DPP_ADD(s4.y, 0x124);      // row_ror:4
        s4.z = T2V_DPP_ADD(s4.z, 0x124); s4.w = T2V_DPP_ADD(s4.w, 0x124);
        s4.x = T2V_DPP_ADD(s4.x, 0x128); s4.y = T2V_DPP_ADD(s4.y, 0x128);      // row_ror:8
        s4.z = T2V_DPP_ADD(s4.z, 0x128); s4.w = T2V_DPP_ADD(s4.w, 0x128);
        if ((lane & 15) < 4) *(float4*)(scr + (tid >> 4) * 16 + 4 * (lane & 3)) = s4;      // scr[row 0..15][16 d]
    }
    __syncthreads();
    T2V_STAMP(a, 9);
    T2V_STAMP(a, 10);
    float4 q4;
    {
        float4 r[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) r[u] = *(const float4*)(scr + 16 * u + 4 * g);
#pragma unroll
        for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
            for (int u = 0; u < w; ++u) {
                r[u].x += r[u + w].x; r[u].y += r[u + w].y; r[u].z += r[u + w].z; r[u].w += r[u + w].w;
            }
        q4 = r[0];
    }
    __syncthreads();        // scr is reused by the context reduction

    T2V_STAMP(a, 1);
    // ---- 2. location features (K = 64 fused filter) + partial energies of this d-slice, tile by tile
    t2v_u64* exb = a.ex + ((size_t)b * AF_NS + s) * Tcap;
    for (int c0 = 0; c0 < Tp; c0 += TPAD) {
        if (BIG && c0 > 0) {
            __syncthreads();            // previous chunk's window fully consumed
            for (int i = tid; i < 2 * (TPAD + 32); i += AF_THREADS) {
                const int ch = i / (TPAD + 32), x = i - ch * (TPAD + 32);
                const int j = c0 + x - 15;
                float v = 0.f;
                if (j >= 0 && j < Tp) v = (ch == 0 ? a.al_prev : a.acum_prev)[(size_t)b * Tp + j];
                ap[ch][x] = v;
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int jt = wave + 4 * i;
            if (jt < NJT && c0 + 16 * jt < Tp) {
                const int j = c0 + 16 * jt + c16;
                float4 pm4;
                if constexpr (BIG) pm4 = j < Tp ? *(const float4*)(a.pm + ((size_t)b * Tp + j) * T2V_A + 16 * s + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
                else pm4 = pmr[i];
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                if constexpr (BIG) {
#pragma unroll
                    for (int st = 0; st < 16; ++st) {
                        const int kk = 4 * st + g;
                        acc = mfma16x4(areg[st], ap[kk >> 5][16 * jt + c16 + (kk & 31)], acc);
                    }
                } else {
                    acc = lacc[i];
                }
                float4 sv;
                sv.x = tanhf_(q4.x + acc[0] + pm4.x);
                sv.y = tanhf_(q4.y + acc[1] + pm4.y);
                sv.z = tanhf_(q4.z + acc[2] + pm4.z);
                sv.w = tanhf_(q4.w + acc[3] + pm4.w);
                if (a.s_save && j < Tp) *(float4*)(a.s_save + ((size_t)b * Tp + j) * T2V_A + 16 * s + 4 * g) = sv;
                float esum = vr.x * sv.x + vr.y * sv.y + vr.z * sv.z + vr.w * sv.w;
                esum += __shfl_xor(esum, 16, 64);
                esum += __shfl_xor(esum, 32, 64);
                if (g == 0 && j < Tp) af_put(exb + j, esum, a.epoch);
            }
        }
        if (!BIG) break;
    }

    T2V_STAMP(a, 2);
    // ---- 3. gather the 8 partial-energy granules of every position (tag == epoch <=> written this step), masked
    // softmax: 16-lane row max / sum with DPP, the 16 row partials through LDS (no LDS-crossbar shuffles)
    __shared__ float rmax[16], rsum[16];
    float ev0 = -INFINITY, mloc = -INFINITY;          // !BIG: this thread's single energy stays in a register
    for (int j = tid; j < Tp; j += AF_THREADS) {
        const t2v_u64* e0 = a.ex + (size_t)b * AF_NS * Tcap + j;
        float p[AF_NS];
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int i = 0; i < AF_NS; ++i) {
                const t2v_u64 x = af_get(e0 + (size_t)i * Tcap);
                p[i] = __uint_as_float((unsigned)x);
                ok = ok && (unsigned)(x >> 32) == a.epoch;
            }
            if (ok) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > AF_SPIN_LIMIT || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok_flag = 0;
                break;
            }
        }
        const float ev = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
        ev0 = j < len ? ev : -INFINITY;
        if (BIG) eall[j] = ev0;
        mloc = fmaxf(mloc, ev0);
    }
    mloc = T2V_DPP_MAX(mloc, 0xB1); mloc = T2V_DPP_MAX(mloc, 0x4E);
    mloc = T2V_DPP_MAX(mloc, 0x141); mloc = T2V_DPP_MAX(mloc, 0x140);
    if (c16 == 0) rmax[tid >> 4] = mloc;
    __syncthreads();
    if (!ok_flag) return;
    T2V_STAMP(a, 3);
    {
        float mr[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) mr[u] = rmax[u];
#pragma unroll
        for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
            for (int u = 0; u < w; ++u) mr[u] = fmaxf(mr[u], mr[u + w]);
        const float m = mr[0];
        float e0v = 0.f, sloc = 0.f;
        if constexpr (BIG) {
            for (int j = tid; j < Tp; j += AF_THREADS) { const float e = expf(eall[j] - m); eall[j] = e; sloc += e; }
        } else {
            e0v = tid < Tp ? expf(ev0 - m) : 0.f;
            sloc = e0v;
        }
        sloc = row16_sum(sloc);
        if (c16 == 0) rsum[tid >> 4] = sloc;
        __syncthreads();
        float sr[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) sr[u] = rsum[u];
#pragma unroll
        for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
            for (int u = 0; u < w; ++u) sr[u] += sr[u + w];
        const float inv = 1.0f / sr[0];
        for (int j = tid; j < Tp; j += AF_THREADS) {
            const float al = (BIG ? eall[j] : e0v) * inv;
            eall[j] = al;
            if (s == 0) {
                a.al_cur[(size_t)b * Tp + j] = al;
                const float cprev = BIG ? a.acum_prev[(size_t)b * Tp + j] : ap[1][15 + j];
                a.acum_cur[(size_t)b * Tp + j] = cprev + al;
            }
        }
    }
    __syncthreads();

    T2V_STAMP(a, 4);
    // ---- 4. context chunk s
    {
        if constexpr (BIG) {
            const int part = tid >> 6;
            float acc0 = 0.f, acc1 = 0.f;
            const float* mb = a.memory + (size_t)b * Tp * T2V_E + 64 * s + (tid & 63);
            int j = part;
            for (; j + 28 < len; j += 32) {          // 8 independent loads in flight per thread
                float mv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) mv[u] = mb[(size_t)(j + 4 * u) * T2V_E];
#pragma unroll
                for (int u = 0; u < 8; u += 2) {
                    acc0 = fmaf(eall[j + 4 * u], mv[u], acc0);
                    acc1 = fmaf(eall[j + 4 * u + 4], mv[u + 1], acc1);
                }
            }
            for (; j < len; j += 4) acc0 = fmaf(eall[j], mb[(size_t)j * T2V_E], acc0);
            __syncthreads();
            scr[tid] = acc0 + acc1;
            __syncthreads();
            if (tid < 64)
                a.xs_next[(size_t)b * T2V_XW + T2V_H + 64 * s + tid] = (scr[tid] + scr[64 + tid]) + (scr[128 + tid] + scr[192 + tid]);
        } else {
            // thread (cq, rg): partial context of columns 4cq..4cq+3 over rows rg + 16 i; 16 row groups through LDS
            float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < NJT; ++i) {
                const int j = (tid >> 4) + 16 * i;
                const float al = j < Tp ? eall[j] : 0.f;
                c4.x = fmaf(al, memr[i].x, c4.x); c4.y = fmaf(al, memr[i].y, c4.y);
                c4.z = fmaf(al, memr[i].z, c4.z); c4.w = fmaf(al, memr[i].w, c4.w);
            }
            __shared__ __attribute__((aligned(16))) float cred[16][68];
            *(float4*)&cred[tid >> 4][4 * (tid & 15)] = c4;
            __syncthreads();
            if (tid < 64) {
                float v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = cred[u][tid];
#pragma unroll
                for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
                    for (int u = 0; u < w; ++u) v[u] += v[u + w];
                a.xs_next[(size_t)b * T2V_XW + T2V_H + 64 * s + tid] = v[0];
            }
        }
    }
    T2V_STAMP(a, 5);
}

// f.ex / f.err / f.epoch are set by the caller (granule area and sync words behind the query partials, t2v_kernels.h)
void t2v_launch_attn_fwd(const AttnFwdArgs& f, int B, int T_in, hipStream_t stream) {
    const dim3 grid(B, AF_NS);
    const size_t lds = sizeof(float) * t2v_tcap(T_in);
    if (T_in <= 96) k_attn_fwd<6, false><<<grid, AF_THREADS, lds, stream>>>(f);
    else if (T_in <= 128) k_attn_fwd<8, false><<<grid, AF_THREADS, lds, stream>>>(f);
    else if (T_in <= 256) k_attn_fwd<16, false><<<grid, AF_THREADS, lds, stream>>>(f);
    else k_attn_fwd<16, true><<<grid, AF_THREADS, lds, stream>>>(f);
}
