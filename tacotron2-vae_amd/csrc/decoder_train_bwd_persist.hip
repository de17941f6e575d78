// Hand-written BPTT of the teacher-forced decoder loop as persistent launches (reference: autograd over Decoder.decode,
// model.py:346-389 / train.py:225).  The launch-per-step backward (decoder_bwd.hip: 2 launches and a 67 MB transposed
// weight stream per reverse step) stays the general path; these kernels serve the shapes of the persistent forward
// (B <= 6, T_in <= 224).
//
// The reverse recurrence splits into two chains that meet only through a time-batched GEMM:
//   D  decoder_rnn:   dgd(t+1) -> W_hh_dec^T -> dh_dec(t) -> cell backward -> dgd(t).  Nothing of the attention path
//      enters it (teacher forcing: h_dec feeds only the projection and its own next step), so it runs FIRST, for all
//      steps, as k_dchain_bwd: 256 workgroups x 4 hidden units, W_hh_dec^T (16 MB) in registers.
//   -- then ONE GEMM  E = DGD · W_ih_dec  (T*B x 4096 x 1536, own fp32 MFMA GEMM, host side) gives every step's
//      decoder_rnn contribution to d h_att(t) and d ctx(t) at once.
//   A  attention_rnn + attention (k_achain_bwd): dga(t+1) -> Wcat_att^T -> [d h_att(t) | d ctx(t)] -> attention(t)
//      backward (workgroups split over encoder positions) -> dq(t) -> W_q^T -> cell backward -> dga(t).
// Hand-offs as in decoder_train_persist.hip: every exchanged value is produced exactly once per pass, so the exchange
// buffers are pre-filled with a NaN sentinel (0xFFFFFFFF) and a word that is no longer the sentinel IS the data — 4 bytes
// per value on the wire, sc1 (write-through) stores, sc1 loads, no flags, no ordering.  Gate-gradient rows travel as
// [plane][k][items] so that a consumer's 16-byte (items 0..3) / 8-byte (items 4, 5) load is an LDS-ready GEMV operand.
#include "t2v_common.h"
#include "t2v_kernels.h"

#ifndef PBA_WHOLE_ITEM
#define PBA_WHOLE_ITEM 0      // 1 (measurement): ONE attention workgroup per item up to 96 symbols instead of 16-position slices
#endif
#ifndef PBA_DQ_DIRECT
#define PBA_DQ_DIRECT 0       // 1 (measurement): the attention_rnn workgroups sum the slices' partial dq rows themselves
#endif
#define PB_THREADS 512
#define PB_MAXB 6
#define PB_MAXT 224                   // 16- / 32-position attention slices up to here
#define PB_MAXT_LONG 576              // 96-position slices on eight waves beyond (k_achain_bwd<.., true>): at most six per item
#define PB_SPIN 400000
#define PB_SENT 0xFFFFFFFFu
#define PB_KJ (T2V_G / PB_THREADS)          // 8 gate rows per thread: k = tid + 512 j

typedef unsigned pb_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned pb_u32x2 __attribute__((ext_vector_type(2)));
typedef float pb_f32x2 __attribute__((ext_vector_type(2)));
#define PB_SC1 16
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pb_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ f32x4 pb_ld16(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, PB_SC1));
}
__device__ __forceinline__ pb_f32x2 pb_ld8(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(pb_f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, PB_SC1));
}
__device__ __forceinline__ unsigned pb_ld4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, PB_SC1);
}
__device__ __forceinline__ void pb_st16(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pb_u32x4, v), r, (int)off, 0, PB_SC1);
}
__device__ __forceinline__ void pb_st8(__amdgpu_buffer_rsrc_t r, unsigned off, pb_f32x2 v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(pb_u32x2, v), r, (int)off, 0, PB_SC1);
}
__device__ __forceinline__ void pb_st4(__amdgpu_buffer_rsrc_t r, unsigned off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)off, 0, PB_SC1);
}
__device__ __forceinline__ bool pb_ok(float v) { return __float_as_uint(v) != PB_SENT; }

// ---- a gate-gradient row (4096 gate rows x B items) in the exchange buffer / in LDS:
//   plane 0: k -> 16 bytes (items 0..3) at byte 16 k          (64 KB)
//   plane 1: k -> 8 bytes (items 4, 5)  at byte 65536 + 8 k   (32 KB, B > 4 only)
#define PB_ROW_BYTES(NB) ((NB) > 4 ? 98304u : 65536u)

// gather one row into LDS (X0: f32x4[4096], X1: f32x2[4096]); nap first, then poll the payload itself.  Returns rounds.
template <int NB>
__device__ __forceinline__ int pb_gather_row(f32x4* X0, pb_f32x2* X1, __amdgpu_buffer_rsrc_t r, unsigned row_off, int B, int nap,
                                             unsigned* err, int* flag) {
    const int tid = threadIdx.x;
    for (int i = 0; i < nap; i += 8) __builtin_amdgcn_s_sleep(8);
    f32x4 v0[PB_KJ];
    pb_f32x2 v1[PB_KJ];
    int rounds = 0;
    const int nw0 = min(B, 4), nw1 = B - 4;
    for (;;) {
#pragma unroll
        for (int j = 0; j < PB_KJ; ++j) v0[j] = pb_ld16(r, row_off + 16u * (unsigned)(tid + PB_THREADS * j));
        if (NB > 4) {
#pragma unroll
            for (int j = 0; j < PB_KJ; ++j) v1[j] = pb_ld8(r, row_off + 65536u + 8u * (unsigned)(tid + PB_THREADS * j));
        }
        bool ok = true;
#pragma unroll
        for (int j = 0; j < PB_KJ; ++j) {
            ok = ok && pb_ok(v0[j][0]) && (nw0 < 2 || pb_ok(v0[j][1])) && (nw0 < 3 || pb_ok(v0[j][2])) && (nw0 < 4 || pb_ok(v0[j][3]));
            if (NB > 4) ok = ok && pb_ok(v1[j][0]) && (nw1 < 2 || pb_ok(v1[j][1]));
        }
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(2);
        if (++rounds > PB_SPIN || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = 0;
            break;
        }
    }
#pragma unroll
    for (int j = 0; j < PB_KJ; ++j) {
        X0[tid + PB_THREADS * j] = v0[j];
        if (NB > 4) X1[tid + PB_THREADS * j] = v1[j];
    }
    return rounds;
}

// ---- the one-launch reverse pass exchanges (d c, d h) of a cell instead of its four gate gradients: a row is
//   plane 0: unit U -> 32 bytes [dc items 0..3 | dh items 0..3] at byte 32 U          (32 KB)
//   plane 1: unit U -> 16 bytes [dc items 4, 5 | dh items 4, 5] at byte 32768 + 16 U  (16 KB, B > 4 only)
// and the gate gradient of row k = r*1024 + U is F[k] * (r == 3 ? dh[U] : dc[U]) with the factor F a function of the
// forward activations alone (k_pb_factors, before the pass): half the bytes on the wire, and since thread tid consumes
// exactly the gate rows tid + 512 j — units tid and tid + 512 — it polls its own 96 bytes and nobody else's.
#define PB_DROW_BYTES(NB) ((NB) > 4 ? 49152u : 32768u)
// The factors do not wait for anybody: a phase early (their latency hides behind the wait for the other role) the row is
// copied global -> LDS as it lies (planes 0 and 1 are contiguous in both), by LDS-DMA — 1 KB per wave instruction, no
// staging registers (the 168 weight registers leave room for only a few loads in flight).  pb_build_row multiplies in
// place; a barrier must lie between the two.
// a 4-byte word per lane global -> LDS (lane i lands at ldsbase + 4 i) without a destination register; aux: cache policy
__device__ __forceinline__ void pb_dma4(const void* g, float* ldsbase, const int aux_sc1) {
    if (aux_sc1)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)ldsbase, 4, 0, PB_SC1);
    else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)ldsbase, 4, 0, 0);
}

template <int NB>
__device__ __forceinline__ void pb_park_factors(float* ldsX, const float* Frow) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int NCH = PB_ROW_BYTES(NB) / 1024;
#pragma unroll
    for (int i = 0; i < NCH / 8; ++i) {
        const int c = wave + 8 * i;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Frow + 256 * c + 4 * lane),
                                         (__attribute__((address_space(3))) void*)(ldsX + 256 * c), 16, 0, 0);
    }
}
template <int NB>
__device__ __forceinline__ int pb_build_row(f32x4* X0, pb_f32x2* X1, __amdgpu_buffer_rsrc_t r, unsigned row_off, int B,
                                            int nap, unsigned* err, int* flag) {
    const int tid = threadIdx.x;
    for (int i = 0; i < nap; i += 8) __builtin_amdgcn_s_sleep(8);
    f32x4 dc[2], dh[2], dx[2];
    int rounds = 0;
    const int nw0 = min(B, 4), nw1 = B - 4;
    for (;;) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned U = (unsigned)(tid + PB_THREADS * h);
            dc[h] = pb_ld16(r, row_off + 32u * U);
            dh[h] = pb_ld16(r, row_off + 32u * U + 16u);
            if (NB > 4) dx[h] = pb_ld16(r, row_off + 32768u + 16u * U);
        }
        bool ok = true;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            ok = ok && pb_ok(dc[h][0]) && (nw0 < 2 || pb_ok(dc[h][1])) && (nw0 < 3 || pb_ok(dc[h][2])) && (nw0 < 4 || pb_ok(dc[h][3]));
            ok = ok && pb_ok(dh[h][0]) && (nw0 < 2 || pb_ok(dh[h][1])) && (nw0 < 3 || pb_ok(dh[h][2])) && (nw0 < 4 || pb_ok(dh[h][3]));
            if (NB > 4) ok = ok && pb_ok(dx[h][0]) && pb_ok(dx[h][2]) && (nw1 < 2 || (pb_ok(dx[h][1]) && pb_ok(dx[h][3])));
        }
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(2);
        if (++rounds > PB_SPIN || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = 0;
            break;
        }
    }
#pragma unroll
    for (int j = 0; j < PB_KJ; ++j) {
        const int h = j & 1;
        const f32x4 d = (j >> 1) == 3 ? dh[h] : dc[h];
        X0[tid + PB_THREADS * j] = X0[tid + PB_THREADS * j] * d;
        if (NB > 4) X1[tid + PB_THREADS * j] = X1[tid + PB_THREADS * j] * ((j >> 1) == 3 ? pb_f32x2{dx[h][2], dx[h][3]} : pb_f32x2{dx[h][0], dx[h][1]});
    }
    return rounds;
}

// F[t][k][item] of one cell: the factor that turns (d c_t, d h_t) into the gate gradient of row k (layout of a gate row)
__global__ __launch_bounds__(256) void k_pb_factors(const float* __restrict__ G, const float* __restrict__ C, float* __restrict__ F,
                                                    int B, int T, int nb, float p, int stream_c, uint64_t seed0,
                                                    const t2v_step_params* step) {
    const uint64_t seed = t2v_step_seed(seed0, step);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)T * T2V_G) return;
    const int t = (int)(i / T2V_G), k = (int)(i % T2V_G), r = k >> 10, U = k & (T2V_H - 1);
    float f[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        f[b] = 0.f;
        if (b < B) {
            const float* gp = G + ((size_t)t * B + b) * T2V_G + U;
            const float gi = gp[0], gf = gp[T2V_H], gg = gp[2 * T2V_H], go = gp[3 * T2V_H];
            if (r == 0) f[b] = gg * gi * (1.0f - gi);
            else if (r == 2) f[b] = gi * (1.0f - gg * gg);
            else if (r == 3) f[b] = tanhf_(C[((size_t)(t + 1) * B + b) * T2V_H + U]) * go * (1.0f - go);
            else {
                float cprev = C[((size_t)t * B + b) * T2V_H + U];
                if (t > 0) cprev *= t2v_drop_scale(seed, stream_c, t - 1, (uint32_t)b * T2V_H + U, p);
                f[b] = cprev * gf * (1.0f - gf);
            }
        }
    }
    float* row = F + (size_t)t * (nb > 4 ? 6 : 4) * T2V_G;
    *(f32x4*)(row + 4 * (size_t)k) = f32x4{f[0], f[1], f[2], f[3]};
    if (nb > 4) *(pb_f32x2*)(row + 4 * T2V_G + 2 * (size_t)k) = pb_f32x2{f[4], f[5]};
}

// CP[t][U][item (nb slots)][8] of one cell: everything the cell backward of (unit U, item) needs at step t that is a function
// of the forward activations alone — {fh, fc, go (1 - tanh(c)^2), gf,  gg gi (1 - gi), c' gf (1 - gf), gi (1 - gg^2),
// tanh(c) go (1 - go)} (fh / fc: state-dropout scales of step t, c': the cell handed to step t, dropout applied).  Round 4:
// the attention_rnn workgroups used to compute these in the loop — three counter-based RNG draws and a tanh per
// (unit, item) and step, 1.5 us of every 12.9 us reverse step on the chain; now they copy 32 bytes.
__global__ __launch_bounds__(256) void k_pb_cellpre(const float* __restrict__ G, const float* __restrict__ C, float* __restrict__ CP,
                                                    int B, int T, int nb, float p, int stream_h, int stream_c, uint64_t seed0,
                                                    const t2v_step_params* step) {
    const uint64_t seed = t2v_step_seed(seed0, step);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)T * T2V_H * nb) return;
    const int b = (int)(i % nb), U = (int)((i / nb) % T2V_H), t = (int)(i / ((size_t)nb * T2V_H));
    float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0;
    if (b < B) {
        const float* gp = G + ((size_t)t * B + b) * T2V_G + U;
        const float gi = gp[0], gf = gp[T2V_H], gg = gp[2 * T2V_H], go = gp[3 * T2V_H];
        const float cc = C[((size_t)(t + 1) * B + b) * T2V_H + U];
        float cprev = C[((size_t)t * B + b) * T2V_H + U];
        const uint32_t idx = (uint32_t)b * T2V_H + U;
        const float fh = t2v_drop_scale(seed, stream_h, t, idx, p);
        const float fc = t2v_drop_scale(seed, stream_c, t, idx, p);
        if (t > 0) cprev *= t2v_drop_scale(seed, stream_c, t - 1, idx, p);
        const float tc = tanhf_(cc);
        c0 = make_float4(fh, fc, go * (1.0f - tc * tc), gf);
        c1 = make_float4(gg * gi * (1.0f - gi), cprev * gf * (1.0f - gf), gi * (1.0f - gg * gg), tc * go * (1.0f - go));
    }
    float4* o = (float4*)(CP + i * 8);
    o[0] = c0;
    o[1] = c1;
}

// acc[c][pair] += w[c][j] * x[k_j][pair] for NC output columns: packed FMAs (two items per op, weight broadcast through
// op_sel; even j = low word of the weight pair, odd j = high word) in volatile asm so the k loop keeps its shape.
template <bool ODD>
__device__ __forceinline__ void pb_pk3(pb_f32x2& a01, pb_f32x2& a23, pb_f32x2& a45, pb_f32x2 w, pb_f32x2 x01, pb_f32x2 x23, pb_f32x2 x45) {
    if (ODD)
        asm volatile("v_pk_fma_f32 %0, %3, %4, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                     "v_pk_fma_f32 %1, %3, %5, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                     "v_pk_fma_f32 %2, %3, %6, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                     : "+v"(a01), "+v"(a23), "+v"(a45) : "v"(w), "v"(x01), "v"(x23), "v"(x45));
    else
        asm volatile("v_pk_fma_f32 %0, %3, %4, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                     "v_pk_fma_f32 %1, %3, %5, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                     "v_pk_fma_f32 %2, %3, %6, %2 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                     : "+v"(a01), "+v"(a23), "+v"(a45) : "v"(w), "v"(x01), "v"(x23), "v"(x45));
}
template <int NC, int NB>
__device__ __forceinline__ void pb_gemv(const pb_f32x2 (&w)[NC][PB_KJ / 2], const f32x4* X0, const pb_f32x2* X1, pb_f32x2 (&acc)[NC][3]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c][0] = acc[c][1] = acc[c][2] = pb_f32x2{0.f, 0.f};
    f32x4 xa[PB_KJ];
    pb_f32x2 xb[PB_KJ];
#pragma unroll
    for (int j = 0; j < PB_KJ; ++j) {
        xa[j] = X0[tid + PB_THREADS * j];
        xb[j] = pb_f32x2{0.f, 0.f};
        if (NB > 4) xb[j] = X1[tid + PB_THREADS * j];
    }
#pragma unroll
    for (int j = 0; j < PB_KJ; ++j) {
        const pb_f32x2 x01 = {xa[j][0], xa[j][1]}, x23 = {xa[j][2], xa[j][3]};
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (j & 1) pb_pk3<true>(acc[c][0], acc[c][1], acc[c][2], w[c][j / 2], x01, x23, xb[j]);
            else pb_pk3<false>(acc[c][0], acc[c][1], acc[c][2], w[c][j / 2], x01, x23, xb[j]);
        }
    }
}

// Sum NV <= 32 per-thread values over the 512 threads of the workgroup: 16-lane transposing butterfly (lane c of a row
// ends with the row sums of values 2c, 2c + 1), then the 32 row partials (8 waves x 4 rows) through LDS:
// part[(wave * 4 + row) * 32 + idx].  The caller syncs and sums the 32 partials of the values it needs.
#define PB_DPP(v, CTRL) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (CTRL), 0xF, 0xF, true))
__device__ __forceinline__ void pb_reduce32(float (&v)[32], float* part) {
    const int tid = threadIdx.x, lane = tid & 63;
    const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
    float w16[16], w8[8], w4[4], w2[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float keep = b3 ? v[16 + i] : v[i], send = b3 ? v[i] : v[16 + i];
        w16[i] = keep + PB_DPP(send, 0x140);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float keep = b2 ? w16[8 + i] : w16[i], send = b2 ? w16[i] : w16[8 + i];
        w8[i] = keep + PB_DPP(send, 0x141);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float keep = b1 ? w8[4 + i] : w8[i], send = b1 ? w8[i] : w8[4 + i];
        w4[i] = keep + PB_DPP(send, 0x4E);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float keep = b0 ? w4[2 + i] : w4[i], send = b0 ? w4[i] : w4[2 + i];
        w2[i] = keep + PB_DPP(send, 0xB1);
    }
    *(float2*)(part + ((tid >> 6) * 4 + (lane >> 4)) * 32 + 2 * (lane & 15)) = make_float2(w2[0], w2[1]);
}
__device__ __forceinline__ float pb_sum32(const float* part, int idx) {
    float s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = (part[(4 * i) * 32 + idx] + part[(4 * i + 1) * 32 + idx]) + (part[(4 * i + 2) * 32 + idx] + part[(4 * i + 3) * 32 + idx]);
    return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

// Round 4, attention_rnn role: the same reduction over 16 values (2 columns x 8 item slots) at a time.  Twice the rounds of
// pb_reduce32 for the same number of DPP operations, but the GEMV's working set next to the weight registers halves
// (12 accumulator registers + 16 values instead of 24 + 32) — with 32-value rounds the role spilled loop-invariant offsets
// and every reload sat behind an s_waitcnt vmcnt(0) on the chain.  The first two levels write complementary register banks
// with bank-masked DPP adds (2 operations per output instead of select + select + add).  part[(wave * 4 + row) * 16 + idx].
__device__ __forceinline__ float pb_dpp_pair_add(float lo, float hi, const int level) {
    float t;
    if (level == 0)
        asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                     "v_add_f32_dpp %0, %2, %2 row_mirror row_mask:0xf bank_mask:0xc" : "=&v"(t) : "v"(lo), "v"(hi));
    else
        asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                     "v_add_f32_dpp %0, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xa" : "=&v"(t) : "v"(lo), "v"(hi));
    return t;
}
__device__ __forceinline__ void pb_reduce16(float (&v)[16], float* part) {
    const int tid = threadIdx.x, lane = tid & 63;
    const bool b1 = lane & 2, b0 = lane & 1;
    float w8[8], w4[4], w2[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) w8[i] = pb_dpp_pair_add(v[i], v[8 + i], 0);          // lanes 0..7 of a row: values i, lanes 8..15: 8 + i
#pragma unroll
    for (int i = 0; i < 4; ++i) w4[i] = pb_dpp_pair_add(w8[i], w8[4 + i], 1);        // bit 2 of the lane picks the half
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float keep = b1 ? w4[2 + i] : w4[i], send = b1 ? w4[i] : w4[2 + i];
        w2[i] = keep + PB_DPP(send, 0x4E);
    }
    const float keep = b0 ? w2[1] : w2[0], send = b0 ? w2[0] : w2[1];
    part[((tid >> 6) * 4 + (lane >> 4)) * 16 + (lane & 15)] = keep + PB_DPP(send, 0xB1);
}
__device__ __forceinline__ float pb_sum16(const float* part, int idx) {
    float s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = (part[(4 * i) * 16 + idx] + part[(4 * i + 1) * 16 + idx]) + (part[(4 * i + 2) * 16 + idx] + part[(4 * i + 3) * 16 + idx]);
    return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

// ============================================================================================ chain D: decoder_rnn
struct PBDArgs {
    const float* w_hh_dec;      // (4096,1024)
    const float* dHC;           // (T,B,1536): [:, :, :1024] = grad wrt h_dec from the projection
    const float* GD;            // (T,B,4096) gate activations of decoder_rnn
    const float* CD;            // (T+1,B,1024): CD[t+1] = c_dec(t) (pre-dropout), CD[0] = 0
    float* DGD;                 // (T,B,4096) out
    float* GX;                  // exchange: T rows of PB_ROW_BYTES, sentinel-filled
    unsigned* err;
    int B, T;
    float p_dec;
    uint64_t seed;
    const t2v_step_params* step;
};

template <int NB>
__global__ __launch_bounds__(PB_THREADS) void k_dchain_bwd(PBDArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x4* X0 = (f32x4*)lds;                               // [4096] items 0..3
    pb_f32x2* X1 = (pb_f32x2*)(lds + 4 * T2V_G);           // [4096] items 4, 5
    float* part = lds + (NB > 4 ? 6 : 4) * T2V_G;          // [32][32]
    float* stage = part + 32 * 32;                         // [4 units][4 gates][8 items]
    int* flag = (int*)(stage + 128);
    const uint64_t seed = t2v_step_seed(a.seed, a.step);
    const int wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int B = a.B, T = a.T;
    const __amdgpu_buffer_rsrc_t rX = pb_rsrc(a.GX);
    // W_hh_dec^T columns of this workgroup's 4 units: w[u][j] = W_hh_dec[k = tid + 512 j][4 wg + u]
    pb_f32x2 w[4][PB_KJ / 2];
#pragma unroll
    for (int j = 0; j < PB_KJ; ++j) {
        const float4 w4 = *(const float4*)(a.w_hh_dec + (size_t)(tid + PB_THREADS * j) * T2V_H + 4 * wg);
        w[0][j / 2][j & 1] = w4.x; w[1][j / 2][j & 1] = w4.y; w[2][j / 2][j & 1] = w4.z; w[3][j / 2][j & 1] = w4.w;
    }
    if (tid == 0) flag[0] = 1;
    // cell threads: tid = u * 8 + b (u < 4, b < B)
    const int cu = tid >> 3, cb = tid & 7;
    const bool cell_on = tid < 32 && cb < B;
    const int U = 4 * wg + (cu & 3);
    const uint32_t idx = (uint32_t)cb * T2V_H + U;
    float dcd = 0.f;                                        // grad wrt the (post-dropout) cell handed to step t+1
    int nap = 0;
    __syncthreads();

    for (int t = T - 1; t >= 0; --t) {
        float yd = 0.f;
        if (t < T - 1) {
            // dgd(t+1) from everybody, then this workgroup's 4 columns of W_hh_dec^T · dgd(t+1)
            const int rounds = pb_gather_row<NB>(X0, X1, rX, (unsigned)(t + 1) * PB_ROW_BYTES(NB), B, nap, a.err, flag);
            nap = t2v_adapt_nap(nap, rounds);
            __syncthreads();
            if (flag[0] != 1) return;
            pb_f32x2 acc[4][3];
            pb_gemv<4, NB>(w, X0, X1, acc);
            float v[32];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 3; ++i) { v[u * 8 + 2 * i] = acc[u][i][0]; v[u * 8 + 2 * i + 1] = acc[u][i][1]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u * 8 + 6] = v[u * 8 + 7] = 0.f;
            pb_reduce32(v, part);
            __syncthreads();
            if (cell_on) yd = pb_sum32(part, tid);
        }
        if (tid < 64) {
            // ---- cell backward of decoder_rnn(t) for (unit U, item cb)
            float dg[4] = {0.f, 0.f, 0.f, 0.f};
            if (cell_on) {
                const float dh = a.dHC[((size_t)t * B + cb) * (T2V_H + T2V_E) + U] + yd;
                const float* gp = a.GD + ((size_t)t * B + cb) * T2V_G + U;
                const float gi = gp[0], gf = gp[T2V_H], gg = gp[2 * T2V_H], go = gp[3 * T2V_H];
                const float cdc = a.CD[((size_t)(t + 1) * B + cb) * T2V_H + U];
                float cprev = a.CD[((size_t)t * B + cb) * T2V_H + U];
                const float fh = t2v_drop_scale(seed, T2V_RNG_DEC_H, t, idx, a.p_dec);
                const float fc = t2v_drop_scale(seed, T2V_RNG_DEC_C, t, idx, a.p_dec);
                if (t > 0) cprev *= t2v_drop_scale(seed, T2V_RNG_DEC_C, t - 1, idx, a.p_dec);
                const float tc = tanhf_(cdc);
                const float dht = dh * fh;
                const float dct = dcd * fc + dht * go * (1.0f - tc * tc);
                dg[0] = dct * gg * gi * (1.0f - gi);
                dg[1] = dct * cprev * gf * (1.0f - gf);
                dg[2] = dct * gi * (1.0f - gg * gg);
                dg[3] = dht * tc * go * (1.0f - go);
                dcd = dct * gf;
                float* o = a.DGD + ((size_t)t * B + cb) * T2V_G + U;
                o[0] = dg[0]; o[T2V_H] = dg[1]; o[2 * T2V_H] = dg[2]; o[3 * T2V_H] = dg[3];
            }
            // publish the 16 gate-gradient rows of this workgroup (only while somebody still needs them: t > 0)
            if (t > 0) {
                if (tid < 32) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) stage[((cu & 3) * 4 + r) * 8 + cb] = cb < B ? dg[r] : 0.f;
                }
                // (same wave: LDS operations of one wave complete in order)
                if (tid < 16) {
                    const int u = tid >> 2, r = tid & 3;
                    const int k = r * T2V_H + 4 * wg + u;
                    const float* sp = stage + (u * 4 + r) * 8;
                    pb_st16(rX, (unsigned)t * PB_ROW_BYTES(NB) + 16u * (unsigned)k, f32x4{sp[0], sp[1], sp[2], sp[3]});
                    if (NB > 4) pb_st8(rX, (unsigned)t * PB_ROW_BYTES(NB) + 65536u + 8u * (unsigned)k, pb_f32x2{sp[4], sp[5]});
                }
            }
        }
        __syncthreads();            // part / stage / X are reused by the next step
    }
}

// sentinel fill (16 bytes per thread and iteration)
__global__ __launch_bounds__(256) void k_pb_fill(uint4* p, size_t n16) {
    const uint4 s = {PB_SENT, PB_SENT, PB_SENT, PB_SENT};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = s;
}

static size_t pb_row_bytes(int B) { return B > 4 ? 98304u : 65536u; }
static size_t pbd_lds_bytes(int B) { return sizeof(float) * ((B > 4 ? 6 : 4) * T2V_G + 32 * 32 + 128 + 4); }
#define PB_LDS_MAX (160 * 1024)

static int pb_device_ok() {
    static int cus = -1;
    if (cus < 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
        cus = prop.multiProcessorCount;
    }
    return cus >= T2V_NWG;
}

extern "C" int t2v_decoder_bwd_persist_supported(int B, int T_in) {
    if (!(B >= 1 && B <= PB_MAXB && T_in >= 1 && T_in <= PB_MAXT_LONG)) return 0;
    return pb_device_ok();
}
// floats of exchange scratch for the decoder_rnn chain (t2v_decoder_bwd_dchain)
extern "C" long t2v_decoder_bwd_dchain_scratch_floats(int B, int T_out) {
    if (B < 1 || B > PB_MAXB || T_out < 1) return 0;
    return (long)((size_t)T_out * pb_row_bytes(B) / 4);
}

extern "C" int t2v_decoder_bwd_dchain(const float* w_hh_dec, const float* dHC, const float* GD, const float* CD, float* DGD,
                                      float* scratch, uint32_t* err_word, int B, int T_out, float p_dec, uint64_t seed, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!w_hh_dec || !dHC || !GD || !CD || !DGD || !scratch || !err_word || B < 1 || B > PB_MAXB || T_out < 1 || !pb_device_ok())
        return T2V_ERR_ARG;
    if (((uintptr_t)scratch & 15) || (size_t)T_out * pb_row_bytes(B) >= 0x7fffffffull) return T2V_ERR_ARG;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_dchain_bwd<4>, hipFuncAttributeMaxDynamicSharedMemorySize, PB_LDS_MAX) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_dchain_bwd<6>, hipFuncAttributeMaxDynamicSharedMemorySize, PB_LDS_MAX) != hipSuccess)
            return t2v_check_launch();
        attr_set = true;
    }
    (void)hipMemsetAsync(err_word, 0, sizeof(uint32_t), stream);
    k_pb_fill<<<1024, 256, 0, stream>>>((uint4*)scratch, (size_t)T_out * pb_row_bytes(B) / 16);
    PBDArgs a;
    a.w_hh_dec = w_hh_dec; a.dHC = dHC; a.GD = GD; a.CD = CD; a.DGD = DGD; a.GX = scratch; a.err = err_word;
    a.B = B; a.T = T_out; a.p_dec = p_dec; a.seed = seed; a.step = t2v_step_for(stream);
    if (B > 4) k_dchain_bwd<6><<<T2V_NWG, PB_THREADS, pbd_lds_bytes(B), stream>>>(a);
    else k_dchain_bwd<4><<<T2V_NWG, PB_THREADS, pbd_lds_bytes(B), stream>>>(a);
    return t2v_check_launch();
}

// ======================================================================= chain A: attention_rnn + attention, one launch
// Roles (one workgroup per CU):
//   T : workgroups [0, B*S) — attention(t) backward of item b, encoder positions [s*JS, s*JS + JS) (the position-split
//       body of decoder_bwd.hip: softmax / tanh / fused-location-filter backward), operands that do not change over the
//       pass (memory rows, W_comb^T tile, v) resident in registers, the cumulative-weights gradient resident in LDS.
//       Runs on the first 256 threads; waves 4..7 only keep the barriers company.
//   L : the other workgroups — workgroup j owns hidden units [j*1024/NL, ..) (4 or 5) and context columns
//       [j*512/NL, ..) (2 or 3): their columns of Wcat_att^T (attention_rnn) AND of Wcat_dec^T (decoder_rnn) live in
//       registers (thread = 8 of the 4096 gate rows, <= 21 columns).  decoder_rnn's chain runs ONE STEP AHEAD in the
//       shadow of attention(t): cell D(t-1), all-gather of dgd(t-1), D-GEMV(t-1) happen while the T workgroups work.
// Per reverse step the dependency chain is
//   all-gather dga(t+1) -> A-GEMV -> [d ctx(t)] -> hop -> attention(t) backward -> [dq(t)] -> hop -> W_q^T dq + cell A(t)
//   -> [dga(t)] -> all-gather ...
struct PBAArgs {
    // weights
    const float* w_ih_att; const float* w_hh_att; const float* w_ih_dec; const float* w_hh_dec;
    const float* wq;            // (128,1024)
    const float* wcomb;         // fused location filter (both copies)
    const float* v;             // (128)
    // saved by the forward pass
    const float* memory; const float* XS; const float* CA; const float* CD; const float* GA; const float* GD; const float* AL;
    float* S;                   // (T,B,T_in,128) in: tanh outputs, out: dpre
    const float* dHC;           // (T,B,1536)
    // outputs
    float* DGA; float* DGD; float* DCTX; float* DV;       // DV (B,S,128)
    // exchange (sentinel-filled): gate-gradient rows of both cells, context gradients, dq partials, window partials
    float* GXA; float* GXD; float* CX; float* DQX; float* GPX; float* EX;
    float* DQT;                 // (T,B,128) dq(t) summed over the position slices (published by slice 0 of each item)
    const float* CPA;           // (T,1024,NB,8) cell-layout factors of attention_rnn (k_pb_cellpre)
    const float* CPD;           // the same for decoder_rnn
    const float* FA; const float* FD;    // gate-gradient factors of both cells (k_pb_factors)      // EX (T,B,1536): E(t) = Wcat_dec[:, :1536]^T dgd(t)
    unsigned* err;
    int B, T_in, T, S_sl;
    float p_att, p_dec;
    uint64_t seed;
    const t2v_step_params* step;
    unsigned long long* prof;
};
// phase profile (tools/dbg/persist_bwd_prof.py): slot I accumulates, over all steps, the cycles since the previous stamp
#define PBA_STAMP(COND, I) do { if (a.prof && (COND) && threadIdx.x == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); \
        lprof_[(I)] += now_ - tprev_; tprev_ = now_; } } while (0)
// per-workgroup time line of ONE step (t = T/2) on the chip-wide 100 MHz counter: prof[64 + workgroup * 8 + slot]
#define PBA_RT(SLOT) do { if (a.prof && t == a.T / 2 && threadIdx.x == 0) a.prof[64 + blockIdx.x * 8 + (SLOT)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define PBA_PROF_INIT(FLAGP) unsigned long long* lprof_ = (unsigned long long*)((FLAGP) + 4); \
    if (threadIdx.x < 16) lprof_[threadIdx.x] = 0ull
#define PBA_PROF_FLUSH(COND, I0, N) do { if (a.prof && (COND) && threadIdx.x < (N)) a.prof[(I0) + threadIdx.x] = lprof_[(I0) + threadIdx.x]; } while (0)

// context-gradient row of a step in CX: [item (4 or 8 slots)][512 columns] — item-major (round 4): an attention_rnn workgroup
// publishes its <= 7 columns of an item as ONE 16-byte store (+ <= 3 words) instead of 4-byte stores 16 bytes apart, and an
// attention workgroup polls its item's 2 KB with 128 sixteen-byte loads instead of 512 four-byte loads spread over 8 KB
// (second form, round 4: [item][attention_rnn workgroup (128 slots)][8 floats] — every producer owns an aligned 32-byte slot,
// no cache line sector is ever written by two workgroups)
#define PB_CX_ROW_BYTES(NB) ((NB) > 4 ? 32768u : 16384u)

__host__ __device__ static inline int pba_na(int NL);
// JS: positions per slice (16 / 32: position-split slices on 4 waves; 96 (round 4): ONE workgroup per item on all 8 waves for
// T_in <= 96 — no partial dq rows to sum, no window partials to exchange: one dependent hand-off less per reverse step).
// NWV: waves that compute (4 or 8).  GPW: floats per channel of a slice's window-partial row in GPX.
template <int JS, int NWV>
__device__ __forceinline__ void pba_attention_role(const PBAArgs& a, float* lds, const int b, const int s, const int NB) {
    constexpr int NJT = JS / 16;
    constexpr int PW = JS + 30;
    constexpr int GPW = PW <= 64 ? 64 : 128;
    constexpr int NRG = 2 * NWV;                     // row groups of 32 lanes in the dpre loop
    constexpr int DPS = JS == 96 ? JS + 17 : JS + 1; // row stride of dpT: = 17 mod 32, the four k-rows of an MFMA operand read land in
                                                     // disjoint banks (JS + 1 = 97 = 1 mod 32 made that read 4-way conflicted)
    static_assert(JS % NRG == 0 && JS % NWV == 0 && PW <= GPW, "slice geometry");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool act = tid < 64 * NWV;
    const int g = lane >> 4, c16 = lane & 15;
    const int B = a.B, Tp = a.T_in, T = a.T, S = a.S_sl, j0 = s * JS;
    const int Tcap = (Tp + 15) & ~15;
    const int nown = min(JS, Tp - j0);
    const int NAw = pba_na(T2V_NWG - B * S);          // attention_rnn workgroups: producers of the context gradient
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
    // (8-wave form: the W_comb^T operand tile of the location backward lives in LDS, not in 32 registers per thread — next to the
    // 96 registers of memory rows they spilled)
    constexpr bool AREG_LDS = NWV == 8;
    float* wcs = (float*)(flag + 40);         // [64 rows (c,k)][128] when AREG_LDS
    const __amdgpu_buffer_rsrc_t rC = pb_rsrc(a.CX), rQ = pb_rsrc(a.DQX), rP = pb_rsrc(a.GPX), rQT = pb_rsrc(a.DQT);
    // ---- operands resident for the whole pass
    const int d4 = tid & 31, rg = (tid >> 5) & (NRG - 1);
    float4 m0[JS / NWV], m1[JS / NWV];
    float areg[AREG_LDS ? 1 : 32];
    float4 vd4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act) {
#pragma unroll
        for (int r = 0; r < JS / NWV; ++r) {
            const int jl = wave + NWV * r;
            const float* mrow = a.memory + ((size_t)b * Tp + j0 + (jl < nown ? jl : 0)) * T2V_E + lane * 4;
            m0[r] = *(const float4*)mrow;
            m1[r] = *(const float4*)(mrow + 256);
        }
        if (!AREG_LDS) {
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
        for (int i = tid; i < 64 * 128; i += PB_THREADS)          // row stride 132, 33 floats per k-group: conflict-free operand reads
            wcs[(i >> 7) * 132 + ((i & 127) >> 5) * 33 + (i & 31)] = a.wcomb[T2V_A * 64 + i];
    for (int j = tid; j < Tcap; j += PB_THREADS) gcum[j] = 0.f;
    if (tid == 0) flag[0] = 1;
    float dvacc = 0.f;                          // tid < 128: running dv[tid] of this slice
    int nap = 0;
    PBA_PROF_INIT(flag);
    __syncthreads();
    unsigned long long tprev_ = __builtin_readcyclecounter();

    for (int t = T - 1; t >= 0; --t) {
        // (thread-derived indices are recomputed per step from an opaque copy — see the attention_rnn role: hoisted, they are
        // spilled next to the 96 registers of memory rows, and every reload is a drain of the wave's memory queue)
        int tid_op = threadIdx.x;
        asm volatile("" : "+v"(tid_op));
        const int tid = tid_op, lane = tid & 63, wave = tid >> 6;
        const bool act = tid < 64 * NWV;
        const int g = lane >> 4, c16 = lane & 15;
        const int d4 = tid & 31, rg = (tid >> 5) & (NRG - 1);
        PBA_STAMP(blockIdx.x == 0, 8);
        // ---- operands that do not wait for the context gradient: tanh outputs, alpha(t), ctx(t), window partials of step t+1
        float4 sreg[JS / NRG];
        float2 ctx2 = make_float2(0.f, 0.f);
        if (act) {
            const float* sp = a.S + (((size_t)t * B + b) * Tp + j0) * T2V_A + 4 * d4;
#pragma unroll
            for (int i = 0; i < JS / NRG; ++i) {
                const int jl = rg + NRG * i;
                sreg[i] = *(const float4*)(sp + (size_t)min(jl, nown - 1) * T2V_A);
            }
            if (tid < 256) ctx2 = *(const float2*)(a.XS + ((size_t)(t + 1) * B + b) * T2V_XW + T2V_H + 2 * tid);
        }
        float dot_g = 0.f;
        for (int j = tid; j < Tp; j += PB_THREADS) {
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
                        x0 = pb_ld4(rP, off);
                        x1 = pb_ld4(rP, off + 4u * GPW);
                        if (x0 != PB_SENT && x1 != PB_SENT) break;
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > PB_SPIN || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                            __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            flag[0] = 0;
                            break;
                        }
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
        // ---- the context gradient of this item (16 bytes per thread of waves 0 and 1), nap first
        if (tid < 2 * NAw) {
            const unsigned off = (unsigned)t * PB_CX_ROW_BYTES(NB) + (unsigned)b * 4096u + 16u * (unsigned)tid;
            for (int i = 0; i < nap; i += 8) __builtin_amdgcn_s_sleep(8);
            // TWO polls in flight, half a round trip apart (round 4): with one, a row that lands just after a poll left is
            // only seen a full memory round trip later — this hand-off is on the chain of every reverse step
            f32x4 x, x0 = pb_ld16(rC, off);
            __builtin_amdgcn_s_sleep(4);
            f32x4 x1 = pb_ld16(rC, off);
            int rounds = 0;
            for (;;) {
                if (__all(pb_ok(x0[0]) && pb_ok(x0[1]) && pb_ok(x0[2]) && pb_ok(x0[3]))) { x = x0; break; }
                x0 = pb_ld16(rC, off);
                if (__all(pb_ok(x1[0]) && pb_ok(x1[1]) && pb_ok(x1[2]) && pb_ok(x1[3]))) { x = x1; break; }
                x1 = pb_ld16(rC, off);
                if (++rounds > PB_SPIN || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    flag[0] = 0;
                    x = x0;
                    break;
                }
            }
            nap = t2v_adapt_nap(nap, rounds);
            // slot (workgroup ja, half h) holds columns c0(ja) + 4 h .. of this item
            const int ja = tid >> 1, h4 = 4 * (tid & 1);
            const int cc0 = (ja * T2V_E) / NAw, ncc = ((ja + 1) * T2V_E) / NAw - cc0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (h4 + i < ncc) dctx[cc0 + h4 + i] = x[i];
        }
        __syncthreads();
        if (flag[0] != 1) return;
        PBA_STAMP(blockIdx.x == 0, 9);
        PBA_RT(0);
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
            float q = 0.f, vv = 0.f;
            {
                const float* p = rq + tid;
                q = ((p[0] + p[T2V_A]) + (p[2 * T2V_A] + p[3 * T2V_A])) + ((p[4 * T2V_A] + p[5 * T2V_A]) + (p[6 * T2V_A] + p[7 * T2V_A]));
                const float* p2 = rv + tid;
                vv = ((p2[0] + p2[T2V_A]) + (p2[2 * T2V_A] + p2[3 * T2V_A])) + ((p2[4 * T2V_A] + p2[5 * T2V_A]) + (p2[6 * T2V_A] + p2[7 * T2V_A]));
                if (NRG > 8) {
                    p += 8 * T2V_A; p2 += 8 * T2V_A;
                    q += ((p[0] + p[T2V_A]) + (p[2 * T2V_A] + p[3 * T2V_A])) + ((p[4 * T2V_A] + p[5 * T2V_A]) + (p[6 * T2V_A] + p[7 * T2V_A]));
                    vv += ((p2[0] + p2[T2V_A]) + (p2[2 * T2V_A] + p2[3 * T2V_A])) + ((p2[4 * T2V_A] + p2[5 * T2V_A]) + (p2[6 * T2V_A] + p2[7 * T2V_A]));
                }
            }
            pb_st4(rQ, (unsigned)(((t * B + b) * S + s) * T2V_A + tid) * 4u, q);       // partial row (the d W_q GEMM reads them later)
            dvacc += vv;
            if (s == 0 && !PBA_DQ_DIRECT) {
                // Round 4: slice 0 of an item sums the S partial rows in slice order and publishes ONE row per item.  The ≥ 79
                // attention_rnn workgroups used to pull all B*S partial rows each (18 KB per workgroup and step through the
                // ≈ 11 B/cycle a CU gets from beyond its L2: 2.7 us from "published" to "gathered"); now they pull B rows (3 KB)
                // (all partial rows are requested in ONE round: a round trip per slice would cost 0.45 us each)
                constexpr int SMAX = PB_MAXT / 16;              // 14 slices at most
                const unsigned off0 = (unsigned)(((t * B + b) * S) * T2V_A + tid) * 4u;
                unsigned x[SMAX];
                int spins = 0;
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int s2 = 1; s2 < SMAX; ++s2) {
                        x[s2] = pb_ld4(rQ, off0 + (unsigned)(min(s2, S - 1) * T2V_A) * 4u);
                    }
#pragma unroll
                    for (int s2 = 1; s2 < SMAX; ++s2) ok = ok && (s2 >= S || x[s2] != PB_SENT);
                    if (__all(ok)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > PB_SPIN || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        flag[0] = 0;
                        break;
                    }
                }
                float tot = q;
#pragma unroll
                for (int s2 = 1; s2 < SMAX; ++s2) tot += s2 < S ? __uint_as_float(x[s2]) : 0.f;
                pb_st4(rQT, (unsigned)((t * B + b) * T2V_A + tid) * 4u, tot);           // the cell workgroups wait for this
            }
        }
        PBA_STAMP(blockIdx.x == 0, 10);
        PBA_RT(1);
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
            for (int jt = (wave >> 2); jt < NJT; jt += NWV / 4) {     // (8 waves: waves 4..7 take the odd position tiles)
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
                pb_st4(rP, (unsigned)(((t * B + b) * S + s) * (2 * GPW) + c * GPW + jj) * 4u, (acc0 + acc1) + (acc2 + acc3));
            }
        }
        __syncthreads();
        PBA_STAMP(blockIdx.x == 0, 11);
    }
    if (tid < T2V_A) a.DV[((size_t)b * S + s) * T2V_A + tid] = dvacc;
    PBA_PROF_FLUSH(blockIdx.x == 0, 8, 4);
}

// y[C0 + c][pair] = sum_j w[C0 + c][j] * x[k_j][pair] for NC of the thread's columns; one LDS operand per gate row with two
// more in flight (the 168 weight registers leave no room for all eight)
template <int NCT, int C0, int NC, int NB, int VS = 8>
__device__ __forceinline__ void pb_gemv_cols(const pb_f32x2 (&w)[NCT][PB_KJ / 2], const f32x4* X0, const pb_f32x2* X1, float (&v)[32]) {
    const int tid = threadIdx.x;
    pb_f32x2 acc[NC][3];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c][0] = acc[c][1] = acc[c][2] = pb_f32x2{0.f, 0.f};
    constexpr int PF = 3;
    f32x4 xa[PF];
    pb_f32x2 xb[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        xa[d] = X0[tid + PB_THREADS * d];
        xb[d] = pb_f32x2{0.f, 0.f};
        if (NB > 4) xb[d] = X1[tid + PB_THREADS * d];
    }
#pragma unroll
    for (int j = 0; j < PB_KJ; ++j) {
        const f32x4 xc = xa[j % PF];
        const pb_f32x2 yc = xb[j % PF];
        const pb_f32x2 x01 = {xc[0], xc[1]}, x23 = {xc[2], xc[3]};
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (j & 1) pb_pk3<true>(acc[c][0], acc[c][1], acc[c][2], w[C0 + c][j / 2], x01, x23, yc);
            else pb_pk3<false>(acc[c][0], acc[c][1], acc[c][2], w[C0 + c][j / 2], x01, x23, yc);
        }
        if (j + PF < PB_KJ) {
            xa[j % PF] = X0[tid + PB_THREADS * (j + PF)];
            if (NB > 4) xb[j % PF] = X1[tid + PB_THREADS * (j + PF)];
        }
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < 3; ++i) { v[c * VS + 2 * i] = acc[c][i][0]; v[c * VS + 2 * i + 1] = acc[c][i][1]; }
}

// the same for <= 2 columns into v[16] (two columns x 8 item slots): the operand rows are re-read per round
template <int NCT, int C0, int NC, int NB>
__device__ __forceinline__ void pb_gemv_cols16(const pb_f32x2 (&w)[NCT][PB_KJ / 2], const f32x4* X0, const pb_f32x2* X1, float (&v)[16]) {
    static_assert(NC >= 1 && NC <= 2, "one or two columns");
    const int tid = threadIdx.x;
    pb_f32x2 acc[NC][3];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c][0] = acc[c][1] = acc[c][2] = pb_f32x2{0.f, 0.f};
    constexpr int PF = 4;
    f32x4 xa[PF];
    pb_f32x2 xb[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        xa[d] = X0[tid + PB_THREADS * d];
        xb[d] = pb_f32x2{0.f, 0.f};
        if (NB > 4) xb[d] = X1[tid + PB_THREADS * d];
    }
#pragma unroll
    for (int j = 0; j < PB_KJ; ++j) {
        const f32x4 xc = xa[j % PF];
        const pb_f32x2 yc = xb[j % PF];
        const pb_f32x2 x01 = {xc[0], xc[1]}, x23 = {xc[2], xc[3]};
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (j & 1) pb_pk3<true>(acc[c][0], acc[c][1], acc[c][2], w[C0 + c][j / 2], x01, x23, yc);
            else pb_pk3<false>(acc[c][0], acc[c][1], acc[c][2], w[C0 + c][j / 2], x01, x23, yc);
        }
        if (j + PF < PB_KJ) {
            xa[j % PF] = X0[tid + PB_THREADS * (j + PF)];
            if (NB > 4) xb[j % PF] = X1[tid + PB_THREADS * (j + PF)];
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < 3; ++i) { v[c * 8 + 2 * i] = acc[c][i][0]; v[c * 8 + 2 * i + 1] = acc[c][i][1]; }
}

// columns [C0, C0 + N) two at a time: round r leaves its 32 row partials x 16 values at part + 512 r
template <int NCT, int C0, int N, int NB, int R = 0>
__device__ __forceinline__ void pba_rounds16(const pb_f32x2 (&w)[NCT][PB_KJ / 2], const f32x4* X0, const pb_f32x2* X1, float* part) {
    if constexpr (2 * R < N) {
        float v[16];
        pb_gemv_cols16<NCT, C0 + 2 * R, (N - 2 * R >= 2 ? 2 : 1), NB>(w, X0, X1, v);
        pb_reduce16(v, part + 512 * R);
        pba_rounds16<NCT, C0, N, NB, R + 1>(w, X0, X1, part);
    }
}

// ---- role split of the LSTM workgroups (NL = 256 - B*S of them).  One workgroup set per cell halves the gate-gradient
// all-gather (a row of 4096 x B values goes to the workgroups of ITS cell only) and decouples the two chains:
//   A role, NA = 3/8 of NL workgroups: attention_rnn.  Workgroup ja owns hidden units [ja*1024/NA, ..) (<= 14) and context
//           columns [ja*512/NA, ..) (<= 7): their columns of Wcat_att^T in registers.  This is the per-step chain.
//   D role, ND = NL - NA workgroups: decoder_rnn.  Workgroup jd owns units [jd*1024/ND, ..) (<= 8, both the h_att-input
//           and the recurrent column of each) and context columns [jd*512/ND, ..) (<= 4): columns of Wcat_dec^T.  Its
//           loop needs only dHC and its own recurrence, so it FREE-RUNS ahead of the A chain and leaves
//           E(t) = Wcat_dec[:, :1536]^T dgd(t) — decoder_rnn's contribution to d h_att(t) / d ctx(t) — in a sentinel-filled
//           array the A workgroups read when they get there.
#define PBA_NUA 13
#define PBA_NCA 7
#define PBA_NUD 8
#define PBA_NCD 4
// (>= 79 workgroups: at most 13 units and 7 context columns each — 20 columns = 160 weight registers per thread)
// Round 4: the decoder_rnn role needs 128 workgroups to stay at 8 units + 4 context columns (20 columns) each; every other
// LSTM workgroup goes to the attention_rnn role, the per-step chain.  From 86 workgroups on it holds <= 12 units + 6 context
// columns (18 columns = 144 weight registers: the slim instantiation — one reduction round less per step and 16 registers
// fewer next to the GEMV's working set); below that the 13 + 7 form.
__host__ __device__ static inline int pba_na(int NL) {
    if (NL - 128 >= 86) return NL - 128;
    const int n = (3 * NL + 4) / 8;
    return n < 79 ? 79 : n;
}

// final sums of a column-grouped GEMV: NCOL columns x 8 item slots -> ysum[col * 8 + b]
__device__ __forceinline__ void pba_finish_sums(const float* part, float* ysum, int ncol) {
    const int tid = threadIdx.x;
    if (tid < ncol * 8) ysum[tid] = pb_sum32(part + (tid >> 5) * 1024, tid & 31);
}

// poll one word of a sentinel-filled array until it is written (bounded)
__device__ __forceinline__ float pba_wait_word(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned* err, int* flag) {
    unsigned x;
    int spins = 0;
    for (;;) {
        x = pb_ld4(r, off);
        if (x != PB_SENT) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > PB_SPIN || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = 0;
            break;
        }
    }
    return __uint_as_float(x);
}

// stage[u][{dc, dh}][8 items] -> one 16-byte (+ one 8-byte) write-through store per unit and quantity
template <int NB>
__device__ __forceinline__ void pba_publish_rows(__amdgpu_buffer_rsrc_t r, unsigned row_off, const float* stage, int u0, int nu) {
    const int tid = threadIdx.x;
    if (tid < 2 * nu) {
        const int u = tid >> 1, q = tid & 1;
        const float* sp = stage + (u * 2 + q) * 8;
        pb_st16(r, row_off + 32u * (unsigned)(u0 + u) + 16u * (unsigned)q, f32x4{sp[0], sp[1], sp[2], sp[3]});
        if (NB > 4) pb_st8(r, row_off + 32768u + 16u * (unsigned)(u0 + u) + 8u * (unsigned)q, pb_f32x2{sp[4], sp[5]});
    }
}

// ------------------------------------------------------------------------------------------------ D role (free-running)
template <int NB>
__device__ __forceinline__ void pba_decoder_role(const PBAArgs& a, float* lds, const int jd, const int ND) {
    const int B = a.B, T = a.T;
    f32x4* X0 = (f32x4*)lds;
    pb_f32x2* X1 = (pb_f32x2*)(lds + 4 * T2V_G);
    float* part = lds + (NB > 4 ? 6 : 4) * T2V_G;          // [10 rounds][32 row partials][16]
    float* stage = part + 10 * 512;                        // [8 units][2][8]
    float* cpd = stage + 256;                              // [2][64 rows][8] cell-layout factors of steps td, td-1 (k_pb_cellpre)
    float* dhcs = cpd + 1024;                              // [2][64] dHC words of the cell threads
    int* flag = (int*)(dhcs + 128);
    const int u0 = (jd * T2V_H) / ND, nu = ((jd + 1) * T2V_H) / ND - u0;      // <= 8 units
    const int c0 = (jd * T2V_E) / ND, nc = ((jd + 1) * T2V_E) / ND - c0;      // <= 4 context columns
    const __amdgpu_buffer_rsrc_t rD = pb_rsrc(a.GXD), rE = pb_rsrc(a.EX), rDG = pb_rsrc(a.DGD);
    // columns: [0, 8) recurrent (W_hh_dec[k][U]), [8, 16) h_att input (W_ih_dec[k][U]), [16, 20) ctx input (W_ih_dec[k][1024 + C])
    pb_f32x2 w[20][PB_KJ / 2];
    {
        const int tid = threadIdx.x;
#pragma unroll
        for (int jj = 0; jj < PB_KJ; ++jj) {
            const size_t k = (size_t)(tid + PB_THREADS * jj);
#pragma unroll
            for (int u = 0; u < PBA_NUD; ++u) {
                const bool on = u < nu;
                const int U = u0 + (on ? u : 0);
                w[u][jj / 2][jj & 1] = on ? a.w_hh_dec[k * T2V_H + U] : 0.f;
                w[8 + u][jj / 2][jj & 1] = on ? a.w_ih_dec[k * T2V_KATT + U] : 0.f;
            }
#pragma unroll
            for (int c = 0; c < PBA_NCD; ++c) {
                const bool on = c < nc;
                w[16 + c][jj / 2][jj & 1] = on ? a.w_ih_dec[k * T2V_KATT + T2V_H + c0 + (on ? c : 0)] : 0.f;
            }
        }
        if (tid == 0) flag[0] = 1;
    }
    float dcd = 0.f;
    int nap = 0;
    // cell-layout factors of the own units: nu * NB rows of 8 floats, contiguous per step (thread i < 2 nu NB moves float4 i)
    const int ncp4 = 2 * nu * NB;
    const float* cpd0 = a.CPD + (size_t)u0 * NB * 8;
    constexpr size_t CPSTEP = (size_t)T2V_H * NB * 8;
    {
        const int tid = threadIdx.x;
        if (tid < ncp4) *(float4*)(cpd + ((T - 1) & 1) * 512 + 4 * tid) = *(const float4*)(cpd0 + (size_t)(T - 1) * CPSTEP + 4 * tid);
        const int cu = tid >> 3, cb = tid & 7;
        if (tid < 64 && cu < nu && cb < B) dhcs[((T - 1) & 1) * 64 + tid] = a.dHC[((size_t)(T - 1) * B + cb) * (T2V_H + T2V_E) + u0 + cu];
    }
    pb_park_factors<NB>(lds, a.FD + (size_t)(T - 1) * (PB_ROW_BYTES(NB) / 4));
    __syncthreads();

    if (a.prof && jd == 0 && threadIdx.x == 0) a.prof[40] = __builtin_amdgcn_s_memrealtime();
    // iteration t: (t < T) gather (dc, dh)(t) and build dgd(t); the RECURRENT columns of Wcat_dec^T dgd(t) first — they feed cell
    // D(t-1), whose (dc, dh) row is this role's own chain — then, behind the hand-off, the 12 columns of E(t) for the other role
    // (round 4; before, all 20 columns and the in-loop RNG / tanh of the cell sat on the chain: 9.9 -> 8.5 us per step)
    for (int t = T; t >= 0; --t) {
        int tid_op = threadIdx.x;
        asm volatile("" : "+v"(tid_op));       // (see the attention_rnn role: nothing thread-derived is hoisted out of the loop)
        const int tid = tid_op;
        const int cu = tid >> 3, cb = tid & 7;
        const bool cell_thr = tid < 64 && cu < nu && cb < B;
        const int U = u0 + (cu < nu ? cu : 0);
        if (t < T) {
            const int rounds = pb_build_row<NB>(X0, X1, rD, (unsigned)t * PB_DROW_BYTES(NB), B, nap, a.err, flag);
            nap = t2v_adapt_nap(nap, rounds);
            pba_rounds16<20, 0, 8, NB>(w, X0, X1, part);
        }
        // prefetch for the NEXT cell (step t-2): its factors and its dHC words, straight into LDS
        if (t >= 2) {
            if (tid < ncp4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cpd0 + (size_t)(t - 2) * CPSTEP + 4 * tid),
                                                 (__attribute__((address_space(3))) void*)(cpd + ((t - 2) & 1) * 512 + 256 * (tid >> 6)), 16, 0, 0);
            if (cell_thr) pb_dma4(a.dHC + ((size_t)(t - 2) * B + cb) * (T2V_H + T2V_E) + U, dhcs + ((t - 2) & 1) * 64, 0);
        }
        __syncthreads();
        if (flag[0] != 1) return;
        f32x4 dg4 = {0.f, 0.f, 0.f, 0.f};
        if (t >= 1) {
            const int td = t - 1;
            if (cell_thr) {
                const float dh = dhcs[(td & 1) * 64 + tid] + (t < T ? pb_sum16(part + (cu >> 1) * 512, (cu & 1) * 8 + cb) : 0.f);
                const float* cp = cpd + (td & 1) * 512 + (cu * NB + cb) * 8;
                const float4 c0v = *(const float4*)cp, c1v = *(const float4*)(cp + 4);
                const float dht = dh * c0v.x;                         // fh
                const float dct = dcd * c0v.y + dht * c0v.z;          // fc, go (1 - tanh(c)^2)
                dg4 = f32x4{dct * c1v.x, dct * c1v.y, dct * c1v.z, dht * c1v.w};
                dcd = dct * c0v.w;                                    // gf
                float* sp = stage + (cu * 2) * 8 + cb;
                sp[0] = dct; sp[8] = dht;
            }
            // (stage is written and read by wave 0 only: LDS operations of one wave complete in order)
            if (tid < 64) pba_publish_rows<NB>(rD, (unsigned)td * PB_DROW_BYTES(NB), stage, u0, nu);
            if (cell_thr) {       // the saved gate gradients leave after the hand-off
                const unsigned o = (unsigned)((td * B + cb) * T2V_G + U) * 4u;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dg4[0]), rDG, (int)o, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dg4[1]), rDG, (int)(o + 4u * T2V_H), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dg4[2]), rDG, (int)(o + 8u * T2V_H), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dg4[3]), rDG, (int)(o + 12u * T2V_H), 0, 0);
            }
        }
        if (t < T) {
            // ---- off the chain: E(t) = Wcat_dec[:, :1536]^T dgd(t), the decoder_rnn contribution to d h_att(t) / d ctx(t)
            pba_rounds16<20, 8, 12, NB>(w, X0, X1, part + 4 * 512);
            __syncthreads();
            if (t > 0) pb_park_factors<NB>(lds, a.FD + (size_t)(t - 1) * (PB_ROW_BYTES(NB) / 4));
            // E(t): [h_att columns | ctx columns] of this workgroup, plain (T,B,1536) layout
            if (tid >= 64 && tid < 64 + 96) {
                const int i = tid - 64, col = i >> 3, b = i & 7;      // col 0..7: unit, 8..11: ctx column
                if (b < B) {
                    const float val = pb_sum16(part + (4 + (col >> 1)) * 512, (col & 1) * 8 + b);
                    if (col < 8) { if (col < nu) pb_st4(rE, (unsigned)(((t * B + b) * T2V_KATT) + u0 + col) * 4u, val); }
                    else if (col - 8 < nc) pb_st4(rE, (unsigned)(((t * B + b) * T2V_KATT) + T2V_H + c0 + col - 8) * 4u, val);
                }
            }
        }
        __syncthreads();
    }
    if (a.prof && jd == 0 && threadIdx.x == 0) a.prof[41] = __builtin_amdgcn_s_memrealtime();
}

// the activation-only part of attention_rnn's cell backward at step t -> cpre[row][8] = {fh, fc, go(1-tanh(c)^2), gf, e0..e3}.
// Two halves: the loads are issued BEFORE the factor-row DMA and consumed after it — behind the DMA they queued for the
// whole copy on the wave's in-order memory counter and in the CU's load pipe (1.7 us for six L2-warm loads).
struct PBACellIn { float gi, gf, gg, go, cac, cprev; };
__device__ __forceinline__ PBACellIn pba_cell_pre_load(const PBAArgs& a, bool cell_thr, int t, int cb, int U) {
    PBACellIn r = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (!cell_thr) return r;
    const int B = a.B;
    const float* gp = a.GA + ((size_t)t * B + cb) * T2V_G + U;
    r.gi = gp[0]; r.gf = gp[T2V_H]; r.gg = gp[2 * T2V_H]; r.go = gp[3 * T2V_H];
    r.cac = a.CA[((size_t)(t + 1) * B + cb) * T2V_H + U];
    r.cprev = a.CA[((size_t)t * B + cb) * T2V_H + U];
    return r;
}
__device__ __forceinline__ void pba_cell_pre_finish(const PBAArgs& a, float* cpre, uint64_t seed, bool cell_thr, int rowi, int t, uint32_t idx,
                                                    const PBACellIn& r) {
    if (!cell_thr) return;
    const float gi = r.gi, gf = r.gf, gg = r.gg, go = r.go;
    float cprev = r.cprev;
    const float fh = t2v_drop_scale(seed, T2V_RNG_ATT_H, t, idx, a.p_att);
    const float fc = t2v_drop_scale(seed, T2V_RNG_ATT_C, t, idx, a.p_att);
    if (t > 0) cprev *= t2v_drop_scale(seed, T2V_RNG_ATT_C, t - 1, idx, a.p_att);
    const float tc = tanhf_(r.cac);
    *(float4*)(cpre + rowi * 8) = make_float4(fh, fc, go * (1.0f - tc * tc), gf);
    *(float4*)(cpre + rowi * 8 + 4) = make_float4(gg * gi * (1.0f - gi), cprev * gf * (1.0f - gf), gi * (1.0f - gg * gg), tc * go * (1.0f - go));
}
__device__ __forceinline__ void pba_cell_pre(const PBAArgs& a, float* cpre, uint64_t seed, bool cell_thr, int rowi, int t, int cb, int U,
                                             uint32_t idx) {
    const PBACellIn r = pba_cell_pre_load(a, cell_thr, t, cb, U);
    pba_cell_pre_finish(a, cpre, seed, cell_thr, rowi, t, idx, r);
}

// ------------------------------------------------------------------------------------------------ A role (the chain)
template <int NB, int NUA, int NCA>
__device__ __forceinline__ void pba_attention_rnn_role(const PBAArgs& a, float* lds, const int ja, const int NA) {
    constexpr int NCT = NUA + NCA;
    static_assert(NCA > 4 && NCA <= 8 && NUA > 8 && NUA <= 16, "two context rounds, three or four recurrent rounds");
    constexpr int NCR = (NCA + 1) / 2;                        // reduction rounds (2 columns each) of the context columns
    const uint64_t seed = t2v_step_seed(a.seed, a.step);
    const int tid = threadIdx.x;
    const int B = a.B, T = a.T, S = a.S_sl;
    f32x4* X0 = (f32x4*)lds;
    pb_f32x2* X1 = (pb_f32x2*)(lds + 4 * T2V_G);
    float* part = lds + (NB > 4 ? 6 : 4) * T2V_G;          // [6 groups][32 partials][32]
    float* ysum = part + 6 * 1024;                         // [21 cols -> 24][8]
    float* dhA = ysum + 192;                               // [14 units -> 16][8]  E_h(t) + ya_h(t+1)
    float* stage = dhA + 128;                              // [14 -> 16 units][4][8]
    float* wqs = stage + 512;                              // [14 -> 16][128] W_q^T rows of the own units
    float* dqs = wqs + 16 * T2V_A;                         // [8][128] dq(t) per item
    float* cpre = dqs + 8 * T2V_A;                         // [2][128 rows][8] activation-only factors of cell steps t, t-1
    float* eps = cpre + 2048;                              // [256] E(t) words / dHC words of the P2 threads (thread-private)
    int* flag = (int*)(eps + 256);                         // [4] + the phase profile (16 x 8 bytes)
    float* dump = eps + 256 + 64;                          // [8 waves][64] landing zone of the prefetch DMAs (never read)
    const int u0 = (ja * T2V_H) / NA, nu = ((ja + 1) * T2V_H) / NA - u0;      // <= 14 units
    const int c0 = (ja * T2V_E) / NA, nc = ((ja + 1) * T2V_E) / NA - c0;      // <= 7 context columns
    const __amdgpu_buffer_rsrc_t rA = pb_rsrc(a.GXA), rC = pb_rsrc(a.CX), rE = pb_rsrc(a.EX);
#if PBA_DQ_DIRECT
    const __amdgpu_buffer_rsrc_t rQ = pb_rsrc(a.DQX);
#else
    const __amdgpu_buffer_rsrc_t rQT = pb_rsrc(a.DQT);
#endif
    const __amdgpu_buffer_rsrc_t rDC = pb_rsrc(a.DCTX), rDG = pb_rsrc(a.DGA);
    // columns: [0, NUA) W_hh_att[k][U], [NUA, NUA + NCA) W_ih_att[k][256 + C]
    pb_f32x2 w[NCT][PB_KJ / 2];
#pragma unroll
    for (int jj = 0; jj < PB_KJ; ++jj) {
        const size_t k = (size_t)(tid + PB_THREADS * jj);
#pragma unroll
        for (int u = 0; u < NUA; ++u) {
            const bool on = u < nu;
            w[u][jj / 2][jj & 1] = on ? a.w_hh_att[k * T2V_H + u0 + (on ? u : 0)] : 0.f;
        }
#pragma unroll
        for (int c = 0; c < NCA; ++c) {
            const bool on = c < nc;
            w[NUA + c][jj / 2][jj & 1] = on ? a.w_ih_att[k * (T2V_PRE + T2V_E) + T2V_PRE + c0 + (on ? c : 0)] : 0.f;
        }
    }
    for (int i = tid; i < 16 * T2V_A; i += PB_THREADS) {
        const int u = i >> 7, d = i & 127;
        wqs[i] = u < nu ? a.wq[(size_t)d * T2V_H + u0 + u] : 0.f;
    }
    for (int i = tid; i < 192; i += PB_THREADS) ysum[i] = 0.f;
    if (tid == 0) flag[0] = 1;
    float dca = 0.f;
    int napA = 0, napQ = 0;
    PBA_PROF_INIT(flag);
    if (a.prof && ja == 0 && tid == 0) a.prof[42] = __builtin_amdgcn_s_memrealtime();
    // cell-layout factors of the own units (k_pb_cellpre): nu * NB rows of 8 floats, contiguous per step -> thread i < 2 nu NB
    // moves float4 number i into cpre (row = i >> 1)
    const int ncp4 = 2 * nu * NB;
    const float* cpa0 = a.CPA + (size_t)u0 * NB * 8;
    constexpr size_t CPSTEP = (size_t)T2V_H * NB * 8;
    if (tid < ncp4) *(float4*)(cpre + ((T - 1) & 1) * 1024 + 4 * tid) = *(const float4*)(cpa0 + (size_t)(T - 1) * CPSTEP + 4 * tid);
    __syncthreads();
    unsigned long long tprev_ = __builtin_readcyclecounter();

    for (int t = T - 1; t >= 0; --t) {
        // Everything derived from the thread index is recomputed per step from an OPAQUE copy: hoisted out of the loop, the two
        // dozen offsets / addresses / predicates of this body live next to 144 weight registers for the whole pass, get
        // spilled, and every reload costs an s_waitcnt vmcnt(0) — a drain of this wave's whole memory queue — on the chain.
        int tid_op = threadIdx.x;
        asm volatile("" : "+v"(tid_op));
        const int tid = tid_op;
        // cell rows: row = tid >> 2 = u * NB + b (4 lanes per row, 32 attention dims each); lane 0 of a row owns (unit u, item b)
        const int rowi = tid >> 2, cu = rowi / NB, cb = rowi - cu * NB;
        const bool row_on = cu < nu;
        const bool cell_thr = (tid & 3) == 0 && cu < nu && cb < B;
        const int U = u0 + (cu < nu ? cu : 0);
        PBA_STAMP(ja == 0, 0);
        PBA_RT(0);
        // decoder_rnn's contribution E(t) was published long ago (that role runs ahead): fetched before the chain needs it
        // (LDS-DMA: no destination registers next to the weight registers; eps[tid] / eps[192 + tid] of the lanes that own a word)
        unsigned e_off = 0u;
        if (tid < 192) {
            const int wv = tid >> 6;
            if (tid < 56) {
                const int c = tid >> 3, b = tid & 7;
                if (c < nc && b < B) {
                    e_off = (unsigned)(((t * B + b) * T2V_KATT) + T2V_H + c0 + c) * 4u;
                    pb_dma4(a.EX + (e_off >> 2), eps + 64 * wv, 1);
                    pb_dma4(a.dHC + ((size_t)t * B + b) * (T2V_H + T2V_E) + T2V_H + c0 + c, eps + 192, 0);
                }
            } else if (tid >= 64 && tid < 64 + 112) {
                const int i = tid - 64, u = i >> 3, b = i & 7;
                if (u < nu && b < B) {
                    e_off = (unsigned)(((t * B + b) * T2V_KATT) + u0 + u) * 4u;
                    pb_dma4(a.EX + (e_off >> 2), eps + 64 * wv, 1);
                }
            }
        }
        // ---- P1a: the CONTEXT columns of ya = Wcat_att^T dga(t+1) first — they are what the attention workgroups wait for
        const bool have = t < T - 1;
        if (have) {
            const int rounds = pb_build_row<NB>(X0, X1, rA, (unsigned)(t + 1) * PB_DROW_BYTES(NB), B, napA, a.err, flag);
            napA = t2v_adapt_nap(napA, rounds);
            PBA_STAMP(ja == 0, 1);
            PBA_RT(1);
        }
        // (the poll has drained the memory queue: E / dHC are in LDS and cost no wait on the vector-memory counter later)
        if (have) {
            pba_rounds16<NCT, NUA, NCA, NB>(w, X0, X1, part);
            __syncthreads();
        }
        if (!have) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // first step: nothing else has waited for the E / dHC copies
        PBA_STAMP(ja == 0, 2);
        // ---- P2: context gradient of the own columns -> attention workgroups (E(t) comes from the decoder_rnn workgroups)
        if (tid < 64) {
            const int c = tid >> 3, b = tid & 7;
            float val = 0.f;
            const bool on = c < nc && b < B;
            if (on) {
                float e = eps[tid];
                if (__float_as_uint(e) == PB_SENT) e = pba_wait_word(rE, e_off, a.err, flag);
                val = (e + eps[192 + tid]) + (have ? pb_sum16(part + (tid >> 4) * 512, tid & 15) : 0.f);        // column c: round c / 2
            }
            // lane b collects the columns of item b (lanes b + 8 c) and publishes them as one 16-byte store + the rest
            float g[NCA];
#pragma unroll
            for (int c2 = 0; c2 < NCA; ++c2) g[c2] = __shfl(val, (tid & 7) + 8 * c2, 64);
            if (tid < B) {
                const unsigned o = (unsigned)t * PB_CX_ROW_BYTES(NB) + (unsigned)tid * 4096u + 32u * (unsigned)ja;
                f32x4 hi = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c2 = 4; c2 < NCA; ++c2) hi[c2 - 4] = g[c2];          // (columns past nc carry 0: val = 0 there)
                pb_st16(rC, o, f32x4{g[0], g[1], g[2], g[3]});
                pb_st16(rC, o + 16u, hi);
            }
            // (the saved copy for the d_memory GEMM goes out AFTER the hand-off, addressed off a scalar base: nothing the
            // attention workgroups wait for may sit behind a wait on this wave's memory counter)
            if (on) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val), rDC, (int)((unsigned)((t * B + b) * T2V_E + c0 + c) * 4u), 0, 0);
        }
        PBA_STAMP(ja == 0, 3);
        PBA_RT(2);
        // ---- P1b: the recurrent columns (d h_att partial for the cell) while the attention workgroups work on step t
        if (have) {
            pba_rounds16<NCT, 0, NUA, NB>(w, X0, X1, part + 512 * NCR);
            __syncthreads();
        }
        if (tid >= 64 && tid < 64 + 112) {
            const int i = tid - 64, u = i >> 3, b = i & 7;
            if (u < nu && b < B) {
                float e = eps[tid];
                if (__float_as_uint(e) == PB_SENT) e = pba_wait_word(rE, e_off, a.err, flag);
                dhA[i] = e + (have ? pb_sum16(part + 512 * NCR + (i >> 4) * 512, i & 15) : 0.f);                 // unit u: round u / 2
            }
        }
        PBA_STAMP(ja == 0, 5);
        PBA_RT(3);
        // (issued HERE, not right after the context hand-off: 768 line fetches from HBM in this CU's memory pipe in front of
        // the publishing store delayed its landing by more than a microsecond)
        // warm this XCD's L2 with the factor row that is parked one step from now (first touch comes from HBM): one word per
        // 128-byte line, consumed only at the end of the step
        // (LDS-DMA into a dump area: the touched words are never read, so they need no register either)
        {
            float* dmp = dump + 64 * (tid >> 6);
            if (tid >= 192 && t >= 2) {
                const float* fr = a.FA + (size_t)(t - 1) * (PB_ROW_BYTES(NB) / 4);
                const int i = tid - 192;
                constexpr int NLINE = PB_ROW_BYTES(NB) / 128;
                pb_dma4(fr + 32 * i, dmp, 0);
                if (i + 320 < NLINE) pb_dma4(fr + 32 * (i + 320), dmp, 0);
                if (i + 640 < NLINE) pb_dma4(fr + 32 * (i + 640), dmp, 0);
            } else if (tid >= 64 && tid < 64 + 24 && t >= 2) {
                // ... and with the cell-layout factor lines of step t-2 (copied into cpre at the end of the next step)
                const int i = tid - 64;
                if (32 * i < 4 * ncp4) pb_dma4(cpa0 + (size_t)(t - 2) * CPSTEP + 32 * i, dmp, 0);
            } else if (tid >= 100 && tid < 100 + B && t >= 1) {
                // ... and with the lines of dHC(t-1) the loop top of the next step reads (cold HBM otherwise, and the row poll
                // right behind them waits for every older load of its wave)
                const float* q = a.dHC + ((size_t)(t - 1) * B + (tid - 100)) * (T2V_H + T2V_E) + T2V_H + c0;
                pb_dma4(q, dmp, 0);
                pb_dma4(q + nc - 1, dmp, 0);
            }
        }

        // while dq(t) is on its way, prepare the next step: the factors of the next gather (row t) into the operand slots
        // (the recurrent GEMV was their last reader), and the part of cell A(t-1) that does not depend on d h_att
        if (t > 0) {
            // cell-layout factors of step t-1 straight into cpre (16 bytes per lane), issued BEFORE the big copy: small loads
            // behind it would queue for the whole copy on the wave's in-order memory counter
            if (tid < ncp4)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cpa0 + (size_t)(t - 1) * CPSTEP + 4 * tid),
                                                 (__attribute__((address_space(3))) void*)(cpre + ((t - 1) & 1) * 1024 + 256 * (tid >> 6)), 16, 0, 0);
            pb_park_factors<NB>(lds, a.FA + (size_t)t * (PB_ROW_BYTES(NB) / 4));
            PBA_STAMP(ja == 0, 12);
            PBA_STAMP(ja == 0, 13);
        }
        // ---- P4: dq(t) of every item (sum of the position slices' partial rows)
#if PBA_DQ_DIRECT
        // (measurement variant: the attention_rnn workgroups sum the S partial rows of every item themselves — one hop less than
        // through slice 0, but B*S*512 bytes per workgroup and step instead of B*512)
        if (tid >= 320 && tid < 320 + B * 32) {
            const int i = tid - 320, b = i >> 5, q = i & 31;
            const unsigned off = (unsigned)(((t * B + b) * S) * T2V_A + 4 * q) * 4u;
            for (int n = 0; n < napQ; n += 8) __builtin_amdgcn_s_sleep(8);
            f32x4 sum = {0.f, 0.f, 0.f, 0.f};
            int rounds = 0;
            for (;;) {
                constexpr int SMAX = PB_MAXT / 16;
                f32x4 x[SMAX];
#pragma unroll
                for (int k = 0; k < SMAX; ++k)
                    if (k < S) x[k] = pb_ld16(rQ, off + (unsigned)(k * T2V_A) * 4u);
                bool ok = true;
                sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < SMAX; ++k)
                    if (k < S) { ok = ok && pb_ok(x[k][0]) && pb_ok(x[k][1]) && pb_ok(x[k][2]) && pb_ok(x[k][3]); sum += x[k]; }
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++rounds > PB_SPIN || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    flag[0] = 0;
                    break;
                }
            }
            napQ = t2v_adapt_nap(napQ, rounds);
            *(f32x4*)(dqs + b * T2V_A + 4 * q) = sum;
        }
#else
        if (tid >= 320 && tid < 320 + B * 32) {
            const int i = tid - 320, b = i >> 5, q = i & 31;
            const unsigned off = (unsigned)((t * B + b) * T2V_A + 4 * q) * 4u;
            for (int n = 0; n < napQ; n += 8) __builtin_amdgcn_s_sleep(8);
            f32x4 sum = {0.f, 0.f, 0.f, 0.f};
            int rounds = 0;
            for (;;) {
                sum = pb_ld16(rQT, off);
                const bool ok = pb_ok(sum[0]) && pb_ok(sum[1]) && pb_ok(sum[2]) && pb_ok(sum[3]);
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++rounds > PB_SPIN || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    flag[0] = 0;
                    break;
                }
            }
            napQ = t2v_adapt_nap(napQ, rounds);
            *(f32x4*)(dqs + b * T2V_A + 4 * q) = sum;
        }
#endif
        __syncthreads();
        if (flag[0] != 1) return;
        PBA_STAMP(ja == 0, 4);
        PBA_RT(4);
        // ---- W_q^T dq for the own units: row (u, b) x 4 lanes x 32 attention dims, quad sum; P5: cell A(t)
        f32x4 dg4 = {0.f, 0.f, 0.f, 0.f};
        {
            float acc = 0.f;
            if (row_on) {
                const float* wr = wqs + cu * T2V_A + 32 * (tid & 3);
                const float* dr = dqs + cb * T2V_A + 32 * (tid & 3);
#pragma unroll
                for (int i = 0; i < 32; i += 4) {
                    const float4 w4 = *(const float4*)(wr + i), d4 = *(const float4*)(dr + i);
                    acc = fmaf(w4.x, d4.x, acc); acc = fmaf(w4.y, d4.y, acc); acc = fmaf(w4.z, d4.z, acc); acc = fmaf(w4.w, d4.w, acc);
                }
            }
            acc = T2V_DPP_ADD(acc, 0xB1);
            acc = T2V_DPP_ADD(acc, 0x4E);
            if (cell_thr) {
                const float* cp = cpre + (t & 1) * 1024 + rowi * 8;
                const float4 c0v = *(const float4*)cp, c1v = *(const float4*)(cp + 4);
                const float dh = dhA[cu * 8 + cb] + acc;
                const float dht = dh * c0v.x;                         // cfh
                const float dct = dca * c0v.y + dht * c0v.z;          // cfc, go (1 - tanh(c)^2)
                const float d0 = dct * c1v.x, d1 = dct * c1v.y, d2 = dct * c1v.z, d3 = dht * c1v.w;
                dca = dct * c0v.w;                                    // gf
                float* sp = stage + (cu * 2) * 8 + cb;
                sp[0] = dct; sp[8] = dht;
                dg4 = f32x4{d0, d1, d2, d3};
            }
        }
        __syncthreads();
        if (t > 0) pba_publish_rows<NB>(rA, (unsigned)t * PB_DROW_BYTES(NB), stage, u0, nu);
        if (cell_thr) {       // the saved gate gradients (operands of the weight-gradient GEMMs) leave after the hand-off
            const unsigned o = (unsigned)((t * B + cb) * T2V_G + U) * 4u;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dg4[0]), rDG, (int)o, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dg4[1]), rDG, (int)(o + 4u * T2V_H), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dg4[2]), rDG, (int)(o + 8u * T2V_H), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dg4[3]), rDG, (int)(o + 12u * T2V_H), 0, 0);
        }
        PBA_STAMP(ja == 0, 6);
        PBA_RT(5);
        __syncthreads();
        PBA_STAMP(ja == 0, 7);
    }
    if (a.prof && ja == 0 && tid == 0) a.prof[43] = __builtin_amdgcn_s_memrealtime();
    PBA_PROF_FLUSH(ja == 0, 0, 8);
    PBA_PROF_FLUSH(ja == 0, 12, 4);
}

// NB — 4: B <= 4, 6: B = 5, 6.  LONG — 224 < T_in <= 576: the attention role as 96-position slices on all eight waves (the form
// round 4 built for "one workgroup per item"; with S <= 6 slices an item of 555 symbols takes as many workgroups as the
// headline shape's 84 symbols in 16-position slices, so the LSTM roles keep their 4 / 5 units per workgroup).
template <int NB, bool LONG>
__global__ __launch_bounds__(PB_THREADS) void k_achain_bwd(PBAArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wg = blockIdx.x;
    const int S = a.S_sl, NT = a.B * S, NL = T2V_NWG - NT, NA = pba_na(NL), ND = NL - NA;
    // (-DPBA_ONLY=1..4 builds ONE role into the kernel: `hipcc -Rpass-analysis=kernel-resource-usage` then reports that role's
    // own register pressure — the combined kernel always shows the maximum over the roles; tools/dbg/role_regs.sh)
#if defined(PBA_ONLY) && PBA_ONLY == 1
    if (PBA_WHOLE_ITEM) pba_attention_role<96, 8>(a, lds, wg / S, wg % S, NB); else pba_attention_role<16, 4>(a, lds, wg / S, wg % S, NB);
#elif defined(PBA_ONLY) && PBA_ONLY == 2
    pba_attention_rnn_role<NB, 12, 6>(a, lds, wg - NT, NA);
#elif defined(PBA_ONLY) && PBA_ONLY == 3
    pba_attention_rnn_role<NB, PBA_NUA, PBA_NCA>(a, lds, wg - NT, NA);
#elif defined(PBA_ONLY) && PBA_ONLY == 4
    pba_decoder_role<NB>(a, lds, wg - NT - NA, ND);
#else
    if (wg < NT) {
        if (LONG || (PBA_WHOLE_ITEM && a.T_in <= 96)) pba_attention_role<96, 8>(a, lds, wg / S, wg % S, NB);
        else if (a.T_in <= 128) pba_attention_role<16, 4>(a, lds, wg / S, wg % S, NB);
        else pba_attention_role<32, 4>(a, lds, wg / S, wg % S, NB);
    } else if (wg < NT + NA) {
        if (NA >= 114) pba_attention_rnn_role<NB, 9, 5>(a, lds, wg - NT, NA);
        else if (NA >= 86) pba_attention_rnn_role<NB, 12, 6>(a, lds, wg - NT, NA);
        else pba_attention_rnn_role<NB, PBA_NUA, PBA_NCA>(a, lds, wg - NT, NA);
    } else {
        pba_decoder_role<NB>(a, lds, wg - NT - NA, ND);
    }
#endif
}

// Round 4: the same roles as TWO launches on two streams — the attention chain (T + A roles, NT + NA workgroups) and the
// free-running decoder_rnn chain (D role, ND workgroups).  Together they still fill the chip and talk through the same
// sentinel-filled arrays; but the decoder_rnn chain ends ~20 % earlier (8.5 vs 10.5 us per step), and as a launch of its own
// its END is something a stream can wait for: the two decoder_rnn weight-gradient GEMMs (which need all of DGD and nothing
// of the other chain) are queued behind it and run on the CUs it frees while the attention chain is still going.
template <int NB, bool LONG>
__global__ __launch_bounds__(PB_THREADS) void k_achain_bwd_ta(PBAArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wg = blockIdx.x;
    const int S = a.S_sl, NT = a.B * S, NL = T2V_NWG - NT, NA = pba_na(NL);
    if (wg < NT) {
        if (LONG || (PBA_WHOLE_ITEM && a.T_in <= 96)) pba_attention_role<96, 8>(a, lds, wg / S, wg % S, NB);
        else if (a.T_in <= 128) pba_attention_role<16, 4>(a, lds, wg / S, wg % S, NB);
        else pba_attention_role<32, 4>(a, lds, wg / S, wg % S, NB);
    } else {
        if (NA >= 114) pba_attention_rnn_role<NB, 9, 5>(a, lds, wg - NT, NA);
        else if (NA >= 86) pba_attention_rnn_role<NB, 12, 6>(a, lds, wg - NT, NA);
        else pba_attention_rnn_role<NB, PBA_NUA, PBA_NCA>(a, lds, wg - NT, NA);
    }
}
template <int NB>
__global__ __launch_bounds__(PB_THREADS) void k_dchain_bwd_free(PBAArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int S = a.S_sl, NT = a.B * S, NL = T2V_NWG - NT, NA = pba_na(NL), ND = NL - NA;
    pba_decoder_role<NB>(a, lds, blockIdx.x, ND);
}

// slice geometry of the one-launch reverse pass: ONE workgroup per item up to 96 symbols, else the launch-per-step geometry
// (PBA_WHOLE_ITEM: measured at B = 6, T_in = 84 — 11.0 us per reverse step against 10.4 with six 16-position slices per item:
// the hand-off through slice 0 disappears (-1.4 us), but ONE workgroup needs 3.3 us from "context gradient seen" to "dq
// published" (1.3 with slices) and 11 us for its whole loop, so it becomes the chain.  Parity-green, kept for the record.)
static inline int pba_js(int T_in) { return ((PBA_WHOLE_ITEM && T_in <= 96) || T_in > PB_MAXT) ? 96 : t2v_attn_bwd_js(T_in); }
static inline int pba_slices(int T_in) { const int js = pba_js(T_in); return (T_in + js - 1) / js; }
extern "C" int t2v_decoder_bwd_persist_slices(int T_in) { return T_in < 1 ? 0 : pba_slices(T_in); }

static size_t pba_lds_bytes(int B, int T_in) {
    const size_t lrole = (B > 4 ? 6 : 4) * T2V_G + 6 * 1024 + 192 + 128 + 512 + 16 * T2V_A + 8 * T2V_A + 2048 + 256 + 64 + 512;
    const size_t Tcap = (size_t)((T_in + 15) / 16) * 16, JS = (size_t)pba_js(T_in), NWV = JS == 96 ? 8 : 4;
    const size_t trole = 4 * Tcap + T2V_E + JS + (1 + JS / NWV) * 4 * NWV + T2V_A * (JS == 96 ? JS + 17 : JS + 1) + 64 * (JS + 1) +
                         2 * 2 * NWV * T2V_A + 40 + (NWV == 8 ? 64 * 132 : 0);
    return sizeof(float) * (lrole > trole ? lrole : trole);
}

// exchange scratch of the attention chain (floats): gate-gradient rows of both cells, context gradients, window partials.
// The dq partials (T,B,S,128) are an OUTPUT (the caller reduces them into d W_q) and are passed separately.
extern "C" long t2v_decoder_bwd_achain_scratch_floats(int B, int T_in, int T_out) {
    if (B < 1 || B > PB_MAXB || T_in < 1 || T_in > PB_MAXT_LONG || T_out < 1) return 0;
    const size_t S = (size_t)pba_slices(T_in), gpw = pba_js(T_in) + 30 <= 64 ? 64 : 128;
    const size_t cx = (size_t)T_out * (B > 4 ? 32768 : 16384) / 4;
    // (dc, dh) rows of both cells (half a gate row each) + context rows + window partials + E + the two factor arrays
    const size_t cp = 2 * (size_t)T_out * T2V_H * (B > 4 ? 6 : 4) * 8;   // cell-layout factors of both cells (k_pb_cellpre)
    const size_t dqt = (size_t)T_out * 8 * T2V_A;                         // dq(t) summed over the slices (B padded to 8: 16-byte rows)
    return (long)((size_t)T_out * pb_row_bytes(B) / 4 + cx + (size_t)T_out * B * S * 2 * gpw + (size_t)T_out * B * T2V_KATT + dqt +
                  2 * (size_t)T_out * pb_row_bytes(B) / 4 + cp);
}

// float offset, inside `scratch`, of dq(t) summed over the position slices — (T_out, B, 128), complete when the pass has ended:
// the d W_q product of the caller reads it there (the sum over the slice axis of DQP was a reduction launch behind the pass)
extern "C" long t2v_decoder_bwd_achain_dq_offset(int B, int T_in, int T_out) {
    if (!t2v_decoder_bwd_persist_supported(B, T_in) || T_out < 1) return -1;
    const int S = pba_slices(T_in);
    const size_t gpw = pba_js(T_in) + 30 <= 64 ? 64 : 128;
    const size_t rowf = pb_row_bytes(B) / 4, cxf = (size_t)(B > 4 ? 32768 : 16384) / 4;
    const size_t n_gx = (size_t)T_out * rowf / 2, n_cx = (size_t)T_out * cxf, n_gp = (size_t)T_out * B * S * 2 * gpw;
    const size_t n_ex = (size_t)T_out * B * T2V_KATT;
    return (long)(2 * n_gx + n_cx + n_gp + n_ex);
}

extern "C" int t2v_decoder_bwd_achain2(const t2v_dec_train_persist_weights* w, const float* w_unused, const t2v_dec_train_bufs* s,
                                       const float* dHC, float* DGA, float* DGD, float* DCTX, float* DV, float* DQP, float* scratch,
                                       uint32_t* err_word, int B, int T_in, int T_out, float p_att, float p_dec, uint64_t seed,
                                       void* stream_, void* stream_d_);
extern "C" int t2v_decoder_bwd_achain(const t2v_dec_train_persist_weights* w, const float* w_unused, const t2v_dec_train_bufs* s,
                                      const float* dHC, float* DGA, float* DGD, float* DCTX, float* DV, float* DQP, float* scratch,
                                      uint32_t* err_word, int B, int T_in, int T_out, float p_att, float p_dec, uint64_t seed,
                                      void* stream_) {
    return t2v_decoder_bwd_achain2(w, w_unused, s, dHC, DGA, DGD, DCTX, DV, DQP, scratch, err_word, B, T_in, T_out, p_att, p_dec, seed,
                                   stream_, nullptr);
}

// do_prepare: error word, sentinel fills, the factor arrays of both cells (functions of the forward activations alone: they may
// run long before the reverse pass, next to the Postnet).  do_run: the pass itself.
static int pba_launch(const t2v_dec_train_persist_weights* w, const t2v_dec_train_bufs* s,
                      const float* dHC, float* DGA, float* DGD, float* DCTX, float* DV, float* DQP, float* scratch,
                      uint32_t* err_word, int B, int T_in, int T_out, float p_att, float p_dec, uint64_t seed,
                      void* stream_, void* stream_d_, bool do_prepare, bool do_run) {
    hipStream_t stream = (hipStream_t)stream_, stream_d = (hipStream_t)stream_d_;
    if (!s || !DQP || !scratch || !err_word) return T2V_ERR_ARG;
    if (do_run && (!w || !dHC || !DGA || !DGD || !DCTX || !DV)) return T2V_ERR_ARG;
    if (!t2v_decoder_bwd_persist_supported(B, T_in) || T_out < 1) return T2V_ERR_ARG;
    if (do_run && (!w->w_ih_att || !w->w_hh_att || !w->w_ih_dec || !w->w_hh_dec || !w->wq || !w->wcomb || !w->v)) return T2V_ERR_ARG;
    if (!s->memory || !s->XS || !s->CA || !s->CD || !s->GA || !s->GD || !s->AL || !s->S) return T2V_ERR_ARG;
    const int S = pba_slices(T_in);
    const size_t gpw = pba_js(T_in) + 30 <= 64 ? 64 : 128;
    const size_t rowf = pb_row_bytes(B) / 4, cxf = (size_t)(B > 4 ? 32768 : 16384) / 4;
    const size_t n_gx = (size_t)T_out * rowf / 2, n_f = (size_t)T_out * rowf, n_cx = (size_t)T_out * cxf, n_dq = (size_t)T_out * B * S * 128;
    const size_t n_gp = (size_t)T_out * B * S * 2 * gpw;
    const size_t n_ex = (size_t)T_out * B * T2V_KATT, n_dqt = (size_t)T_out * 8 * T2V_A;
    if (((uintptr_t)scratch & 15) || ((uintptr_t)DQP & 15) || n_gx * 4 >= 0x7fffffffull || n_dq * 4 >= 0x7fffffffull) return T2V_ERR_ARG;
    if (pba_lds_bytes(B, T_in) > PB_LDS_MAX) return T2V_ERR_ARG;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_achain_bwd<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, PB_LDS_MAX) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_achain_bwd<6, false>, hipFuncAttributeMaxDynamicSharedMemorySize, PB_LDS_MAX) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_achain_bwd_ta<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, PB_LDS_MAX) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_achain_bwd_ta<6, false>, hipFuncAttributeMaxDynamicSharedMemorySize, PB_LDS_MAX) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_achain_bwd<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PB_LDS_MAX) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_achain_bwd<6, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PB_LDS_MAX) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_achain_bwd_ta<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PB_LDS_MAX) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_achain_bwd_ta<6, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PB_LDS_MAX) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_dchain_bwd_free<4>, hipFuncAttributeMaxDynamicSharedMemorySize, PB_LDS_MAX) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_dchain_bwd_free<6>, hipFuncAttributeMaxDynamicSharedMemorySize, PB_LDS_MAX) != hipSuccess)
            return t2v_check_launch();
        attr_set = true;
    }
    if (do_prepare) {
        (void)hipMemsetAsync(err_word, 0, sizeof(uint32_t), stream);
        k_pb_fill<<<1024, 256, 0, stream>>>((uint4*)scratch, (2 * n_gx + n_cx + n_gp + n_ex + n_dqt) / 4);
        k_pb_fill<<<256, 256, 0, stream>>>((uint4*)DQP, n_dq / 4);
    }
    PBAArgs a;
    if (do_run) {
        a.w_ih_att = w->w_ih_att; a.w_hh_att = w->w_hh_att; a.w_ih_dec = w->w_ih_dec; a.w_hh_dec = w->w_hh_dec;
        a.wq = w->wq; a.wcomb = w->wcomb; a.v = w->v;
    }
    a.memory = s->memory; a.XS = s->XS; a.CA = s->CA; a.CD = s->CD; a.GA = s->GA; a.GD = s->GD; a.AL = s->AL; a.S = s->S;
    a.dHC = dHC; a.DGA = DGA; a.DGD = DGD; a.DCTX = DCTX; a.DV = DV;
    a.GXA = scratch; a.GXD = scratch + n_gx; a.CX = scratch + 2 * n_gx; a.GPX = scratch + 2 * n_gx + n_cx; a.EX = scratch + 2 * n_gx + n_cx + n_gp; a.DQX = DQP;
    a.DQT = scratch + 2 * n_gx + n_cx + n_gp + n_ex;
    float* FA = a.DQT + n_dqt;
    float* FD = FA + n_f;
    float* CPA = FD + n_f;
    float* CPD = CPA + (size_t)T_out * T2V_H * (B > 4 ? 6 : 4) * 8;
    a.FA = FA; a.FD = FD; a.CPA = CPA; a.CPD = CPD;
    a.err = err_word;
    a.B = B; a.T_in = T_in; a.T = T_out; a.S_sl = S; a.p_att = p_att; a.p_dec = p_dec; a.seed = seed;
    a.step = t2v_step_for(stream);
    a.prof = g_t2v_prof;
    if (do_prepare) {
        const unsigned nblk = (unsigned)(((size_t)T_out * T2V_G + 255) / 256);
        k_pb_factors<<<nblk, 256, 0, stream>>>(s->GA, s->CA, FA, B, T_out, B, p_att, T2V_RNG_ATT_C, seed, a.step);
        k_pb_factors<<<nblk, 256, 0, stream>>>(s->GD, s->CD, FD, B, T_out, B, p_dec, T2V_RNG_DEC_C, seed, a.step);
        const int nbs = B > 4 ? 6 : 4;
        const unsigned ncp = (unsigned)(((size_t)T_out * T2V_H * nbs + 255) / 256);
        k_pb_cellpre<<<ncp, 256, 0, stream>>>(s->GA, s->CA, CPA, B, T_out, nbs, p_att, T2V_RNG_ATT_H, T2V_RNG_ATT_C, seed, a.step);
        k_pb_cellpre<<<ncp, 256, 0, stream>>>(s->GD, s->CD, CPD, B, T_out, nbs, p_dec, T2V_RNG_DEC_H, T2V_RNG_DEC_C, seed, a.step);
    }
    if (!do_run) return t2v_check_launch();
    const size_t lds = pba_lds_bytes(B, T_in);
    const bool lng = T_in > PB_MAXT;
    if (!stream_d || stream_d == stream) {
        if (lng) {
            if (B > 4) k_achain_bwd<6, true><<<T2V_NWG, PB_THREADS, lds, stream>>>(a);
            else k_achain_bwd<4, true><<<T2V_NWG, PB_THREADS, lds, stream>>>(a);
        } else {
            if (B > 4) k_achain_bwd<6, false><<<T2V_NWG, PB_THREADS, lds, stream>>>(a);
            else k_achain_bwd<4, false><<<T2V_NWG, PB_THREADS, lds, stream>>>(a);
        }
        return t2v_check_launch();
    }
    // two launches: stream_d joins `stream` here (fills / factor kernels above), the caller joins it back (DGD is complete when
    // stream_d is, everything else when `stream` is)
    static thread_local hipEvent_t ev = nullptr;
    if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return t2v_check_launch();
    if (hipEventRecord(ev, stream) != hipSuccess || hipStreamWaitEvent(stream_d, ev, 0) != hipSuccess) return t2v_check_launch();
    const int NT = B * S, NL = T2V_NWG - NT, NA = pba_na(NL), ND = NL - NA;
    if (B > 4) {
        k_dchain_bwd_free<6><<<ND, PB_THREADS, lds, stream_d>>>(a);
        if (lng) k_achain_bwd_ta<6, true><<<NT + NA, PB_THREADS, lds, stream>>>(a);
        else k_achain_bwd_ta<6, false><<<NT + NA, PB_THREADS, lds, stream>>>(a);
    } else {
        k_dchain_bwd_free<4><<<ND, PB_THREADS, lds, stream_d>>>(a);
        if (lng) k_achain_bwd_ta<4, true><<<NT + NA, PB_THREADS, lds, stream>>>(a);
        else k_achain_bwd_ta<4, false><<<NT + NA, PB_THREADS, lds, stream>>>(a);
    }
    return t2v_check_launch();
}

extern "C" int t2v_decoder_bwd_achain2(const t2v_dec_train_persist_weights* w, const float* w_unused, const t2v_dec_train_bufs* s,
                                       const float* dHC, float* DGA, float* DGD, float* DCTX, float* DV, float* DQP, float* scratch,
                                       uint32_t* err_word, int B, int T_in, int T_out, float p_att, float p_dec, uint64_t seed,
                                       void* stream_, void* stream_d_) {
    (void)w_unused;
    return pba_launch(w, s, dHC, DGA, DGD, DCTX, DV, DQP, scratch, err_word, B, T_in, T_out, p_att, p_dec, seed, stream_, stream_d_, true, true);
}

// Round 4: the preparation of the reverse pass on its own (everything it needs exists when the FORWARD pass has ended) ...
extern "C" int t2v_decoder_bwd_achain_prepare(const t2v_dec_train_bufs* s, float* DQP, float* scratch, uint32_t* err_word, int B, int T_in,
                                              int T_out, float p_att, float p_dec, uint64_t seed, void* stream_) {
    return pba_launch(nullptr, s, nullptr, nullptr, nullptr, nullptr, nullptr, DQP, scratch, err_word, B, T_in, T_out, p_att, p_dec, seed,
                      stream_, nullptr, true, false);
}
// ... and the pass without it (same arguments as t2v_decoder_bwd_achain2; scratch / DQP / err_word as handed to _prepare)
extern "C" int t2v_decoder_bwd_achain_prepared(const t2v_dec_train_persist_weights* w, const float* w_unused, const t2v_dec_train_bufs* s,
                                               const float* dHC, float* DGA, float* DGD, float* DCTX, float* DV, float* DQP, float* scratch,
                                               uint32_t* err_word, int B, int T_in, int T_out, float p_att, float p_dec, uint64_t seed,
                                               void* stream_, void* stream_d_) {
    (void)w_unused;
    return pba_launch(w, s, dHC, DGA, DGD, DCTX, DV, DQP, scratch, err_word, B, T_in, T_out, p_att, p_dec, seed, stream_, stream_d_, false, true);
}
