// Shared device helpers for libt2vae_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Model geometry of the decoder path at the reference's default hparams
// (reference hparams.py:81-101).  The C API rejects anything else.
#define T2V_H 1024      // attention_rnn_dim == decoder_rnn_dim
#define T2V_E 512       // encoder_embedding_dim
#define T2V_PRE 256     // prenet_dim
#define T2V_A 128       // attention_dim
#define T2V_F 32        // attention_location_n_filters
#define T2V_KS 31       // attention_location_kernel_size
#define T2V_NMEL 80
#define T2V_G (4 * T2V_H)                 // 4096 gate rows
#define T2V_XW (T2V_H + T2V_E + T2V_H)    // 2560: [h_att | ctx | h_dec]
#define T2V_KATT (T2V_H + T2V_E)          // 1536: recurrent K of the attention LSTM
#define T2V_KATT_INF (T2V_KATT + T2V_PRE) // 1792: + prenet columns (inference)
#define T2V_NWG 256                       // one workgroup per CU

// v_mfma_f32_16x16x4_f32: D(16x16) += A(16x4) * B(4x16); lane l holds A[l&15][l>>4],
// B[l>>4][l&15]; D: col = l&15, row = 4*(l>>4)+r.
__device__ __forceinline__ f32x4 mfma16x4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// bf16 weight streams of the decoder LSTM kernels (hparams bf16_run): a lane's four consecutive k of a 16x16x16 tile.
// v_mfma_f32_16x16x16_bf16: lane l holds A[l&15][4(l>>4)+i], B[4(l>>4)+i][l&15], i = 0..3 — the same (row, k) ownership as
// four steps of the fp32 16x16x4 tile, so the bf16 packs are the fp32 packs rounded element by element (RNE).
typedef short t2v_s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 t2v_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned t2v_pack_bf16x2(float lo, float hi) {
    t2v_bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return *(unsigned*)&v;
}
__device__ __forceinline__ uint2 t2v_pack_bf16x4(float4 v) { return make_uint2(t2v_pack_bf16x2(v.x, v.y), t2v_pack_bf16x2(v.z, v.w)); }
__device__ __forceinline__ f32x4 mfma16x16_bf16(uint2 a, uint2 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(*(const t2v_s16x4*)&a, *(const t2v_s16x4*)&b, c, 0, 0, 0);
}

// one k-block of 16 of a 16x16 tile: four fp32 MFMA steps, or one bf16 MFMA on the rounded operands
__device__ __forceinline__ void mfma_block(f32x4& acc, const float4& wv, const float4& xv) {
    acc = mfma16x4(wv.x, xv.x, acc); acc = mfma16x4(wv.y, xv.y, acc);
    acc = mfma16x4(wv.z, xv.z, acc); acc = mfma16x4(wv.w, xv.w, acc);
}
// 16x16x32 bf16 tile step: lane l holds A[l&15][8(l>>4)+i], B[8(l>>4)+i][l&15], i = 0..7 (one uint4 each)
typedef __bf16 t2v_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mfma16x32_bf16(uint4 a, float4 x0, float4 x1, f32x4 c) {
    const uint4 b = make_uint4(t2v_pack_bf16x2(x0.x, x0.y), t2v_pack_bf16x2(x0.z, x0.w), t2v_pack_bf16x2(x1.x, x1.y), t2v_pack_bf16x2(x1.z, x1.w));
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const t2v_bf16x8*)&a, *(const t2v_bf16x8*)&b, c, 0, 0, 0);
}
__device__ __forceinline__ uint4 t2v_pack_bf16x8(float4 a, float4 b) {
    return make_uint4(t2v_pack_bf16x2(a.x, a.y), t2v_pack_bf16x2(a.z, a.w), t2v_pack_bf16x2(b.x, b.y), t2v_pack_bf16x2(b.z, b.w));
}
__device__ __forceinline__ void mfma_block(f32x4& acc, const uint2& wv, const float4& xv) {
    acc = mfma16x16_bf16(wv, t2v_pack_bf16x4(xv), acc);
}

// exp via v_exp_f32 (2^x, ~1 ulp) and v_rcp_f32: absolute error of sigmoid/tanh ~1e-7, which is
// what the fp32 parity budget (mel-L1 < 1e-4 through 400 recurrent steps) needs; far cheaper
// than the libm tanhf/expf call sequences.
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// streamed-once weights: non-temporal so the 67 MB per-step stream does not evict the small
// recurrent state / attention operands from the XCD L2s
__device__ __forceinline__ float4 ld_nt(const float4* p) {
    const f32x4 v = __builtin_nontemporal_load((const f32x4*)p);
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ float sigmoidf_(float x) { return fast_rcp(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * fast_rcp(1.0f + __expf(2.0f * x)); }

// 16-lane row sum with DPP (no LDS crossbar): every lane of a row ends with the row's sum.
#define T2V_DPP_ADD(v, CTRL) \
    ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (CTRL), 0xF, 0xF, true)))
#define T2V_DPP_MAX(v, CTRL) \
    fmaxf((v), __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (CTRL), 0xF, 0xF, true)))
__device__ __forceinline__ float row16_sum(float v) {
    v = T2V_DPP_ADD(v, 0xB1);    // quad_perm [1,0,3,2]
    v = T2V_DPP_ADD(v, 0x4E);    // quad_perm [2,3,0,1]
    v = T2V_DPP_ADD(v, 0x141);   // row_half_mirror
    v = T2V_DPP_ADD(v, 0x140);   // row_mirror
    return v;
}

// Counter-based RNG (splitmix64 finaliser) for dropout keep-masks: a pure function of
// (seed, stream, t, idx) so the backward pass regenerates the forward's masks.
__device__ __forceinline__ uint32_t t2v_rng_u32(uint64_t seed, uint32_t stream, uint32_t t, uint32_t idx) {
    uint64_t x = seed ^ ((uint64_t)stream << 58) ^ ((uint64_t)t << 32) ^ (uint64_t)idx;
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (uint32_t)(x >> 32);
}
// seed of this launch: the by-value seed, advanced by the device-side step counter when one is installed
// (t2v_set_step_params): under graph replay the argument is frozen, the counter is not
struct t2v_step_params;
__device__ __forceinline__ uint64_t t2v_step_seed(uint64_t seed, const t2v_step_params* step) {
    return step ? seed + *(const uint64_t*)step * 0x9E3779B97F4A7C15ull : seed;
}
// returns the multiplicative dropout factor: 0 or 1/(1-p)
__device__ __forceinline__ float t2v_drop_scale(uint64_t seed, uint32_t stream, uint32_t t, uint32_t idx, float p) {
    if (p <= 0.0f) return 1.0f;
    const float u = (float)(t2v_rng_u32(seed, stream, t, idx) >> 8) * (1.0f / 16777216.0f);
    return u >= p ? 1.0f / (1.0f - p) : 0.0f;
}

// Attention context partial of one thread: sum over the positions j = part, part + 8, ... < Tp of w[j] * m[j * 64 + c] (both in
// LDS), W positions per round with all 2 W reads issued before the first FMA.  As a plain loop the compiler keeps ONE pair of
// reads in flight per FMA: 107 cycles per position measured (round 6: 70 positions per thread at 555 symbols = 8 000 cycles of a
// 30 000-cycle step; 11 at the headline's 84 symbols).  No masks in the loop (seven compare results held in SGPR pairs pushed the
// forward kernel from 44 to 68 spilled SGPRs and made it slower as a whole): the CALLER keeps w[Tp .. Tp + 8 W) at zero
// (T2V_CTX_PAD extra floats behind the Tcap weights, cleared once), rows past the end re-read row Tp - 1.
#define T2V_CTX_PAD 64
template <int W>
__device__ __forceinline__ float t2v_ctx_partial(const float* w, const float* m, int part_, int c, int Tp) {
    static_assert(8 * W <= T2V_CTX_PAD, "zero padding of the weights");
    // `part` is the wave index: as a scalar the row arithmetic runs on the SALU and the reads of a full round are ONE base address
    // + immediate offsets (with per-lane index arithmetic the loop was bound by VALU issue: ~80 instructions per round of eight
    // positions on two waves per SIMD = 700 cycles, measured)
    const int part = __builtin_amdgcn_readfirstlane(part_);
    float acc[W];
#pragma unroll
    for (int k = 0; k < W; ++k) acc[k] = 0.f;
    int j0 = part;
    for (; j0 + 8 * (W - 1) < Tp; j0 += 8 * W) {          // full rounds
        const float* wp = w + j0;
        const float* mp = m + j0 * 64 + c;
        float e[W], x[W];
#pragma unroll
        for (int k = 0; k < W; ++k) {
            e[k] = wp[8 * k];
            x[k] = mp[8 * 64 * k];
        }
#pragma unroll
        for (int k = 0; k < W; ++k) acc[k] = fmaf(e[k], x[k], acc[k]);
    }
    if (j0 < Tp) {                                        // the last, partial round: zero weights past the end, clamped rows
        float e[W], x[W];
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const int j = j0 + 8 * k;
            e[k] = w[j];
            x[k] = m[min(j, Tp - 1) * 64 + c];
        }
#pragma unroll
        for (int k = 0; k < W; ++k) acc[k] = fmaf(e[k], x[k], acc[k]);
    }
#pragma unroll
    for (int s = W / 2; s > 0; s >>= 1)
#pragma unroll
        for (int k = 0; k < s; ++k) acc[k] += acc[k + s];
    return acc[0];
}

// The same sum with per-lane index arithmetic and no main / tail split: every round clamps its rows.  The decode kernel takes this
// form with W = 4 (25 positions per thread at 200 symbols: 13.70 us per frame against 13.90 with the scalar form above and 14.28
// with the plain loop, same box; W = 8: 13.9).
template <int W>
__device__ __forceinline__ float t2v_ctx_partial_v(const float* w, const float* m, int part, int c, int Tp) {
    static_assert(8 * W <= T2V_CTX_PAD, "zero padding of the weights");
    float acc[W];
#pragma unroll
    for (int k = 0; k < W; ++k) acc[k] = 0.f;
    for (int j0 = part; j0 < Tp; j0 += 8 * W) {
        float e[W], x[W];
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const int j = j0 + 8 * k;
            e[k] = w[j];
            x[k] = m[min(j, Tp - 1) * 64 + c];
        }
#pragma unroll
        for (int k = 0; k < W; ++k) acc[k] = fmaf(e[k], x[k], acc[k]);
    }
#pragma unroll
    for (int s = W / 2; s > 0; s >>= 1)
#pragma unroll
        for (int k = 0; k < s; ++k) acc[k] += acc[k + s];
    return acc[0];
}

// Cross-row steps WITHOUT the LDS crossbar (`__shfl_xor(v, 16 / 32)` is a ds_bpermute: ~100 cycles each, two in a row per wave
// reduction — on the per-frame / per-step chains of the persistent kernels that adds up, round 6).  `v` holds, in every lane of a
// 16-lane row, that row's partial (after row16_sum or a DPP row max): four v_readlane_b32 and a few scalar-operand VALU ops.
// Bit-identical to the shuffle forms: (r0 + r1) + (r2 + r3) in every lane, fp addition is commutative.
__device__ __forceinline__ float t2v_readlane_f(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float rows4_sum(float v) {
    return (t2v_readlane_f(v, 0) + t2v_readlane_f(v, 16)) + (t2v_readlane_f(v, 32) + t2v_readlane_f(v, 48));
}
__device__ __forceinline__ float rows4_max(float v) {
    return fmaxf(fmaxf(t2v_readlane_f(v, 0), t2v_readlane_f(v, 16)), fmaxf(t2v_readlane_f(v, 32), t2v_readlane_f(v, 48)));
}
// the same for the two halves of a wave separately: lanes 0..31 get r0 + r1, lanes 32..63 r2 + r3 (= v + __shfl_xor(v, 16))
__device__ __forceinline__ float rows2_sum(float v) {
    const float lo = t2v_readlane_f(v, 0) + t2v_readlane_f(v, 16), hi = t2v_readlane_f(v, 32) + t2v_readlane_f(v, 48);
    return (threadIdx.x & 32) ? hi : lo;
}
__device__ __forceinline__ float wave_sum_rl(float v) { return rows4_sum(row16_sum(v)); }
// value of lane Q of the own quad (DPP quad_perm broadcast)
#define T2V_DPP_QUAD_F(v, Q) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (Q) * 0x55, 0xF, 0xF, true))

__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Adaptive nap ahead of a polled hand-off (s_sleep units of 64 clocks): `rounds` = failed polls of the last wait.  The nap
// grows while polls fail, shrinks by a quarter when the data was already there, and is CLAMPED — the first wait of a
// pass can be a hundred times longer than a steady-state one (weights still loading), and an unclamped additive rule
// turned that into naps of hundreds of microseconds that then decayed over hundreds of steps.
__device__ __forceinline__ int t2v_adapt_nap(int nap, int rounds) {
    if (rounds > 1) return min(192, nap + 8 * min(rounds - 1, 4));
    if (rounds == 0) return (3 * nap) >> 2;
    return nap;
}

// phase stamp for the optional in-kernel profile (one thread of one workgroup)
#define T2V_STAMP(ARGS, I)                                                                          \
    do {                                                                                            \
        if ((ARGS).prof && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0)                 \
            (ARGS).prof[(I)] = __builtin_readcyclecounter();                                        \
    } while (0)
