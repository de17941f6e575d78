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

#define PB_THREADS 512
#define PB_MAXB 6
#define PB_MAXT 224
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
            nap = rounds > 1 ? nap + 12 * (rounds - 1) : (rounds == 0 ? max(0, nap - 6) : nap);
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
    if (!(B >= 1 && B <= PB_MAXB && T_in >= 1 && T_in <= PB_MAXT)) return 0;
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
