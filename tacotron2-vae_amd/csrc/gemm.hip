// Generic fp32 GEMM on v_mfma_f32_32x32x2_f32 for the time-batched dense layers of the decoder
// (Prenet model.py:91-102, memory_layer model.py:37/290, the hoisted attention_rnn input term,
// linear_projection + gate_layer model.py:237-243 as one 81-column tile) and their gradients:
//      C[i][j] (+)= sum_k A(i,k) * B(j,k)  (+ bias[j]) (relu / dropout epilogue optional)
// A(i,k) = A[i*sAi + k*sAk], B(j,k) = B[j*sBj + k*sBk]: any of the NT / NN / TN forms by strides.
// 64x64x32 block tile, 256 threads (2x2 waves, one 32x32 accumulator each), operands staged k-major in
// LDS, next tile's global loads in flight during the MFMAs.  Threads run along whichever index of an
// operand is contiguous in memory so the staging loads are coalesced.
#include "t2v_common.h"
#include "t2v_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define GM_BM 64
#define GM_BN 64
#define GM_BK 32

struct GemmArgs {
    const float* A; const float* B; const float* bias; float* C;
    long sAi, sAk, sBj, sBk;
    int M, N, K, ldc;
    int relu, accumulate;
    float p_drop; uint64_t seed; uint32_t rng_stream, rng_t;
    const t2v_step_params* step;
};

template <bool A_KC, bool B_KC>   // operand contiguous along k?
__global__ __launch_bounds__(256) void k_gemm_f32(GemmArgs a) {
    __shared__ float As[2][GM_BK][GM_BM + 1];
    __shared__ float Bs[2][GM_BK][GM_BN + 1];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int i0 = blockIdx.y * GM_BM, j0 = blockIdx.x * GM_BN;
    constexpr int NE = GM_BM * GM_BK / 256;   // 8
    float ra[NE], rb[NE];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int e8 = 0; e8 < NE; ++e8) {
            const int e = tid + 256 * e8;
            {
                const int kk = A_KC ? (e & (GM_BK - 1)) : (e >> 6);
                const int ii = A_KC ? (e >> 5) : (e & (GM_BM - 1));
                const int i = i0 + ii, k = k0 + kk;
                ra[e8] = (i < a.M && k < a.K) ? a.A[(long)i * a.sAi + (long)k * a.sAk] : 0.f;
            }
            {
                const int kk = B_KC ? (e & (GM_BK - 1)) : (e >> 6);
                const int jj = B_KC ? (e >> 5) : (e & (GM_BN - 1));
                const int j = j0 + jj, k = k0 + kk;
                rb[e8] = (j < a.N && k < a.K) ? a.B[(long)j * a.sBj + (long)k * a.sBk] : 0.f;
            }
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int e8 = 0; e8 < NE; ++e8) {
            const int e = tid + 256 * e8;
            if (A_KC) As[buf][e & (GM_BK - 1)][e >> 5] = ra[e8]; else As[buf][e >> 6][e & (GM_BM - 1)] = ra[e8];
            if (B_KC) Bs[buf][e & (GM_BK - 1)][e >> 5] = rb[e8]; else Bs[buf][e >> 6][e & (GM_BN - 1)] = rb[e8];
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nkt = (a.K + GM_BK - 1) / GM_BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    const int ai = 32 * wm + (lane & 31), bj = 32 * wn + (lane & 31), kh = lane >> 5;
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) load_tiles((kt + 1) * GM_BK);
#pragma unroll
        for (int s = 0; s < GM_BK / 2; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[buf][2 * s + kh][ai], Bs[buf][2 * s + kh][bj], acc, 0, 0, 0);
        if (kt + 1 < nkt) store_tiles(buf ^ 1);
        __syncthreads();
    }
    const int j = j0 + 32 * wn + (lane & 31);
    if (j < a.N) {
        const float bv = a.bias ? a.bias[j] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (i < a.M) {
                const size_t idx = (size_t)i * a.ldc + j;
                float v = acc[r] + bv;
                if (a.accumulate) v += a.C[idx];
                if (a.relu) v = fmaxf(v, 0.f);
                if (a.p_drop > 0.f) v *= t2v_drop_scale(t2v_step_seed(a.seed, a.step), a.rng_stream, a.rng_t, (uint32_t)idx, a.p_drop);
                a.C[idx] = v;
            }
        }
    }
}

// ---- bf16 variant (hparams bf16_run): operands rounded to bf16 (RNE) while staged, fp32 accumulate on
// v_mfma_f32_32x32x8_bf16, fp32 in / fp32 out.  LDS tiles are k-contiguous [row][32 (+4 pad)] so each MFMA operand
// is one ds_read_b64 (row stride 72 B: conflict-free).
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_g __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned gemm_pack_bf16x2(float lo, float hi) {
    bf16x2_g p = {(__bf16)lo, (__bf16)hi};
    return *(unsigned*)&p;
}
#define GB_RS 36

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void k_gemm_bf16(GemmArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short As[2][GM_BM][GB_RS];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][GM_BN][GB_RS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int i0 = blockIdx.y * GM_BM, j0 = blockIdx.x * GM_BN;
    // each thread stages 8 consecutive k of one row of A and of B per k-tile; lanes run along the operand's
    // memory-contiguous index
    const int a_r = A_KC ? tid >> 2 : tid & 63, a_k = A_KC ? 8 * (tid & 3) : 8 * (tid >> 6);
    const int b_r = B_KC ? tid >> 2 : tid & 63, b_k = B_KC ? 8 * (tid & 3) : 8 * (tid >> 6);
    const bool a_rok = i0 + a_r < a.M, b_rok = j0 + b_r < a.N;
    const float* a_p = a.A + (long)min(i0 + a_r, a.M - 1) * a.sAi;
    const float* b_p = a.B + (long)min(j0 + b_r, a.N - 1) * a.sBj;
    float ra[8], rb[8];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int ka = k0 + a_k + u, kb = k0 + b_k + u;
            const float va = a_p[(long)min(ka, a.K - 1) * a.sAk], vb = b_p[(long)min(kb, a.K - 1) * a.sBk];
            ra[u] = (a_rok && ka < a.K) ? va : 0.f;
            rb[u] = (b_rok && kb < a.K) ? vb : 0.f;
        }
    };
    auto store_tiles = [&](int buf) {
        uint2* pa = (uint2*)&As[buf][a_r][a_k];
        pa[0] = make_uint2(gemm_pack_bf16x2(ra[0], ra[1]), gemm_pack_bf16x2(ra[2], ra[3]));
        pa[1] = make_uint2(gemm_pack_bf16x2(ra[4], ra[5]), gemm_pack_bf16x2(ra[6], ra[7]));
        uint2* pb = (uint2*)&Bs[buf][b_r][b_k];
        pb[0] = make_uint2(gemm_pack_bf16x2(rb[0], rb[1]), gemm_pack_bf16x2(rb[2], rb[3]));
        pb[1] = make_uint2(gemm_pack_bf16x2(rb[4], rb[5]), gemm_pack_bf16x2(rb[6], rb[7]));
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nkt = (a.K + GM_BK - 1) / GM_BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    const int ai = 32 * wm + (lane & 31), bj = 32 * wn + (lane & 31), kh = lane >> 5;
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) load_tiles((kt + 1) * GM_BK);
#pragma unroll
        for (int s4 = 0; s4 < GM_BK / 8; ++s4) {
            const s16x4 av = *(const s16x4*)&As[buf][ai][8 * s4 + 4 * kh];
            const s16x4 bv = *(const s16x4*)&Bs[buf][bj][8 * s4 + 4 * kh];
            acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(av, bv, acc, 0, 0, 0);
        }
        if (kt + 1 < nkt) store_tiles(buf ^ 1);
        __syncthreads();
    }
    const int j = j0 + 32 * wn + (lane & 31);
    if (j < a.N) {
        const float bv = a.bias ? a.bias[j] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (i < a.M) {
                const size_t idx = (size_t)i * a.ldc + j;
                float v = acc[r] + bv;
                if (a.accumulate) v += a.C[idx];
                if (a.relu) v = fmaxf(v, 0.f);
                if (a.p_drop > 0.f) v *= t2v_drop_scale(t2v_step_seed(a.seed, a.step), a.rng_stream, a.rng_t, (uint32_t)idx, a.p_drop);
                a.C[idx] = v;
            }
        }
    }
}

extern "C" int t2v_gemm_bf16(const float* A, long sAi, long sAk, const float* B, long sBj, long sBk, const float* bias,
                             float* C, int ldc, int M, int N, int K, int relu, int accumulate, float p_drop,
                             uint64_t seed, uint32_t rng_stream, uint32_t rng_t, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!A || !B || !C || M < 1 || N < 1 || K < 1 || ldc < N) return T2V_ERR_ARG;
    GemmArgs a;
    a.A = A; a.B = B; a.bias = bias; a.C = C; a.sAi = sAi; a.sAk = sAk; a.sBj = sBj; a.sBk = sBk;
    a.M = M; a.N = N; a.K = K; a.ldc = ldc; a.relu = relu; a.accumulate = accumulate;
    a.p_drop = p_drop; a.seed = seed; a.rng_stream = rng_stream; a.rng_t = rng_t; a.step = g_t2v_step;
    dim3 grid((N + GM_BN - 1) / GM_BN, (M + GM_BM - 1) / GM_BM);
    const bool akc = sAk == 1, bkc = sBk == 1;
    if (akc && bkc) k_gemm_bf16<true, true><<<grid, 256, 0, stream>>>(a);
    else if (akc) k_gemm_bf16<true, false><<<grid, 256, 0, stream>>>(a);
    else if (bkc) k_gemm_bf16<false, true><<<grid, 256, 0, stream>>>(a);
    else k_gemm_bf16<false, false><<<grid, 256, 0, stream>>>(a);
    return t2v_check_launch();
}

extern "C" int t2v_gemm_f32(const float* A, long sAi, long sAk, const float* B, long sBj, long sBk, const float* bias,
                            float* C, int ldc, int M, int N, int K, int relu, int accumulate, float p_drop,
                            uint64_t seed, uint32_t rng_stream, uint32_t rng_t, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!A || !B || !C || M < 1 || N < 1 || K < 1 || ldc < N) return T2V_ERR_ARG;
    GemmArgs a;
    a.A = A; a.B = B; a.bias = bias; a.C = C; a.sAi = sAi; a.sAk = sAk; a.sBj = sBj; a.sBk = sBk;
    a.M = M; a.N = N; a.K = K; a.ldc = ldc; a.relu = relu; a.accumulate = accumulate;
    a.p_drop = p_drop; a.seed = seed; a.rng_stream = rng_stream; a.rng_t = rng_t; a.step = g_t2v_step;
    dim3 grid((N + GM_BN - 1) / GM_BN, (M + GM_BM - 1) / GM_BM);
    const bool akc = sAk == 1, bkc = sBk == 1;
    if (akc && bkc) k_gemm_f32<true, true><<<grid, 256, 0, stream>>>(a);
    else if (akc) k_gemm_f32<true, false><<<grid, 256, 0, stream>>>(a);
    else if (bkc) k_gemm_f32<false, true><<<grid, 256, 0, stream>>>(a);
    else k_gemm_f32<false, false><<<grid, 256, 0, stream>>>(a);
    return t2v_check_launch();
}
