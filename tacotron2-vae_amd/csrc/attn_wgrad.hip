// Deferred weight gradients of the location layer (reference model.py:24-28 LocationLayer, 45-65 energies),
// reduced over ALL (time step, item, encoder position) triples of the decoder pass in one streaming kernel:
//   d_location_dense[d][f]   = sum_{t,b,j} dpre[t,b,j,d] * conv[t,b,f,j]                       (128 x 32)
//   d_location_conv [f][c][k] = sum_{t,b,j} dc[t,b,f,j]  * a_c[t,b, j + k - 15]                 (32 x 2 x 31)
// with a_0 = attention weights, a_1 = cumulative weights entering step t (zero outside [0, T_in)).
// Both are skinny GEMMs with a huge reduction dimension (K = T*B*T_in = 201 600 at the bench shape) — a library
// GEMM has 4096 outputs to parallelise over and runs them at ~5 TFLOP/s; here 512 workgroups each stream a
// share of the (t, b) blocks (dpre 43 KB + conv/dc 21 KB per block) through LDS into fp32 MFMA accumulators and
// write one partial tile; a second tiny kernel adds the partials in a fixed order (deterministic).
// HBM-bound: 155 MB per call at the bench shape.
#include "t2v_common.h"
#include "t2v_kernels.h"

#define AW_WGS 512
#define AW_JC 64                  // encoder positions per LDS chunk
#define AW_PS 144                 // dpre chunk row stride (128 + 16): the 4 k-rows of an A read hit disjoint banks
#define AW_CS 66                  // conv / dc chunk row stride (== 2 mod 32)
#define AW_AS 128                 // padded alignment-row chunk: 64 + 30 halo (+ slack)
#define AW_PART (T2V_A * T2V_F + T2V_F * 64)     // floats per workgroup partial: 128x32 + 32x64

struct AttnWgradArgs {
    const float* dpre;    // (T, B, T_in, 128)
    const float* conv;    // (T, B, 32, T_in)
    const float* dc;      // (T, B, 32, T_in)
    const float* al;      // (T+1, B, T_in): row t = attention weights entering step t
    const float* acum;    // (T+1, B, T_in)
    float* part;          // (AW_WGS, AW_PART)
    int B, T_in, T;
};

__global__ __launch_bounds__(256) void k_attn_wgrad_part(AttnWgradArgs a) {
    __shared__ float ps[AW_JC][AW_PS];
    __shared__ float cs[T2V_F][AW_CS];
    __shared__ float ds[T2V_F][AW_CS];
    __shared__ float as_[2][AW_AS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kq = lane >> 4, i16 = lane & 15;
    const int nblk = a.T * a.B, Tp = a.T_in;
    // wave w: d_location_dense row tiles 2w, 2w+1 (d = 32w .. 32w+31) x both f tiles; d_location_conv column
    // tile w (n = 16w .. 16w+15 of the 62 (c,k) columns) x both f tiles
    f32x4 accd[2][2], accc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        accc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < 2; ++n) accd[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // this lane's (c, k) column of the location-conv gradient: B operand = a_c[j + k - 15] = as_[c][jl + k]
    const int ncol = 16 * wave + i16;                       // 0..63 (62, 63 unused)
    const int cc = ncol < T2V_KS ? 0 : 1, ck = ncol < 2 * T2V_KS ? ncol - cc * T2V_KS : 0;
    for (int blk = blockIdx.x; blk < nblk; blk += AW_WGS) {
        const float* dp = a.dpre + (size_t)blk * Tp * T2V_A;
        const float* cv = a.conv + (size_t)blk * T2V_F * Tp;
        const float* dcp = a.dc + (size_t)blk * T2V_F * Tp;
        const float* a0 = a.al + (size_t)blk * Tp;          // (t, b) row: blk = t*B + b
        const float* a1 = a.acum + (size_t)blk * Tp;
        for (int j0 = 0; j0 < Tp; j0 += AW_JC) {
            const int nj = min(AW_JC, Tp - j0);
            __syncthreads();        // previous chunk fully consumed
            // ---- stage: dpre chunk (float4 along d), conv/dc chunks (scalar along j), alignment rows with halo
            for (int e = tid; e < AW_JC * (T2V_A / 4); e += 256) {
                const int jl = e >> 5, d4 = e & 31;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (jl < nj) v = *(const float4*)(dp + (size_t)(j0 + jl) * T2V_A + 4 * d4);
                *(float4*)&ps[jl][4 * d4] = v;
            }
            for (int e = tid; e < T2V_F * AW_JC; e += 256) {
                const int f = e >> 6, jl = e & 63;
                const bool ok = jl < nj;
                cs[f][jl] = ok ? cv[(size_t)f * Tp + j0 + jl] : 0.f;
                ds[f][jl] = ok ? dcp[(size_t)f * Tp + j0 + jl] : 0.f;
            }
            for (int e = tid; e < 2 * (AW_JC + 30); e += 256) {
                const int c = e / (AW_JC + 30), x = e - c * (AW_JC + 30);
                const int j = j0 + x - 15;
                as_[c][x] = (j >= 0 && j < Tp) ? (c ? a1 : a0)[j] : 0.f;
            }
            __syncthreads();
            // ---- K loop over the chunk's positions, 4 per MFMA
            const int nks = (nj + 3) >> 2;
            for (int s = 0; s < nks; ++s) {
                const int jl = 4 * s + kq;
                const float b0 = cs[i16][jl], b1 = cs[16 + i16][jl];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const float av = ps[jl][32 * wave + 16 * m + i16];
                    accd[m][0] = mfma16x4(av, b0, accd[m][0]);
                    accd[m][1] = mfma16x4(av, b1, accd[m][1]);
                }
                const float bc = as_[cc][jl + ck];
                accc[0] = mfma16x4(ds[i16][jl], bc, accc[0]);
                accc[1] = mfma16x4(ds[16 + i16][jl], bc, accc[1]);
            }
        }
    }
    // ---- partial tiles: D row = 4*kq + r, col = i16
    float* out = a.part + (size_t)blockIdx.x * AW_PART;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(32 * wave + 16 * m + 4 * kq + r) * T2V_F + 16 * n + i16] = accd[m][n][r];
    float* outc = out + T2V_A * T2V_F;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) outc[(16 * m + 4 * kq + r) * 64 + ncol] = accc[m][r];
}

// fixed-order sum of the AW_WGS partials; thread = one output element
__global__ void k_attn_wgrad_reduce(const float* __restrict__ part, float* __restrict__ d_dense, float* __restrict__ d_conv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= AW_PART) return;
    float s = 0.f;
    for (int w = 0; w < AW_WGS; w += 4) {
        const float v0 = part[(size_t)w * AW_PART + i], v1 = part[(size_t)(w + 1) * AW_PART + i];
        const float v2 = part[(size_t)(w + 2) * AW_PART + i], v3 = part[(size_t)(w + 3) * AW_PART + i];
        s += (v0 + v1) + (v2 + v3);
    }
    if (i < T2V_A * T2V_F) {
        d_dense[i] = s;
    } else {
        const int q = i - T2V_A * T2V_F, f = q >> 6, n = q & 63;
        if (n < 2 * T2V_KS) d_conv[f * 2 * T2V_KS + n] = s;       // (32, 2, 31): n = c*31 + k
    }
}

extern "C" int t2v_attn_wgrad(const float* dpre, const float* conv, const float* dc, const float* al,
                              const float* acum, float* part_scratch, float* d_loc_dense, float* d_loc_conv,
                              int B, int T_in, int T, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!dpre || !conv || !dc || !al || !acum || !part_scratch || !d_loc_dense || !d_loc_conv) return T2V_ERR_ARG;
    if (B < 1 || T_in < 1 || T < 1) return T2V_ERR_ARG;
    AttnWgradArgs a;
    a.dpre = dpre; a.conv = conv; a.dc = dc; a.al = al; a.acum = acum; a.part = part_scratch;
    a.B = B; a.T_in = T_in; a.T = T;
    k_attn_wgrad_part<<<AW_WGS, 256, 0, stream>>>(a);
    k_attn_wgrad_reduce<<<(AW_PART + 255) / 256, 256, 0, stream>>>(part_scratch, d_loc_dense, d_loc_conv);
    return t2v_check_launch();
}

extern "C" int t2v_attn_wgrad_scratch_floats(void) { return AW_WGS * AW_PART; }
