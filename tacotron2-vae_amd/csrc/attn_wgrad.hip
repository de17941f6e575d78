// Deferred weight gradients of the location layer (reference model.py:24-28 LocationLayer, 45-65 energies).
// The forward evaluates LocationLayer through the fused filter bank W_comb[d][c,k] = sum_f dense[d][f] conv[f][c][k]
// (attn_fwd.hip), so the chain rule gives both weight gradients from ONE reduction over ALL (time step, item,
// encoder position) triples of the decoder pass:
//   dW_comb[d][c,k]          = sum_{t,b,j} dpre[t,b,j,d] * a_c[t,b, j + k - 15]                  (128 x 62)
//   d_location_dense[d][f]   = sum_{c,k} dW_comb[d][c,k] * conv[f][c][k]
//   d_location_conv [f][c][k] = sum_d    dense[d][f]     * dW_comb[d][c,k]
// with a_0 = attention weights, a_1 = cumulative weights entering step t (zero outside [0, T_in)).
// dW_comb is a skinny GEMM with a huge reduction dimension (K = T*B*T_in = 201 600 at the bench shape): 512
// workgroups each stream a share of the (t, b) blocks (dpre 43 KB + two alignment rows per block) through LDS into
// fp32 MFMA accumulators and write one partial 128 x 64 tile; a second kernel adds the partials in a fixed order
// (deterministic), a third applies the two small contractions.  HBM-bound: 103 MB of dpre per call at the bench shape
// (the conv outputs and their gradients are no longer stored or streamed).
#include "t2v_common.h"
#include "t2v_kernels.h"

#define AW_WGS 512
#define AW_JC 64                  // encoder positions per LDS chunk
#define AW_PS 144                 // dpre chunk row stride (128 + 16): the 4 k-rows of an A read hit disjoint banks
#define AW_AS 128                 // padded alignment-row chunk: 64 + 30 halo (+ slack)
#define AW_PART (T2V_A * 64)      // floats per workgroup partial: 128 x 64 (columns 31, 63 unused)

struct AttnWgradArgs {
    const float* dpre;    // (T, B, T_in, 128)
    const float* al;      // (T+1, B, T_in): row t = attention weights entering step t
    const float* acum;    // (T+1, B, T_in)
    float* part;          // (AW_WGS, AW_PART)
    int B, T_in, T;
};

__global__ __launch_bounds__(256) void k_attn_wgrad_part(AttnWgradArgs a) {
    __shared__ float ps[AW_JC][AW_PS];
    __shared__ float as_[2][AW_AS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kq = lane >> 4, i16 = lane & 15;
    const int nblk = a.T * a.B, Tp = a.T_in;
    // wave w: rows d = 32w .. 32w+31 (two m-tiles) x all four (c,k) column tiles
    f32x4 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int blk = blockIdx.x; blk < nblk; blk += AW_WGS) {
        const float* dp = a.dpre + (size_t)blk * Tp * T2V_A;
        const float* a0 = a.al + (size_t)blk * Tp;          // (t, b) row: blk = t*B + b
        const float* a1 = a.acum + (size_t)blk * Tp;
        for (int j0 = 0; j0 < Tp; j0 += AW_JC) {
            const int nj = min(AW_JC, Tp - j0);
            __syncthreads();        // previous chunk fully consumed
            // ---- stage: dpre chunk (float4 along d), alignment rows with halo
            for (int e = tid; e < AW_JC * (T2V_A / 4); e += 256) {
                const int jl = e >> 5, d4 = e & 31;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (jl < nj) v = *(const float4*)(dp + (size_t)(j0 + jl) * T2V_A + 4 * d4);
                *(float4*)&ps[jl][4 * d4] = v;
            }
            for (int e = tid; e < 2 * AW_AS; e += 256) {
                const int c = e / AW_AS, x = e - c * AW_AS;
                const int j = j0 + x - 15;
                as_[c][x] = (x < AW_JC + 31 && j >= 0 && j < Tp) ? (c ? a1 : a0)[j] : 0.f;
            }
            __syncthreads();
            // ---- K loop over the chunk's positions, 4 per MFMA: B[kq][n = (c,k)] = a_c[j + k - 15] = as_[c][jl + k]
            const int nks = (nj + 3) >> 2;
            for (int s = 0; s < nks; ++s) {
                const int jl = 4 * s + kq;
                const float av0 = ps[jl][32 * wave + i16], av1 = ps[jl][32 * wave + 16 + i16];
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const float bv = as_[n >> 1][jl + 16 * (n & 1) + i16];      // column 16n + i16 -> c = n>>1, k = 16(n&1) + i16
                    acc[0][n] = mfma16x4(av0, bv, acc[0][n]);
                    acc[1][n] = mfma16x4(av1, bv, acc[1][n]);
                }
            }
        }
    }
    // ---- partial tile: D row = 4*kq + r, col = i16
    float* out = a.part + (size_t)blockIdx.x * AW_PART;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(32 * wave + 16 * m + 4 * kq + r) * 64 + 16 * n + i16] = acc[m][n][r];
}

// fixed-order sum of the AW_WGS partials; thread = one element of dW_comb (128 x 64)
__global__ void k_attn_wgrad_reduce(const float* __restrict__ part, float* __restrict__ dwc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= AW_PART) return;
    float s = 0.f;
    for (int w = 0; w < AW_WGS; w += 4) {
        const float v0 = part[(size_t)w * AW_PART + i], v1 = part[(size_t)(w + 1) * AW_PART + i];
        const float v2 = part[(size_t)(w + 2) * AW_PART + i], v3 = part[(size_t)(w + 3) * AW_PART + i];
        s += (v0 + v1) + (v2 + v3);
    }
    dwc[i] = s;
}

// chain rule through W_comb = dense · conv; thread = one output element (128*32 of d_dense, then 32*62 of d_conv)
__global__ void k_attn_wgrad_final(const float* __restrict__ dwc, const float* __restrict__ conv,
                                   const float* __restrict__ dense, float* __restrict__ d_dense,
                                   float* __restrict__ d_conv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < T2V_A * T2V_F) {
        const int d = i / T2V_F, f = i - d * T2V_F;
        float s = 0.f;
        for (int c = 0; c < 2; ++c)
            for (int k = 0; k < T2V_KS; ++k) s = fmaf(dwc[d * 64 + 32 * c + k], conv[(f * 2 + c) * T2V_KS + k], s);
        d_dense[i] = s;
    } else if (i < T2V_A * T2V_F + T2V_F * 2 * T2V_KS) {
        const int q = i - T2V_A * T2V_F, f = q / (2 * T2V_KS), n = q - f * 2 * T2V_KS;
        const int c = n / T2V_KS, k = n - c * T2V_KS;
        float s = 0.f;
        for (int d = 0; d < T2V_A; ++d) s = fmaf(dense[d * T2V_F + f], dwc[d * 64 + 32 * c + k], s);
        d_conv[q] = s;
    }
}

extern "C" int t2v_attn_wgrad(const float* dpre, const float* al, const float* acum, const float* loc_conv,
                              const float* loc_dense, float* part_scratch, float* d_loc_dense, float* d_loc_conv,
                              int B, int T_in, int T, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!dpre || !al || !acum || !loc_conv || !loc_dense || !part_scratch || !d_loc_dense || !d_loc_conv) return T2V_ERR_ARG;
    if (B < 1 || T_in < 1 || T < 1) return T2V_ERR_ARG;
    AttnWgradArgs a;
    a.dpre = dpre; a.al = al; a.acum = acum; a.part = part_scratch;
    a.B = B; a.T_in = T_in; a.T = T;
    float* dwc = part_scratch + (size_t)AW_WGS * AW_PART;
    k_attn_wgrad_part<<<AW_WGS, 256, 0, stream>>>(a);
    k_attn_wgrad_reduce<<<(AW_PART + 255) / 256, 256, 0, stream>>>(part_scratch, dwc);
    const int nout = T2V_A * T2V_F + T2V_F * 2 * T2V_KS;
    k_attn_wgrad_final<<<(nout + 255) / 256, 256, 0, stream>>>(dwc, loc_conv, loc_dense, d_loc_dense, d_loc_conv);
    return t2v_check_launch();
}

extern "C" int t2v_attn_wgrad_scratch_floats(void) { return (AW_WGS + 1) * AW_PART; }
