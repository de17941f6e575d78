// Conv1d (stride 1, odd kernel, "same" zero padding) as implicit GEMMs on fp32 MFMA, for the
// encoder conv bank (model.py:159-177) and the Postnet (model.py:110-148), forward and backward:
//   forward / data-gradient :  Y[b][m][t] = sum_{c,k} W[m][c][k] * X[b][c][t + k - P]      (+bias[m])
//                              M = Cout, N = B*T, K = Cin*KS      (dX uses the flipped, transposed weight)
//   weight gradient         :  dW[m][c][k] = sum_{b,t} dY[b][m][t] * X[b][c][t + k - P]
//                              M = Cout, N = Cin*KS, K = B*T
// Block tile 64x64x16, 256 threads = 2x2 waves, each wave one 32x32 accumulator driven by
// v_mfma_f32_32x32x2_f32; operands are staged through LDS k-major ([k][m] / [k][n], +1 pad) so the
// per-lane MFMA operand reads are conflict-free, and the next tile's global loads (im2col gather
// done on the fly, coalesced along t) are in flight while the current tile is multiplied.
// The forward epilogue also emits per-channel partial sums / sums of squares of the conv output
// (BatchNorm training statistics, biased variance over B*T incl. padded frames — Appendix B-3)
// so BN needs no extra pass over Y.
#include "t2v_common.h"
#include "t2v_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CG_BM 64
#define CG_BN 64
#define CG_BK 32

struct ConvGemmArgs {
    const float* W;      // forward: (M, Cin*KS) row-major
    const float* X;      // (B, Cin, T)
    const float* dY;     // weight-gradient mode: (B, M, T)
    const float* bias;   // (M) or NULL
    float* Y;            // forward: (B, M, T); weight-gradient: (M, Cin*KS)
    float* stat_part;    // forward: (gridDim.x, M, 2) partial [sum, sumsq] or NULL
    int B, Cin, T, M, KS;
};

// MODE 0: forward / data gradient.  MODE 1: weight gradient.
template <int MODE, int KS>
__global__ __launch_bounds__(256) void k_conv_gemm(ConvGemmArgs a) {
    __shared__ float As[2][CG_BK][CG_BM + 1];
    __shared__ float Bs[2][CG_BK][CG_BN + 1];
    __shared__ float red[2][CG_BM][2];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * CG_BM, n0 = blockIdx.x * CG_BN;
    constexpr int P = KS >> 1;
    const int BT = a.B * a.T, CK = a.Cin * KS;
    const int Kdim = MODE == 0 ? CK : BT;
    const int Ndim = MODE == 0 ? BT : CK;

    // each thread stages 8 A elements and 8 B elements per k-tile (BK = 32); everything that does not
    // depend on the k-tile (row / column decomposition) is computed once
    constexpr int NE = CG_BM * CG_BK / 256;     // 8
    float ra[NE], rb[NE];
    // MODE 0:  A: thread -> (mm = e>>5, kk = e&31);  B: thread -> (kb = e>>6, nn = e&63)
    // MODE 1:  A: (mm = e>>5, kk = e&31) with k=(b,t);  B: (nn = e>>5, kb = e&31)
    int b_bb = 0, b_t = 0;          // MODE 0: this thread's fixed output column n -> (bb, t)
    bool b_nok = false;
    if (MODE == 0) {
        const int n = n0 + (tid & (CG_BN - 1));
        b_nok = n < Ndim;
        b_bb = b_nok ? n / a.T : 0;
        b_t = n - b_bb * a.T;
    }
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + 256 * i;
            if (MODE == 0) {
                const int kk = e & (CG_BK - 1), mm = e >> 5;
                const int m = m0 + mm, k = k0 + kk;
                ra[i] = (m < a.M && k < Kdim) ? a.W[(size_t)m * CK + k] : 0.f;
                const int k2 = k0 + (e >> 6);
                float v = 0.f;
                if (b_nok && k2 < Kdim) {
                    const int c = k2 / KS, kx = k2 - c * KS;
                    const int ts = b_t + kx - P;
                    if (ts >= 0 && ts < a.T) v = a.X[((size_t)b_bb * a.Cin + c) * a.T + ts];
                }
                rb[i] = v;
            } else {
                const int kk = e & (CG_BK - 1), mm = e >> 5;
                const int m = m0 + mm, k = k0 + kk;
                const int bb = k / a.T, t = k - bb * a.T;
                ra[i] = (m < a.M && k < Kdim) ? a.dY[((size_t)bb * a.M + m) * a.T + t] : 0.f;
                const int n = n0 + mm;           // same (kk, mm) decomposition for the B tile
                float w = 0.f;
                if (n < Ndim && k < Kdim) {
                    const int c = n / KS, kx = n - c * KS;
                    const int ts = t + kx - P;
                    if (ts >= 0 && ts < a.T) w = a.X[((size_t)bb * a.Cin + c) * a.T + ts];
                }
                rb[i] = w;
            }
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + 256 * i;
            As[buf][e & (CG_BK - 1)][e >> 5] = ra[i];
            if (MODE == 0) Bs[buf][e >> 6][e & (CG_BN - 1)] = rb[i];
            else Bs[buf][e & (CG_BK - 1)][e >> 5] = rb[i];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int nkt = (Kdim + CG_BK - 1) / CG_BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    const int ai = 32 * wm + (lane & 31), bj = 32 * wn + (lane & 31), kh = lane >> 5;
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) load_tiles((kt + 1) * CG_BK);
#pragma unroll
        for (int s = 0; s < CG_BK / 2; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[buf][2 * s + kh][ai], Bs[buf][2 * s + kh][bj], acc, 0, 0, 0);
        if (kt + 1 < nkt) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // epilogue.  D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int n = n0 + 32 * wn + (lane & 31);
    float psum[16], psq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float v = acc[r];
        if (MODE == 0) {
            if (a.bias && m < a.M) v += a.bias[m];
            if (m < a.M && n < Ndim) {
                const int bb = n / a.T, t = n - bb * a.T;
                a.Y[((size_t)bb * a.M + m) * a.T + t] = v;
            } else {
                v = 0.f;
            }
            psum[r] = v;
            psq[r] = v * v;
        } else {
            if (m < a.M && n < Ndim) a.Y[(size_t)m * CK + n] = v;
        }
    }
    if (MODE == 0 && a.stat_part) {
        // per-channel partial statistics over this block's 64 columns: lanes (same lane>>5) share a row
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s = psum[r], q = psq[r];
            s = row16_sum(s); q = row16_sum(q);
            s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);     // 32 columns of this wave
            if ((lane & 31) == 0) {
                const int ml = 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                red[wn][ml][0] = s;
                red[wn][ml][1] = q;
            }
        }
        __syncthreads();
        if (tid < CG_BM) {
            const int m = m0 + tid;
            if (m < a.M) {
                float* dst = a.stat_part + ((size_t)blockIdx.x * a.M + m) * 2;
                dst[0] = red[0][tid][0] + red[1][tid][0];
                dst[1] = red[0][tid][1] + red[1][tid][1];
            }
        }
    }
}

// W (M, Cin, KS) -> Wt (Cin, M, KS) with the taps flipped: conv(dY, Wt) is the data gradient
__global__ void k_conv_flip_weight(const float* __restrict__ W, float* __restrict__ Wt, int M, int Cin, int KS) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * Cin * KS) return;
    const int k = i % KS, c = (i / KS) % Cin, m = i / (KS * Cin);
    Wt[((size_t)c * M + m) * KS + (KS - 1 - k)] = W[i];
}

extern "C" int t2v_conv1d_fwd(const float* W, const float* X, const float* bias, float* Y, float* stat_part,
                              int B, int Cin, int T, int Cout, int KS, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!W || !X || !Y || B < 1 || Cin < 1 || T < 1 || Cout < 1 || KS < 1 || !(KS & 1)) return T2V_ERR_ARG;
    ConvGemmArgs a;
    a.W = W; a.X = X; a.dY = nullptr; a.bias = bias; a.Y = Y; a.stat_part = stat_part;
    a.B = B; a.Cin = Cin; a.T = T; a.M = Cout; a.KS = KS;
    dim3 grid((B * T + CG_BN - 1) / CG_BN, (Cout + CG_BM - 1) / CG_BM);
    if (KS == 5) k_conv_gemm<0, 5><<<grid, 256, 0, stream>>>(a);
    else if (KS == 3) k_conv_gemm<0, 3><<<grid, 256, 0, stream>>>(a);
    else return T2V_ERR_DIMS;
    return t2v_check_launch();
}

extern "C" int t2v_conv1d_stat_blocks(int B, int T) { return (B * T + CG_BN - 1) / CG_BN; }

extern "C" int t2v_conv1d_bwd(const float* W, const float* X, const float* dY, float* dX, float* dW, float* Wt_scratch,
                              int B, int Cin, int T, int Cout, int KS, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!W || !X || !dY || B < 1 || Cin < 1 || T < 1 || Cout < 1 || KS < 1 || !(KS & 1)) return T2V_ERR_ARG;
    if (dX) {
        if (!Wt_scratch) return T2V_ERR_ARG;
        const int n = Cout * Cin * KS;
        k_conv_flip_weight<<<(n + 255) / 256, 256, 0, stream>>>(W, Wt_scratch, Cout, Cin, KS);
        ConvGemmArgs a;
        a.W = Wt_scratch; a.X = dY; a.dY = nullptr; a.bias = nullptr; a.Y = dX; a.stat_part = nullptr;
        a.B = B; a.Cin = Cout; a.T = T; a.M = Cin; a.KS = KS;
        dim3 grid((B * T + CG_BN - 1) / CG_BN, (Cin + CG_BM - 1) / CG_BM);
        if (KS == 5) k_conv_gemm<0, 5><<<grid, 256, 0, stream>>>(a);
        else if (KS == 3) k_conv_gemm<0, 3><<<grid, 256, 0, stream>>>(a);
        else return T2V_ERR_DIMS;
    }
    if (dW) {
        ConvGemmArgs a;
        a.W = nullptr; a.X = X; a.dY = dY; a.bias = nullptr; a.Y = dW; a.stat_part = nullptr;
        a.B = B; a.Cin = Cin; a.T = T; a.M = Cout; a.KS = KS;
        dim3 grid((Cin * KS + CG_BN - 1) / CG_BN, (Cout + CG_BM - 1) / CG_BM);
        if (KS == 5) k_conv_gemm<1, 5><<<grid, 256, 0, stream>>>(a);
        else if (KS == 3) k_conv_gemm<1, 3><<<grid, 256, 0, stream>>>(a);
        else return T2V_ERR_DIMS;
    }
    return t2v_check_launch();
}
