// Reference-encoder building blocks (reference modules.py:34-85, CoordConv.py:37-74,142-161):
//   k_conv2d_s2_{fwd,dx,dw} : Conv2d 3x3, stride 2, pad 1 (direct form — the six layers total 0.28 GMAC),
//                             with the CoordConv coordinate channels generated on the fly for layer 1
//   k_gru_{fwd,bwd}         : nn.GRU(256 -> 256) recurrence over the <= 16 frames that survive 6 stride-2
//                             convs (input projections are a time-batched GEMM outside)
//   k_loss                  : Tacotron2Loss_VAE forward + gradient in one pass (loss_function.py:27-44)
// BatchNorm2d + ReLU reuse the per-channel kernels of bn_act.hip on the (B, C, H*W) view.
#include "t2v_common.h"
#include "t2v_kernels.h"
#include "t2v_coop.h"

struct Conv2dArgs {
    const float* x;      // (B, Cx, H, W); with coord != 0 the kernel sees Cx + 3 input channels (xx, yy, rr)
    const float* w;      // (Cout, Cin, 3, 3), Cin = Cx (+3)
    const float* bias;
    const float* dy;     // (B, Cout, Ho, Wo)
    float* y;            // fwd: (B, Cout, Ho, Wo); dx: (B, Cx, H, W); dw: (Cout, Cin, 3, 3)
    float* dbias;
    int B, Cx, H, W, Cout, Ho, Wo, coord;
};

// value of input channel c (incl. the generated CoordConv channels) at (h, w); 0 outside the image
__device__ __forceinline__ float refenc_in(const Conv2dArgs& a, int b, int c, int h, int w) {
    if (h < 0 || h >= a.H || w < 0 || w >= a.W) return 0.f;
    if (c < a.Cx) return a.x[(((size_t)b * a.Cx + c) * a.H + h) * a.W + w];
    // CoordConv.py:42-73: xx along H, yy along W, both in [-1,1]; rr = sqrt((xx-.5)^2 + (yy-.5)^2)
    const float xx = (float)h / (float)(a.H - 1) * 2.f - 1.f;
    const float yy = (float)w / (float)(a.W - 1) * 2.f - 1.f;
    const int k = c - a.Cx;
    if (k == 0) return xx;
    if (k == 1) return yy;
    return sqrtf((xx - 0.5f) * (xx - 0.5f) + (yy - 0.5f) * (yy - 0.5f));
}

// grid = (tiles of Ho*Wo, B*Cout): the filter of this output channel sits in LDS (broadcast reads)
__global__ __launch_bounds__(256) void k_conv2d_s2_fwd(Conv2dArgs a) {
    __shared__ float wsh[131 * 9];
    const int Cin = a.Cx + (a.coord ? 3 : 0);
    const int b = blockIdx.y / a.Cout, co = blockIdx.y % a.Cout;
    for (int i = threadIdx.x; i < Cin * 9; i += 256) wsh[i] = a.w[(size_t)co * Cin * 9 + i];
    __syncthreads();
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= a.Ho * a.Wo) return;
    const int ho = r / a.Wo, wo = r - ho * a.Wo;
    float acc = a.bias ? a.bias[co] : 0.f;
    for (int c = 0; c < Cin; ++c)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
                acc = fmaf(wsh[c * 9 + kh * 3 + kw], refenc_in(a, b, c, 2 * ho - 1 + kh, 2 * wo - 1 + kw), acc);
    a.y[((size_t)b * a.Cout + co) * a.Ho * a.Wo + r] = acc;
}

// dx[b][c][h][w] = sum_{co, kh, kw : 2ho-1+kh = h, 2wo-1+kw = w} w[co][c][kh][kw] dy[b][co][ho][wo]
// grid = (tiles of H*W, B*Cx): the 9 x Cout taps of input channel c sit in LDS
__global__ __launch_bounds__(256) void k_conv2d_s2_dx(Conv2dArgs a) {
    __shared__ float wsh[128 * 9];
    const int Cin = a.Cx + (a.coord ? 3 : 0);
    const int b = blockIdx.y / a.Cx, c = blockIdx.y % a.Cx;
    for (int i = threadIdx.x; i < a.Cout * 9; i += 256) wsh[i] = a.w[((size_t)(i / 9) * Cin + c) * 9 + (i % 9)];
    __syncthreads();
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= a.H * a.W) return;
    const int h = r / a.W, w = r - h * a.W;
    float acc = 0.f;
    for (int kh = 0; kh < 3; ++kh) {
        const int hh = h + 1 - kh;
        if (hh < 0 || (hh & 1)) continue;
        const int ho = hh >> 1;
        if (ho >= a.Ho) continue;
        for (int kw = 0; kw < 3; ++kw) {
            const int ww = w + 1 - kw;
            if (ww < 0 || (ww & 1)) continue;
            const int wo = ww >> 1;
            if (wo >= a.Wo) continue;
            const float* dyp = a.dy + ((size_t)b * a.Cout * a.Ho + ho) * a.Wo + wo;
            const int tap = kh * 3 + kw;
#pragma unroll 4
            for (int co = 0; co < a.Cout; ++co) acc = fmaf(wsh[co * 9 + tap], dyp[(size_t)co * a.Ho * a.Wo], acc);
        }
    }
    a.y[((size_t)b * a.Cx + c) * a.H * a.W + r] = acc;
}

// dw[co][c][kh][kw] = sum_{b,ho,wo} dy[b][co][ho][wo] * in[b][c][2ho-1+kh][2wo-1+kw].  One workgroup per
// (co, c, position chunk): the early layers have few (co, c) pairs and tens of thousands of positions (layer 0:
// 128 pairs x 48 000), so the position range is cut into gridDim.y chunks; chunk partials go to `part` and are
// added in a fixed order by k_conv2d_s2_dw_reduce (deterministic; a single chunk writes dw directly).
__global__ __launch_bounds__(256) void k_conv2d_s2_dw(Conv2dArgs a, float* part, int chunk) {
    __shared__ float red[4][9];
    const int Cin = a.Cx + (a.coord ? 3 : 0);
    const int co = blockIdx.x / Cin, c = blockIdx.x % Cin;
    const int tid = threadIdx.x;
    float acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.f;
    const int per = a.Ho * a.Wo;
    const int lo = blockIdx.y * chunk, hi = min(a.B * per, lo + chunk);
    for (int i = lo + tid; i < hi; i += 256) {
        const int b = i / per, r = i - b * per, ho = r / a.Wo, wo = r - ho * a.Wo;
        const float g = a.dy[((size_t)b * a.Cout + co) * per + r];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) acc[kh * 3 + kw] = fmaf(g, refenc_in(a, b, c, 2 * ho - 1 + kh, 2 * wo - 1 + kw), acc[kh * 3 + kw]);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const float v = wave_sum(acc[k]);
        if ((tid & 63) == 0) red[tid >> 6][k] = v;
    }
    __syncthreads();
    if (tid < 9) {
        float* dst = gridDim.y > 1 ? part + (size_t)blockIdx.y * a.Cout * Cin * 9 : a.y;
        dst[((size_t)co * Cin + c) * 9 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    }
}

__global__ void k_conv2d_s2_dw_reduce(const float* __restrict__ part, float* __restrict__ dw, int n, int nsplit) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // eight partials in flight (the order of the additions stays fixed): a serial loop was one memory latency per partial —
    // 24 us for the 96 partials of the first two layers
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= nsplit; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(k + u) * n + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < nsplit; ++k) s += part[(size_t)k * n + i];
    dw[i] = s;
}

// position chunks for the weight gradient: aim at >= 2048 workgroups, chunks of >= 1024 positions
static inline int conv2d_dw_splits(int pairs, int positions) {
    int ns = 1;
    while (pairs * ns < 2048 && positions / (2 * ns) >= 1024 && ns < 64) ns *= 2;
    return ns;
}
extern "C" int t2v_conv2d_s2_dw_scratch_floats(int B, int Cx, int H, int W, int Cout, int coord) {
    const int Cin = Cx + (coord ? 3 : 0), Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int ns = conv2d_dw_splits(Cout * Cin, B * Ho * Wo);
    return ns > 1 ? ns * Cout * Cin * 9 : 0;
}

extern "C" int t2v_conv2d_s2_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Cx, int H, int W,
                                 int Cout, int coord, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !w || !y || B < 1 || Cx < 1 || H < 1 || W < 1 || Cout < 1) return T2V_ERR_ARG;
    if (coord && (H < 2 || W < 2)) return T2V_ERR_ARG;
    Conv2dArgs a;
    a.x = x; a.w = w; a.bias = bias; a.dy = nullptr; a.y = y; a.dbias = nullptr;
    a.B = B; a.Cx = Cx; a.H = H; a.W = W; a.Cout = Cout; a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1; a.coord = coord;
    if (Cx + (coord ? 3 : 0) > 131) return T2V_ERR_DIMS;
    k_conv2d_s2_fwd<<<dim3((a.Ho * a.Wo + 255) / 256, B * Cout), 256, 0, stream>>>(a);
    return t2v_check_launch();
}

extern "C" int t2v_conv2d_s2_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw,
                                 float* dw_scratch, int B, int Cx, int H, int W, int Cout, int coord, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !w || !dy || !dw || B < 1) return T2V_ERR_ARG;
    Conv2dArgs a;
    a.x = x; a.w = w; a.bias = nullptr; a.dy = dy; a.dbias = nullptr;
    a.B = B; a.Cx = Cx; a.H = H; a.W = W; a.Cout = Cout; a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1; a.coord = coord;
    if (dx) {
        a.y = dx;
        if (Cout > 128) return T2V_ERR_DIMS;
        k_conv2d_s2_dx<<<dim3((H * W + 255) / 256, B * Cx), 256, 0, stream>>>(a);
    }
    a.y = dw;
    const int Cin = Cx + (coord ? 3 : 0), npos = B * a.Ho * a.Wo;
    const int ns = conv2d_dw_splits(Cout * Cin, npos);
    if (ns > 1 && !dw_scratch) return T2V_ERR_ARG;
    const int chunk = (npos + ns - 1) / ns;
    k_conv2d_s2_dw<<<dim3(Cout * Cin, ns), 256, 0, stream>>>(a, dw_scratch, chunk);
    if (ns > 1) k_conv2d_s2_dw_reduce<<<(Cout * Cin * 9 + 255) / 256, 256, 0, stream>>>(dw_scratch, dw, Cout * Cin * 9, ns);
    return t2v_check_launch();
}

// ---- the same convolutions as batched GEMMs (round 3).  The direct-form kernels above give one output to a thread and a
// (co, item) pair to a workgroup: the late layers (<= 39 output positions per item) leave most lanes idle and every thread
// walks Cin*9 taps alone — 37..83 us per launch for 12..110 MFLOP.  Here the taps are gathered once (im2col, layout
// col[b][pos][k = c*9 + kh*3 + kw], CoordConv channels included) and the three products run on the 64x64 MFMA GEMM, one
// launch each for all items (t2v_gemm_f32_batched_ex):
//   y_b  (Cout x P)  = W (Cout x K) · col_b^T           + bias per row
//   dW   (Cout x K)  = sum over (item, position chunk) of  dy_b[:, chunk] · col_b[chunk, :]      (partials + fixed-order reduce)
//   dcol_b (P x K)   = dy_b^T (P x Cout) · W             then col2im: every input pixel adds up its <= 4 taps
int t2v_gemm_f32_batched_ex(const float* A, long sAb, long sAs, long sAi, long sAk, const float* B, long sBb, long sBs, long sBj, long sBk,
                            const float* bias_row, float* C, long sCb, int ldc, int nbatch, int nsub, int M, int N, int K, hipStream_t stream);

__global__ __launch_bounds__(256) void k_im2col_s2(Conv2dArgs a, float* __restrict__ col) {
    const int Cin = a.Cx + (a.coord ? 3 : 0), K = Cin * 9, P = a.Ho * a.Wo;
    const size_t n = (size_t)a.B * P * K;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const int k = (int)(e % K);
        const size_t bp = e / K;
        const int pos = (int)(bp % P), b = (int)(bp / P);
        const int c = k / 9, t9 = k - c * 9, kh = t9 / 3, kw = t9 - kh * 3;
        const int ho = pos / a.Wo, wo = pos - ho * a.Wo;
        col[e] = refenc_in(a, b, c, 2 * ho - 1 + kh, 2 * wo - 1 + kw);
    }
}
// dx[b][c][h][w] = sum of dcol[b][(ho, wo)][c*9 + kh*3 + kw] over the taps with 2ho-1+kh = h, 2wo-1+kw = w (c < Cx only)
__global__ __launch_bounds__(256) void k_col2im_s2(Conv2dArgs a, const float* __restrict__ dcol) {
    const int Cin = a.Cx + (a.coord ? 3 : 0), K = Cin * 9, P = a.Ho * a.Wo;
    const size_t n = (size_t)a.B * a.Cx * a.H * a.W;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const int w = (int)(e % a.W);
        size_t r = e / a.W;
        const int h = (int)(r % a.H);
        r /= a.H;
        const int c = (int)(r % a.Cx), b = (int)(r / a.Cx);
        float acc = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int hh = h + 1 - kh;
            if (hh < 0 || (hh & 1) || (hh >> 1) >= a.Ho) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int ww = w + 1 - kw;
                if (ww < 0 || (ww & 1) || (ww >> 1) >= a.Wo) continue;
                acc += dcol[((size_t)b * P + (size_t)(hh >> 1) * a.Wo + (ww >> 1)) * K + c * 9 + kh * 3 + kw];
            }
        }
        a.y[e] = acc;
    }
}
// position chunks of the weight-gradient product: a divisor of P, chunks of >= 64 positions, <= 16 chunks
static inline int conv2d_gemm_chunks(int P) {
    int ns = 1;
    while (ns < 16 && P % (2 * ns) == 0 && P / (2 * ns) >= 64) ns *= 2;
    return ns;
}
// scratch of the GEMM form (floats): [col: B*P*K, reused for dcol][dW partials: B*ns*Cout*K]
extern "C" long t2v_conv2d_s2_gemm_scratch_floats(int B, int Cx, int H, int W, int Cout, int coord) {
    if (B < 1 || Cx < 1 || H < 1 || W < 1 || Cout < 1) return 0;
    const long Cin = Cx + (coord ? 3 : 0), K = Cin * 9, P = (long)((H - 1) / 2 + 1) * ((W - 1) / 2 + 1);
    return B * P * K + (long)B * conv2d_gemm_chunks((int)P) * Cout * K;
}
static Conv2dArgs conv2d_args(const float* x, const float* w, int B, int Cx, int H, int W, int Cout, int coord) {
    Conv2dArgs a;
    a.x = x; a.w = w; a.bias = nullptr; a.dy = nullptr; a.y = nullptr; a.dbias = nullptr;
    a.B = B; a.Cx = Cx; a.H = H; a.W = W; a.Cout = Cout; a.Ho = (H - 1) / 2 + 1; a.Wo = (W - 1) / 2 + 1; a.coord = coord;
    return a;
}
extern "C" int t2v_conv2d_s2_fwd_gemm(const float* x, const float* w, const float* bias, float* y, float* scratch, int B, int Cx, int H,
                                      int W, int Cout, int coord, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !w || !y || !scratch || B < 1 || Cx < 1 || H < 1 || W < 1 || Cout < 1) return T2V_ERR_ARG;
    if (coord && (H < 2 || W < 2)) return T2V_ERR_ARG;
    Conv2dArgs a = conv2d_args(x, w, B, Cx, H, W, Cout, coord);
    const int K = (Cx + (coord ? 3 : 0)) * 9, P = a.Ho * a.Wo;
    const size_t n = (size_t)B * P * K;
    k_im2col_s2<<<(unsigned)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256), 256, 0, stream>>>(a, scratch);
    return t2v_gemm_f32_batched_ex(w, 0, 0, K, 1, scratch, (long)P * K, 0, K, 1, bias, y, (long)Cout * P, P, B, 1, Cout, P, K, stream);
}
// `scratch` must still hold the im2col of the forward pass (t2v_conv2d_s2_fwd_gemm leaves it there); it is overwritten.
extern "C" int t2v_conv2d_s2_bwd_gemm(const float* x, const float* w, const float* dy, float* dx, float* dw, float* scratch, int B, int Cx,
                                      int H, int W, int Cout, int coord, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!x || !w || !dy || !dw || !scratch || B < 1) return T2V_ERR_ARG;
    Conv2dArgs a = conv2d_args(x, w, B, Cx, H, W, Cout, coord);
    const int K = (Cx + (coord ? 3 : 0)) * 9, P = a.Ho * a.Wo;
    float* col = scratch;
    float* part = scratch + (size_t)B * P * K;
    const int ns = conv2d_gemm_chunks(P), chunk = P / ns;
    // dW partials: item b, chunk s: dy_b[:, chunk] (Cout x chunk) · col_b[chunk, :] (as B operand: [j = k][k' = pos])
    int rc = t2v_gemm_f32_batched_ex(dy, (long)Cout * P, chunk, P, 1, col, (long)P * K, (long)chunk * K, 1, K, nullptr, part, (long)Cout * K, K,
                                     B, ns, Cout, K, chunk, stream);
    if (rc != T2V_OK) return rc;
    k_conv2d_s2_dw_reduce<<<(Cout * K + 255) / 256, 256, 0, stream>>>(part, dw, Cout * K, B * ns);
    if (dx) {
        // dcol_b (P x K) = dy_b^T (P x Cout) · W (as B operand: [j = k][k' = co]); overwrites col
        rc = t2v_gemm_f32_batched_ex(dy, (long)Cout * P, 0, 1, P, w, 0, 0, 1, K, nullptr, col, (long)P * K, K, B, 1, P, K, Cout, stream);
        if (rc != T2V_OK) return rc;
        a.y = dx;
        const size_t n = (size_t)B * Cx * H * W;
        k_col2im_s2<<<(unsigned)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256), 256, 0, stream>>>(a, col);
    }
    return t2v_check_launch();
}

// ------------------------------------------------------------------------------------------------ GRU
// nn.GRU semantics: r,z,n gates; n = tanh(gi_n + r * (W_hn h + b_hn)); h' = (1-z) n + z h.
// gi (B,T,768) = x·W_ih^T + b_ih is computed outside (time-batched GEMM).  T = T_out/64 steps (<= 16), B <= 16.
// PERSISTENT cooperative kernels like the encoder BiLSTM: 8 workgroups, each owning 32 hidden units whose three
// gate rows of W_hh (fp32, 96 x 256) stay in VGPRs as v_mfma_f32_16x16x4_f32 A fragments for all steps; a tile
// is 4 units x 4 gate slots (slot 3 empty) so the accumulator registers of a lane are (r, z, n) of ONE unit and
// the gate math is lane-local.  Per step the workgroups exchange the new hidden state (forward) / the gate
// gradients (backward) with write-through stores + one bounded group barrier (t2v_coop.h).
#define GRU_H 256
#define GRU_G (3 * GRU_H)
#define GRU_NW 8
#define GRU_UNITS (GRU_H / GRU_NW)   // 32

struct GruArgs {
    const float* gi;      // (B,T,768)
    const float* whh;     // (768,256)
    const float* bhh;     // (768)
    float* hs;            // (B,T+1,256) hidden states, hs[:,0] = 0 written here
    float* gsave;         // (B,T,4,256): r, z, n, (W_hn h + b_hn) for the backward; or NULL
    const float* dh_last; // bwd: (B,256) gradient of the last hidden state
    float* dgi;           // bwd: (B,T,768) grad wrt gi
    float* dgh;           // bwd: (B,T,768) grad wrt (W_hh h + b_hh) rows
    float* xchg;          // fwd: (2,16,256) h exchange; bwd: (2,16,768) gate-gradient exchange (step parity)
    unsigned* sync;       // [0] arrival counter, [1] error word; zeroed by the launcher
    int B, T;
};

__global__ __launch_bounds__(256) void k_gru_fwd(GruArgs a) {
    const int j = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = lane & 15, g = lane >> 4;
    const bool bvalid = b < a.B;
    __shared__ float hbuf[16][GRU_H + 4];
    // A fragments of this wave's two tiles: row i = lane&15 -> (unit i>>2, gate slot i&3), k = 4s + g
    float wreg[2][64];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int unit = j * GRU_UNITS + (2 * wave + tt) * 4 + ((lane & 15) >> 2);
        const int slot = lane & 3;
#pragma unroll
        for (int s = 0; s < 64; ++s)
            wreg[tt][s] = slot < 3 ? a.whh[(size_t)(slot * GRU_H + unit) * GRU_H + 4 * s + g] : 0.f;
    }
    float bh[2][3];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int U = j * GRU_UNITS + (2 * wave + tt) * 4 + g;
#pragma unroll
        for (int r = 0; r < 3; ++r) bh[tt][r] = a.bhh[r * GRU_H + U];
        if (bvalid) a.hs[((size_t)b * (a.T + 1)) * GRU_H + U] = 0.f;
    }
    for (int i = tid; i < 16 * (GRU_H + 4); i += 256) (&hbuf[0][0])[i] = 0.f;
    __syncthreads();

    for (int t = 0; t < a.T; ++t) {
        float giv[2][3];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int U = j * GRU_UNITS + (2 * wave + tt) * 4 + g;
#pragma unroll
            for (int r = 0; r < 3; ++r) giv[tt][r] = bvalid ? a.gi[((size_t)b * a.T + t) * GRU_G + r * GRU_H + U] : 0.f;
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const float* hrow = &hbuf[b][g];
#pragma unroll
        for (int s = 0; s < 64; ++s) {
            const float hv = hrow[4 * s];
            acc0 = mfma16x4(wreg[0][s], hv, acc0);
            acc1 = mfma16x4(wreg[1][s], hv, acc1);
        }
        float* hx_w = a.xchg + (size_t)(t & 1) * 16 * GRU_H;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const f32x4 acc = tt == 0 ? acc0 : acc1;
            const int U = j * GRU_UNITS + (2 * wave + tt) * 4 + g;
            if (bvalid) {
                const float r = sigmoidf_(giv[tt][0] + acc[0] + bh[tt][0]);
                const float z = sigmoidf_(giv[tt][1] + acc[1] + bh[tt][1]);
                const float hn = acc[2] + bh[tt][2];
                const float n = tanhf_(giv[tt][2] + r * hn);
                const float hnew = (1.f - z) * n + z * hbuf[b][U];
                if (a.gsave) {
                    float* sv = a.gsave + (((size_t)b * a.T + t) * 4) * GRU_H + U;
                    sv[0] = r; sv[GRU_H] = z; sv[2 * GRU_H] = n; sv[3 * GRU_H] = hn;
                }
                a.hs[((size_t)b * (a.T + 1) + t + 1) * GRU_H + U] = hnew;
                st_sc1(hx_w + (size_t)b * GRU_H + U, hnew);
            }
        }
        if (t + 1 == a.T) break;
        if (!group_barrier(a.sync, (unsigned)(GRU_NW * (t + 1)), a.sync + 1)) return;
        {   // the new hidden state of every item: all loads of a thread in flight at once (round 6 — as a plain loop over the items
            // every iteration waited for its own load: one memory round trip per item and step; rows past B re-read row B - 1)
            float hv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) hv[u] = ld_sc1(hx_w + (size_t)min(u, a.B - 1) * GRU_H + tid);
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (u < a.B) hbuf[u][tid] = hv[u];
        }
        __syncthreads();
    }
}

// BPTT.  Thread (item bb = tid>>5 (+8), unit uu = tid&31) does the gate gradients of its (item, unit); the
// transposed recurrent product dh_prev[b][unit] = sum_rows W_hh[row][unit] * dg[b][row] (768 rows split over the
// four waves) runs on MFMA with W_hh^T fragments resident in VGPRs.
__global__ __launch_bounds__(256) void k_gru_bwd(GruArgs a) {
    const int j = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = lane & 15, g = lane >> 4;
    __shared__ float dgbuf[16][GRU_G + 4];
    __shared__ f32x4 red[4][2][64];
    __shared__ float dhrec[16][GRU_UNITS + 1];
    float wreg[2][48];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int col = j * GRU_UNITS + tt * 16 + (lane & 15);
#pragma unroll
        for (int s = 0; s < 48; ++s) wreg[tt][s] = a.whh[(size_t)(192 * wave + 4 * s + g) * GRU_H + col];
    }
    const int uu = tid & 31, U = j * GRU_UNITS + uu;
    for (int i = tid; i < 16 * GRU_UNITS; i += 256) {
        const int bb = i >> 5, u2 = i & 31;
        dhrec[bb][u2] = bb < a.B ? a.dh_last[(size_t)bb * GRU_H + j * GRU_UNITS + u2] : 0.f;
    }
    __syncthreads();

    for (int t = a.T - 1; t >= 0; --t) {
        float* dgx_w = a.xchg + (size_t)(t & 1) * 16 * GRU_G;
        float keep[2] = {0.f, 0.f};
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            const int bb = (tid >> 5) + 8 * rep;
            if (bb < a.B) {
                const float* sv = a.gsave + (((size_t)bb * a.T + t) * 4) * GRU_H + U;
                const float r = sv[0], z = sv[GRU_H], n = sv[2 * GRU_H], hn = sv[3 * GRU_H];
                const float hprev = a.hs[((size_t)bb * (a.T + 1) + t) * GRU_H + U];
                const float d = dhrec[bb][uu];
                const float dn = d * (1.f - z) * (1.f - n * n);
                const float dz = d * (hprev - n) * z * (1.f - z);
                const float dr = dn * hn * r * (1.f - r);
                float* gi = a.dgi + ((size_t)bb * a.T + t) * GRU_G + U;
                gi[0] = dr; gi[GRU_H] = dz; gi[2 * GRU_H] = dn;
                float* gh = a.dgh + ((size_t)bb * a.T + t) * GRU_G + U;
                gh[0] = dr; gh[GRU_H] = dz; gh[2 * GRU_H] = dn * r;
                float* x = dgx_w + (size_t)bb * GRU_G + U;
                st_sc1(x, dr); st_sc1(x + GRU_H, dz); st_sc1(x + 2 * GRU_H, dn * r);
                keep[rep] = d * z;                  // direct path h' = ... + z h
            }
        }
        if (t == 0) break;
        if (!group_barrier(a.sync, (unsigned)(GRU_NW * (a.T - t)), a.sync + 1)) return;
        // the gate gradients of every item (768 per item, three per thread): the 16 loads of a third in flight at once (round 6: the
        // plain loop over B * 768 / 256 values made one memory round trip per value — 18 per step at B = 6)
#pragma unroll
        for (int q3 = 0; q3 < 3; ++q3) {
            float dv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) dv[u] = ld_sc1(dgx_w + (size_t)min(u, a.B - 1) * GRU_G + 256 * q3 + tid);
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (u < a.B) dgbuf[u][256 * q3 + tid] = dv[u];
        }
        __syncthreads();
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const float* drow = &dgbuf[b][192 * wave + g];
#pragma unroll
        for (int s = 0; s < 48; ++s) {
            const float dv = drow[4 * s];
            acc0 = mfma16x4(wreg[0][s], dv, acc0);
            acc1 = mfma16x4(wreg[1][s], dv, acc1);
        }
        red[wave][0][lane] = acc0;
        red[wave][1][lane] = acc1;
        __syncthreads();
        if (wave < 2) {     // wave tt finalises tile tt: lane (col = item b, rows 4g+r = units 16tt+4g+r)
            const f32x4 s4 = (red[0][wave][lane] + red[1][wave][lane]) + (red[2][wave][lane] + red[3][wave][lane]);
            if (b < a.B) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dhrec[b][16 * wave + 4 * g + r] = s4[r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            const int bb = (tid >> 5) + 8 * rep;
            if (bb < a.B) dhrec[bb][uu] += keep[rep];
        }
        __syncthreads();
    }
}

extern "C" int t2v_gru_fwd(const float* gi, const float* whh, const float* bhh, float* hs, float* gsave,
                           float* xchg, uint32_t* sync2, int B, int T, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!gi || !whh || !bhh || !hs || !xchg || !sync2 || B < 1 || B > 16 || T < 1) return T2V_ERR_ARG;
    (void)hipMemsetAsync(sync2, 0, 2 * sizeof(uint32_t), stream);
    GruArgs a;
    a.gi = gi; a.whh = whh; a.bhh = bhh; a.hs = hs; a.gsave = gsave; a.dh_last = nullptr; a.dgi = nullptr; a.dgh = nullptr;
    a.xchg = xchg; a.sync = sync2; a.B = B; a.T = T;
    k_gru_fwd<<<GRU_NW, 256, 0, stream>>>(a);
    return t2v_check_launch();
}

extern "C" int t2v_gru_bwd(const float* whh, const float* hs, const float* gsave, const float* dh_last, float* dgi,
                           float* dgh, float* xchg, uint32_t* sync2, int B, int T, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!whh || !hs || !gsave || !dh_last || !dgi || !dgh || !xchg || !sync2 || B < 1 || B > 16 || T < 1)
        return T2V_ERR_ARG;
    (void)hipMemsetAsync(sync2, 0, 2 * sizeof(uint32_t), stream);
    GruArgs a;
    a.gi = nullptr; a.whh = whh; a.bhh = nullptr; a.hs = (float*)hs; a.gsave = (float*)gsave; a.dh_last = dh_last;
    a.dgi = dgi; a.dgh = dgh; a.xchg = xchg; a.sync = sync2; a.B = B; a.T = T;
    k_gru_bwd<<<GRU_NW, 256, 0, stream>>>(a);
    return t2v_check_launch();
}

// ------------------------------------------------------------------------------------------------ loss
// total = MSE(mel) + MSE(post) + BCEWithLogits(gate) + w * KL,  KL = -0.5 * sum(1 + logvar - mu^2 - e^logvar)
// Writes out[0..3] = total, recon, kl, (unused) and the gradients of `total`, in one launch of one
// workgroup-per-slab grid followed by a fixed-order final reduction (deterministic).
struct LossArgs {
    const float* mel; const float* post; const float* mel_t; const float* gate; const float* gate_t;
    const float* mu; const float* logvar;
    float* dmel; float* dpost; float* dgate; float* dmu; float* dlogvar;
    float* part;     // (nblk, 3)
    float* out;      // (4)
    size_t n_mel; int n_gate, n_lat, nblk;
    float klw;
    const t2v_step_params* step;   // device-side KL weight (graph replay) or NULL
    unsigned* ticket;
};

__global__ __launch_bounds__(256) void k_loss(LossArgs a) {
    __shared__ float scr[4];
    __shared__ int last;
    const int tid = threadIdx.x;
    const float klw = a.step ? a.step->kl_weight : a.klw;
    float s_mel = 0.f, s_gate = 0.f, s_kl = 0.f;
    const float cm = 2.0f / (float)a.n_mel;
    for (size_t i = (size_t)blockIdx.x * 256 + tid; i < a.n_mel; i += (size_t)gridDim.x * 256) {
        const float t = a.mel_t[i];
        const float d0 = a.mel[i] - t, d1 = a.post[i] - t;
        s_mel = fmaf(d0, d0, s_mel);
        s_mel = fmaf(d1, d1, s_mel);
        a.dmel[i] = cm * d0;
        a.dpost[i] = cm * d1;
    }
    for (int i = blockIdx.x * 256 + tid; i < a.n_gate; i += gridDim.x * 256) {
        const float x = a.gate[i], y = a.gate_t[i];
        s_gate += fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
        a.dgate[i] = (1.0f / (1.0f + expf(-x)) - y) / (float)a.n_gate;
    }
    for (int i = blockIdx.x * 256 + tid; i < a.n_lat; i += gridDim.x * 256) {
        const float m = a.mu[i], lv = a.logvar[i], e = expf(lv);
        s_kl += -0.5f * (1.f + lv - m * m - e);
        a.dmu[i] = klw * m;
        a.dlogvar[i] = klw * -0.5f * (1.f - e);
    }
    const float v[3] = {s_mel, s_gate, s_kl};
    for (int k = 0; k < 3; ++k) {
        float x = wave_sum(v[k]);
        __syncthreads();
        if ((tid & 63) == 0) scr[tid >> 6] = x;
        __syncthreads();
        if (tid == 0) a.part[(size_t)blockIdx.x * 3 + k] = (scr[0] + scr[1]) + (scr[2] + scr[3]);
    }
    // last block to finish reduces the partials in index order
    __threadfence();
    if (tid == 0) last = (atomicAdd(a.ticket, 1u) == gridDim.x - 1);
    __syncthreads();
    if (last && tid == 0) {
        __threadfence();
        double m = 0.0, g = 0.0, k = 0.0;
        for (int i = 0; i < a.nblk; ++i) {
            m += (double)__hip_atomic_load(a.part + (size_t)i * 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            g += (double)__hip_atomic_load(a.part + (size_t)i * 3 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            k += (double)__hip_atomic_load(a.part + (size_t)i * 3 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const float recon = (float)(m / (double)a.n_mel + g / (double)a.n_gate);
        a.out[1] = recon;
        a.out[2] = (float)k;
        a.out[0] = recon + klw * (float)k;
        a.out[3] = klw;
        *a.ticket = 0;
    }
}

extern "C" int t2v_loss_fwd_bwd(const float* mel, const float* post, const float* mel_t, const float* gate,
                                const float* gate_t, const float* mu, const float* logvar, float* dmel, float* dpost,
                                float* dgate, float* dmu, float* dlogvar, float* part192, float* out4, uint32_t* ticket,
                                uint64_t n_mel, int n_gate, int n_lat, float kl_weight, void* stream_) {
    if (!mel || !post || !mel_t || !gate || !gate_t || !mu || !logvar || !dmel || !dpost || !dgate || !dmu || !dlogvar ||
        !part192 || !out4 || !ticket)
        return T2V_ERR_ARG;
    LossArgs a;
    a.mel = mel; a.post = post; a.mel_t = mel_t; a.gate = gate; a.gate_t = gate_t; a.mu = mu; a.logvar = logvar;
    a.dmel = dmel; a.dpost = dpost; a.dgate = dgate; a.dmu = dmu; a.dlogvar = dlogvar; a.part = part192; a.out = out4;
    a.n_mel = n_mel; a.n_gate = n_gate; a.n_lat = n_lat; a.nblk = 64; a.klw = kl_weight; a.ticket = ticket; a.step = t2v_step_for((hipStream_t)stream_);
    k_loss<<<64, 256, 0, (hipStream_t)stream_>>>(a);
    return t2v_check_launch();
}
