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
#include <stdlib.h>
#include "t2v_common.h"
#include "t2v_kernels.h"
#include "t2v_coop.h"

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

// ------------------------------------------------------------------------------------------------------------
// Tiled kernels for the k=5 convolutions with Cin % 16 == 0 (every conv of the encoder bank and the Postnet).
// No im2col gather: a workgroup stages the input rows it needs ONCE per channel block ([16 channels][BN+4
// positions], halo included) and all five taps read shifted windows of it, and the K order inside a tile is
// tap-major (k' = 16*tap + channel) so that every MFMA operand address is  lane_base + compile-time constant.
// Output tile 64 x (16*NTW); 4 waves, wave w owns rows 16w..16w+15 and NTW 16x16 accumulators
// (v_mfma_f32_16x16x4_f32: D row = 4*(lane>>4)+r, col = lane&15).  K tile = 80 (= 16 channels x 5 taps, or 80/96
// positions for the weight gradient): 20-24 k-steps x NTW MFMAs between barriers, the next tile's global loads
// in flight meanwhile.  Tiles never straddle utterances (zero padding at utterance edges comes from the halo).
#define CT_BM 64
#define CT_AS 80                 // As row stride: 64 + 16 -> the four k-rows of an A read land in disjoint banks
#define CT_KT 80

struct ConvTiledArgs {
    const float* W;      // fwd: (M, Cin*5) row-major
    const float* X;      // (B, Cin, T)
    const float* dY;     // dW: (B, M, T)
    const float* bias;
    float* Y;            // fwd: (B, M, T);  dW: (M, Cin*5)
    float* stat_part;    // fwd: (gridDim.x, M, 2) or NULL
    int B, Cin, T, M, tiles_per_item;
    unsigned long long* prof;   // optional: [0..1] shader-clock stamps, [2..3] 100 MHz stamps of workgroup (0,0)
    float* ks_part;      // fwd, gridDim.z == 2 (input channels cut in halves): raw accumulator tiles [z][tile][256 threads][4 NTW]
    unsigned* ks_ctr;    // ... and one arrival counter per output tile (zero before and after the launch)
};

template <int NTW>
__global__ __launch_bounds__(256) void k_conv5_fwd(ConvTiledArgs a) {
    constexpr int BN = 16 * NTW;
    constexpr int XW = BN + 4;                           // staged positions per channel (2-halo each side)
    constexpr int XS = BN == 32 ? 48 : BN <= 64 ? 80 : 112;   // row stride >= BN+4, = 16 or 48 mod 64: four k-rows -> disjoint banks
    constexpr int NX = (16 * XW + 255) / 256;            // X elements staged per thread
    __shared__ float As[2][CT_KT][CT_AS];
    __shared__ float Xs[2][16][XS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kq = lane >> 4, j = lane & 15;
    const int bb = blockIdx.x / a.tiles_per_item, t0 = (blockIdx.x % a.tiles_per_item) * BN;
    const int m0 = blockIdx.y * CT_BM;
    const int CK = a.Cin * 5;
    // gridDim.z == 2 (round 4, launches that fill half the chip or less — the encoder bank's 6 x 84 positions are 144
    // workgroups): each half of the input channels in its own workgroup, the one that finishes second adds the other's tile
    const int nkt_all = a.Cin / 16;
    const int kt0 = gridDim.z > 1 ? (int)blockIdx.z * (nkt_all / 2) : 0;
    const int nkt = gridDim.z > 1 ? (blockIdx.z == 0 ? nkt_all / 2 : nkt_all - nkt_all / 2) : nkt_all;

    const bool stamp = a.prof && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;
    if (stamp) { a.prof[0] = __builtin_readcyclecounter(); a.prof[2] = wall_clock64(); }
    // ---- staging plan (everything that does not depend on the k-tile is computed once; no branches around loads:
    //      rows past M are clamped — their outputs are never stored — and out-of-range positions are selected to 0)
    float4 ra[5];
    float rx[NX];
    const int a_m = tid & 63, w4 = tid >> 6;
    const float* a_row = a.W + (size_t)min(m0 + a_m, a.M - 1) * CK + 4 * w4;     // + 80*kt + 16*i
    int a_lds[20];                      // LDS offsets of this thread's 20 A elements: (16*tap + channel)*CT_AS + row
#pragma unroll
    for (int q = 0; q < 20; ++q) {
        const int k = 4 * w4 + 16 * (q >> 2) + (q & 3);       // position in the W row segment: 5*channel + tap
        const int c = k / 5, kx = k - 5 * c;
        a_lds[q] = (16 * kx + c) * CT_AS + a_m;
    }
    const float* x_item = a.X + (size_t)bb * a.Cin * a.T;
    int x_goff[NX], x_lds[NX];
    bool x_ok[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int e = tid + 256 * i;
        const int c = min(e / XW, 15), jj = e - (e / XW) * XW;
        const int t = t0 - 2 + jj;
        x_ok[i] = e < 16 * XW && t >= 0 && t < a.T;
        x_goff[i] = c * a.T + min(max(t, 0), a.T - 1);
        x_lds[i] = e < 16 * XW ? c * XS + jj : -1;
    }
    auto load_one = [&](int kt, int q) {          // q-th global load of tile kt (q compile-time after unrolling)
        if (q < 5) {
            ra[q] = *(const float4*)(a_row + 80 * (kt0 + kt) + 16 * q);
        } else {
            const float v = x_item[(size_t)16 * (kt0 + kt) * a.T + x_goff[q - 5]];
            rx[q - 5] = x_ok[q - 5] ? v : 0.f;
        }
    };
    auto store_one = [&](int buf, int q) {        // q-th LDS store of the staged tile
        if (q < 20) {
            const float4 v4 = ra[q >> 2];
            const float v = (q & 3) == 0 ? v4.x : (q & 3) == 1 ? v4.y : (q & 3) == 2 ? v4.z : v4.w;
            (&As[buf][0][0])[a_lds[q]] = v;
        } else if (q < 20 + NX) {
            if (x_lds[q - 20] >= 0) (&Xs[buf][0][0])[x_lds[q - 20]] = rx[q - 20];
        }
    };
    constexpr int NLOAD = 5 + NX, NSTORE = 20 + NX;
    constexpr int S_ST0 = 11;                      // first k-step that carries LDS stores (3 per step)
    static_assert(NLOAD <= S_ST0 + 1 && NSTORE <= 3 * (20 - S_ST0), "side work must fit the k-steps");

    f32x4 acc[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NLOAD; ++q) load_one(0, q);
#pragma unroll
    for (int q = 0; q < NSTORE; ++q) store_one(0, q);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        const int kn = min(kt + 1, nkt - 1);       // the last tile re-stages itself (harmless) instead of branching
        const float* ap = &As[buf][kq][16 * wave + j];
        const float* xp = &Xs[buf][kq][j];
        // k' = 4s + kq: tap = s/4, channel = 4*(s%4) + kq.  One wave per SIMD, so everything else is threaded
        // through the MFMA stream by hand: the LDS operands of k-step s+1 are fetched while step s multiplies,
        // the global loads of the next tile go out one per step (steps 0..), its LDS stores three per step
        // (steps 11..19, into the buffer nobody reads during this tile)
        float av[2], bv[2][NTW];
        av[0] = ap[0];
#pragma unroll
        for (int n = 0; n < NTW; ++n) bv[0][n] = xp[16 * n];
#pragma unroll
        for (int s = 0; s < 20; ++s) {
            if (s < NLOAD) load_one(kn, s);
            if (s + 1 < 20) {
                av[(s + 1) & 1] = ap[4 * (s + 1) * CT_AS];
#pragma unroll
                for (int n = 0; n < NTW; ++n)
                    bv[(s + 1) & 1][n] = xp[4 * ((s + 1) & 3) * XS + 16 * n + ((s + 1) >> 2)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < NTW; ++n) acc[n] = mfma16x4(av[s & 1], bv[s & 1][n], acc[n]);
            __builtin_amdgcn_sched_barrier(0);
            if (s >= S_ST0) {
#pragma unroll
                for (int q = 3 * (s - S_ST0); q < 3 * (s - S_ST0) + 3; ++q) store_one(buf ^ 1, q);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }
    if (stamp) { a.prof[1] = __builtin_readcyclecounter(); a.prof[3] = wall_clock64(); }
    if (gridDim.z > 1) {
        // write-through partial, arrival counter, and the second arriver adds the first one's tile (a + b == b + a: the same
        // bits whichever half comes second); no fence — see the split-K epilogue of gemm.hip
        const size_t tile = (size_t)blockIdx.y * gridDim.x + blockIdx.x, ntile = (size_t)gridDim.x * gridDim.y;
        float* mine = a.ks_part + ((blockIdx.z * ntile + tile) * 256 + tid) * (4 * NTW);
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) st_sc1(mine + 4 * n + r, acc[n][r]);
        __shared__ unsigned second_;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) second_ = __hip_atomic_fetch_add(a.ks_ctr + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (second_ == 0u) return;
        if (tid == 0) __hip_atomic_store(a.ks_ctr + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float* other = a.ks_part + (((1 - blockIdx.z) * ntile + tile) * 256 + tid) * (4 * NTW);
        float o[NTW][4];
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[n][r] = ld_sc1(other + 4 * n + r);
#pragma unroll
        for (int n = 0; n < NTW; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[n][r] += o[n][r];
    }
    // epilogue: lane holds rows m0 + 16w + 4kq + r (r = 0..3) of column t0 + 16n + j
    float psum[4] = {0.f, 0.f, 0.f, 0.f}, psq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + 16 * wave + 4 * kq + r;
        if (m < a.M) {
            const float bv = a.bias ? a.bias[m] : 0.f;
            float* yrow = a.Y + ((size_t)bb * a.M + m) * a.T;
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                const int t = t0 + 16 * n + j;
                if (t < a.T) {
                    const float v = acc[n][r] + bv;
                    yrow[t] = v;
                    psum[r] += v;
                    psq[r] = fmaf(v, v, psq[r]);
                }
            }
        }
    }
    if (a.stat_part) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float s1 = row16_sum(psum[r]), s2 = row16_sum(psq[r]);
            const int m = m0 + 16 * wave + 4 * kq + r;
            if (j == 0 && m < a.M) {
                float* dst = a.stat_part + ((size_t)blockIdx.x * a.M + m) * 2;
                dst[0] = s1;
                dst[1] = s2;
            }
        }
    }
}

// weight gradient: dW[m][c][kx] = sum_{b,t} dY[b][m][t] X[b][c][t+kx-2].  Workgroup = 64 rows m x 16 channels
// (80 columns n = 5c + kx, contiguous in dW); K runs over (utterance, BT positions).
template <int BT>
__global__ __launch_bounds__(256) void k_conv5_dw(ConvTiledArgs a) {
    constexpr int XW = BT + 4;
    constexpr int XS = (BT == 80) ? 100 : 116;           // rows of 4 consecutive channels -> disjoint 8-bank windows
    constexpr int NA = BT / 16;                          // float4 of dY per thread (BT*64/4/256)
    constexpr int NX = (16 * XW + 255) / 256;
    __shared__ float As[2][BT][CT_AS];
    __shared__ float Xs[2][16][XS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kq = lane >> 4, j = lane & 15;
    const int c0 = blockIdx.x * 16, m0 = blockIdx.y * CT_BM;
    const int CK = a.Cin * 5;
    const int tiles = (a.T + BT - 1) / BT, nkt_all = a.B * tiles;
    // gridDim.z K-splits (layers with few (row, channel-block) tiles): this workgroup reduces k-tiles [kt_lo, kt_hi)
    const int per = (nkt_all + gridDim.z - 1) / gridDim.z;
    const int kt_lo = blockIdx.z * per, kt_hi = min(nkt_all, kt_lo + per);
    const bool vec = (a.T & 3) == 0;

    float4 ra[NA];
    float rx[NX];
    const int a_m = tid & 63;
    const bool a_ok = m0 + a_m < a.M;
    auto load_tiles = [&](int kt) {
        const int bb = kt / tiles, t0 = (kt - bb * tiles) * BT;
        const float* arow = a.dY + ((size_t)bb * a.M + (a_ok ? m0 + a_m : 0)) * a.T;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int t = t0 + 4 * ((tid >> 6) + 4 * i);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a_ok) {
                if (vec) {
                    if (t < a.T) v = *(const float4*)(arow + t);
                } else {
                    if (t < a.T) v.x = arow[t];
                    if (t + 1 < a.T) v.y = arow[t + 1];
                    if (t + 2 < a.T) v.z = arow[t + 2];
                    if (t + 3 < a.T) v.w = arow[t + 3];
                }
            }
            ra[i] = v;
        }
        const float* x_item = a.X + ((size_t)bb * a.Cin + c0) * a.T;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int e = tid + 256 * i;
            const int c = e / XW, jj = e - c * XW;
            const int t = t0 - 2 + jj;
            rx[i] = (c < 16 && t >= 0 && t < a.T) ? x_item[(size_t)c * a.T + t] : 0.f;
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int k = 4 * ((tid >> 6) + 4 * i);
            As[buf][k][a_m] = ra[i].x;
            As[buf][k + 1][a_m] = ra[i].y;
            As[buf][k + 2][a_m] = ra[i].z;
            As[buf][k + 3][a_m] = ra[i].w;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int e = tid + 256 * i;
            const int c = e / XW, jj = e - c * XW;
            if (c < 16) Xs[buf][c][jj] = rx[i];
        }
    };

    // column n = 16*nt + j of the 80-wide tile -> (channel, tap); B operand of k-step s: Xs[c][4s + kq + kx]
    int boff[5];
#pragma unroll
    for (int n = 0; n < 5; ++n) {
        const int nl = 16 * n + j, c = nl / 5, kx = nl - 5 * c;
        boff[n] = c * XS + kx + kq;
    }
    f32x4 acc[5];
#pragma unroll
    for (int n = 0; n < 5; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (kt_lo < kt_hi) {
        load_tiles(kt_lo);
        store_tiles(0);
    }
    __syncthreads();
    for (int kt = kt_lo; kt < kt_hi; ++kt) {
        const int buf = (kt - kt_lo) & 1;
        if (kt + 1 < kt_hi) load_tiles(kt + 1);
        const float* ap = &As[buf][kq][16 * wave + j];
        const float* xp = &Xs[buf][0][0];
        float av[2], bv[2][5];
        av[0] = ap[0];
#pragma unroll
        for (int n = 0; n < 5; ++n) bv[0][n] = xp[boff[n]];
#pragma unroll
        for (int s = 0; s < BT / 4; ++s) {
            if (s + 1 < BT / 4) {
                av[(s + 1) & 1] = ap[4 * (s + 1) * CT_AS];
#pragma unroll
                for (int n = 0; n < 5; ++n) bv[(s + 1) & 1][n] = xp[boff[n] + 4 * (s + 1)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < 5; ++n) acc[n] = mfma16x4(av[s & 1], bv[s & 1][n], acc[n]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kt + 1 < kt_hi) store_tiles(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + 16 * wave + 4 * kq + r;
        if (m < a.M) {
            float* drow = a.Y + (size_t)blockIdx.z * a.M * CK + (size_t)m * CK + 5 * c0;   // split partials are stacked
#pragma unroll
            for (int n = 0; n < 5; ++n) drow[16 * n + j] = acc[n][r];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// bf16 variant of the tiled forward / data-gradient kernel (hparams bf16_run, BASELINE configs[4]): operands are
// rounded to bf16 (RNE) on their way into LDS, products accumulate in fp32 on v_mfma_f32_16x16x16_bf16; inputs,
// outputs, bias and the BatchNorm statistics stay fp32.  One MFMA k-step = one tap x 16 channels.
//   weights : packed once per call by k_conv5_pack_bf16 into Wp[row][channel block][tap][16] (the data-gradient
//             flip/transpose is folded into that pass), so a tile's A rows are 160 contiguous bytes;
//   LDS     : As[tap][row][16 (+8 pad)] and Xt[position][16 (+8 pad)] (channel-contiguous, transposed while
//             staging): every MFMA operand is one ds_read_b64, rows 48 B apart -> conflict-free.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    bf16x2_t p = {(__bf16)lo, (__bf16)hi};
    return *(unsigned*)&p;
}
#define CB_RS 24     // LDS row stride in bf16 elements (48 B)

// W (M, Cin, 5) fp32 -> Wp bf16.  flipT = 0: rows = M, channels = Cin, Wp[r][cb][kx][c16] = W[r][16cb+c16][kx].
// flipT = 1 (data gradient): rows = Cin, channels = M, Wp[r][cb][kx][c16] = W[16cb+c16][r][4-kx].
__global__ void k_conv5_pack_bf16(const float* __restrict__ W, unsigned short* __restrict__ Wp, int M, int Cin, int flipT) {
    const int R = flipT ? Cin : M, Cc = flipT ? M : Cin;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)R * Cc * 5) return;
    const int c16 = i & 15, kx = (i >> 4) % 5;
    const size_t rc = (i >> 4) / 5;
    const int cb = rc % (Cc / 16), r = rc / (Cc / 16);
    const int cc = 16 * cb + c16;
    const float v = flipT ? W[((size_t)cc * Cin + r) * 5 + (4 - kx)] : W[((size_t)r * Cin + cc) * 5 + kx];
    const __bf16 b = (__bf16)v;
    Wp[i] = *(const unsigned short*)&b;
}

struct ConvBf16Args {
    const unsigned short* Wp;   // (M, Cin/16, 5, 16) bf16
    const float* X;             // (B, Cin, T)
    const float* bias;
    float* Y;                   // (B, M, T)
    float* stat_part;
    int B, Cin, T, M, tiles_per_item;
};

template <int NTW>
__global__ __launch_bounds__(256) void k_conv5_fwd_bf16(ConvBf16Args a) {
    constexpr int BN = 16 * NTW;
    constexpr int XW = BN + 4;
    constexpr int NXP = (8 * XW + 255) / 256;            // channel PAIRS x positions staged per thread
    __shared__ __attribute__((aligned(16))) unsigned short As[2][5][CT_BM][CB_RS];
    __shared__ __attribute__((aligned(16))) unsigned short Xt[2][XW][CB_RS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kq = lane >> 4, j = lane & 15;
    const int bb = blockIdx.x / a.tiles_per_item, t0 = (blockIdx.x % a.tiles_per_item) * BN;
    const int m0 = blockIdx.y * CT_BM;
    const int ncb = a.Cin / 16;

    // staging plan: A = 640 chunks of 16 B (row = q/10, chunk = q%10 -> tap = chunk/2, half = chunk&1)
    uint4 ra[3];
    const unsigned short* a_src[3];
    int a_lds[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int q = tid + 256 * i;
        const int row = min(q / 10, CT_BM - 1), ch = q - (q / 10) * 10;
        a_src[i] = a.Wp + (size_t)min(m0 + row, a.M - 1) * ncb * 80 + ch * 8;      // + 80 * kt
        a_lds[i] = q < 640 ? (((ch >> 1) * CT_BM + row) * CB_RS + 8 * (ch & 1)) : -1;
    }
    float rx[NXP][2];
    const float* x_item = a.X + (size_t)bb * a.Cin * a.T;
    int x_goff[NXP], x_lds[NXP];
    bool x_ok[NXP];
#pragma unroll
    for (int i = 0; i < NXP; ++i) {
        const int e = tid + 256 * i;
        const int cp = min(e / XW, 7), pos = e - (e / XW) * XW;
        const int t = t0 - 2 + pos;
        x_ok[i] = e < 8 * XW && t >= 0 && t < a.T;
        x_goff[i] = 2 * cp * a.T + min(max(t, 0), a.T - 1);
        x_lds[i] = e < 8 * XW ? pos * CB_RS + 2 * cp : -1;
    }
    auto load_tiles = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 3; ++i) ra[i] = *(const uint4*)(a_src[i] + 80 * kt);
#pragma unroll
        for (int i = 0; i < NXP; ++i) {
            const float* px = x_item + (size_t)16 * kt * a.T + x_goff[i];
            const float v0 = px[0], v1 = px[a.T];
            rx[i][0] = x_ok[i] ? v0 : 0.f;
            rx[i][1] = x_ok[i] ? v1 : 0.f;
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (a_lds[i] >= 0) *(uint4*)(&As[buf][0][0][0] + a_lds[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < NXP; ++i)
            if (x_lds[i] >= 0) *(unsigned*)(&Xt[buf][0][0] + x_lds[i]) = pack_bf16x2(rx[i][0], rx[i][1]);
    };

    f32x4 acc[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < ncb; ++kt) {
        const int buf = kt & 1;
        load_tiles(min(kt + 1, ncb - 1));
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) {
            const s16x4 av = *(const s16x4*)&As[buf][kx][16 * wave + j][4 * kq];
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                const s16x4 bv = *(const s16x4*)&Xt[buf][16 * n + j + kx][4 * kq];
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, bv, acc[n], 0, 0, 0);
            }
        }
        store_tiles(buf ^ 1);
        __syncthreads();
    }

    float psum[4] = {0.f, 0.f, 0.f, 0.f}, psq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + 16 * wave + 4 * kq + r;
        if (m < a.M) {
            const float bv = a.bias ? a.bias[m] : 0.f;
            float* yrow = a.Y + ((size_t)bb * a.M + m) * a.T;
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                const int t = t0 + 16 * n + j;
                if (t < a.T) {
                    const float v = acc[n][r] + bv;
                    yrow[t] = v;
                    psum[r] += v;
                    psq[r] = fmaf(v, v, psq[r]);
                }
            }
        }
    }
    if (a.stat_part) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float s1 = row16_sum(psum[r]), s2 = row16_sum(psq[r]);
            const int m = m0 + 16 * wave + 4 * kq + r;
            if (j == 0 && m < a.M) {
                float* dst = a.stat_part + ((size_t)blockIdx.x * a.M + m) * 2;
                dst[0] = s1;
                dst[1] = s2;
            }
        }
    }
}

// Round 5, second forward / data-gradient kernel for Cin % 32 == 0 (every wide convolution of the model): the same output tile and
// launch geometry, but (1) a k-tile is 32 input channels x 5 taps on v_mfma_f32_16x16x32_bf16 (twice the rate of the 16x16x16
// instruction, half the barriers), operands 16 bytes per lane ([tap][row][32 channels] / [position][32 channels], row stride 80 B:
// conflict-free ds_read_b128), and (2) TWO k-tiles of global loads are in flight behind the one being multiplied (two register
// sets, loop unrolled by two): the 16-channel kernel above spent ~3 us per k-tile — one memory round trip under load — for 400
// cycles of MFMA work (the B = 16 Postnet convolution: 16.8 GFLOP in 97 us).
#define CB2_RS 40    // LDS row stride in bf16 elements (80 B)
typedef __bf16 cb2_bf16x8 __attribute__((ext_vector_type(8)));
template <int NTW>
__global__ __launch_bounds__(256) void k_conv5_fwd_bf16k32(ConvBf16Args a) {
    constexpr int BN = 16 * NTW;
    constexpr int XW = BN + 4;
    constexpr int NXP = (16 * XW + 255) / 256;           // channel PAIRS x positions staged per thread
    __shared__ __attribute__((aligned(16))) unsigned short As[2][5][CT_BM][CB2_RS];
    __shared__ __attribute__((aligned(16))) unsigned short Xt[2][XW][CB2_RS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kq = lane >> 4, j = lane & 15;
    const int bb = blockIdx.x / a.tiles_per_item, t0 = (blockIdx.x % a.tiles_per_item) * BN;
    const int m0 = blockIdx.y * CT_BM;
    const int ncb = a.Cin / 16, nkt = a.Cin / 32;

    // staging plan: A = 1280 chunks of 16 B per k-tile (row = q/20, chunk = q%20 -> channel block = chunk/10, tap = (chunk%10)/2,
    // half = chunk&1; a row's 20 chunks are 320 contiguous bytes of Wp)
    const unsigned short* a_src[5];
    int a_lds[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int q = tid + 256 * i;
        const int row = q / 20, ch = q - row * 20;
        a_src[i] = a.Wp + (size_t)min(m0 + row, a.M - 1) * ncb * 80 + ch * 8;      // + 160 * kt
        a_lds[i] = (((ch % 10) >> 1) * CT_BM + row) * CB2_RS + 16 * (ch / 10) + 8 * (ch & 1);
    }
    const float* x_item = a.X + (size_t)bb * a.Cin * a.T;
    int x_goff[NXP], x_lds[NXP];
    bool x_ok[NXP];
#pragma unroll
    for (int i = 0; i < NXP; ++i) {
        const int e = tid + 256 * i;
        const int cp = min(e / XW, 15), pos = e - (e / XW) * XW;
        const int t = t0 - 2 + pos;
        x_ok[i] = e < 16 * XW && t >= 0 && t < a.T;
        x_goff[i] = 2 * cp * a.T + min(max(t, 0), a.T - 1);
        x_lds[i] = e < 16 * XW ? pos * CB2_RS + 2 * cp : -1;
    }
    uint4 ra0[5], ra1[5];
    float rx0[NXP][2], rx1[NXP][2];
    auto load_tiles = [&](uint4 (&ra)[5], float (&rx)[NXP][2], int kt) {
#pragma unroll
        for (int i = 0; i < 5; ++i) ra[i] = *(const uint4*)(a_src[i] + 160 * kt);
#pragma unroll
        for (int i = 0; i < NXP; ++i) {
            const float* px = x_item + (size_t)32 * kt * a.T + x_goff[i];
            const float v0 = px[0], v1 = px[a.T];
            rx[i][0] = x_ok[i] ? v0 : 0.f;
            rx[i][1] = x_ok[i] ? v1 : 0.f;
        }
    };
    auto store_tiles = [&](const uint4 (&ra)[5], const float (&rx)[NXP][2], int buf) {
#pragma unroll
        for (int i = 0; i < 5; ++i) *(uint4*)(&As[buf][0][0][0] + a_lds[i]) = ra[i];
#pragma unroll
        for (int i = 0; i < NXP; ++i)
            if (x_lds[i] >= 0) *(unsigned*)(&Xt[buf][0][0] + x_lds[i]) = pack_bf16x2(rx[i][0], rx[i][1]);
    };
    f32x4 acc[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto multiply = [&](int buf) {
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) {
            const cb2_bf16x8 av = *(const cb2_bf16x8*)&As[buf][kx][16 * wave + j][8 * kq];
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                const cb2_bf16x8 bv = *(const cb2_bf16x8*)&Xt[buf][16 * n + j + kx][8 * kq];
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[n], 0, 0, 0);
            }
        }
    };
    const int last = nkt - 1;
    load_tiles(ra0, rx0, 0);
    store_tiles(ra0, rx0, 0);
    load_tiles(ra0, rx0, min(1, last));
    load_tiles(ra1, rx1, min(2, last));
    __syncthreads();
    for (int kt = 0; kt < nkt; kt += 2) {
        // tile kt is in buffer 0, tile kt+1 in set 0 (in flight), tile kt+2 in set 1 (in flight)
        multiply(0);
        store_tiles(ra0, rx0, 1);                       // waits for set 0 only: set 1 stays in flight
        load_tiles(ra0, rx0, min(kt + 3, last));
        __syncthreads();
        if (kt + 1 < nkt) multiply(1);
        store_tiles(ra1, rx1, 0);
        load_tiles(ra1, rx1, min(kt + 4, last));
        __syncthreads();
    }

    float psum[4] = {0.f, 0.f, 0.f, 0.f}, psq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + 16 * wave + 4 * kq + r;
        if (m < a.M) {
            const float bv = a.bias ? a.bias[m] : 0.f;
            float* yrow = a.Y + ((size_t)bb * a.M + m) * a.T;
#pragma unroll
            for (int n = 0; n < NTW; ++n) {
                const int t = t0 + 16 * n + j;
                if (t < a.T) {
                    const float v = acc[n][r] + bv;
                    yrow[t] = v;
                    psum[r] += v;
                    psq[r] = fmaf(v, v, psq[r]);
                }
            }
        }
    }
    if (a.stat_part) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float s1 = row16_sum(psum[r]), s2 = row16_sum(psq[r]);
            const int m = m0 + 16 * wave + 4 * kq + r;
            if (j == 0 && m < a.M) {
                float* dst = a.stat_part + ((size_t)blockIdx.x * a.M + m) * 2;
                dst[0] = s1;
                dst[1] = s2;
            }
        }
    }
}

// bf16 weight gradient (round 5; hparams bf16_run): the tile of k_conv5_dw — 64 rows m x 16 channels (80 columns n = 5c + kx),
// K over (utterance, BT positions) — on v_mfma_f32_16x16x16_bf16 (16 positions per MFMA instead of 4).  dY and X are rounded
// to bf16 (RNE) on their way into LDS, products accumulate in fp32.  Both operands are K-major in memory already ((B, C, T):
// positions contiguous), so nothing is transposed while staging: a dY float4 becomes one 8-byte LDS store, and an MFMA operand
// is one ds_read_b64.  The tap shift would misalign the B reads (4 bf16 from position 4x + kx): X is staged as FOUR copies,
// copy r holding x[p + r] at index p, so tap kx reads copy kx & 3 at an 8-byte-aligned index.
template <int BT>
__global__ __launch_bounds__(256) void k_conv5_dw_bf16(ConvTiledArgs a) {
    constexpr int RS = BT + 8;                           // As row stride (bf16): 44 / 52 words -> 16 rows x 2 k-groups in disjoint bank pairs
    constexpr int XL = BT + 8;                           // staged positions per channel and copy
    constexpr int NA = BT / 16;                          // float4 of dY per thread (64 rows x BT / 4 / 256)
    constexpr int NQ = XL / 4;                           // 4-position groups per channel
    constexpr int NXI = (16 * NQ + 255) / 256;           // (channel, group) items per thread
    __shared__ __attribute__((aligned(16))) unsigned short As[2][64][RS];
    __shared__ __attribute__((aligned(16))) unsigned short Xs[2][4][16][XL];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kq = lane >> 4, j = lane & 15;
    const int c0 = blockIdx.x * 16, m0 = blockIdx.y * CT_BM;
    const int CK = a.Cin * 5;
    const int tiles = (a.T + BT - 1) / BT, nkt_all = a.B * tiles;
    const int per = (nkt_all + gridDim.z - 1) / gridDim.z;
    const int kt_lo = blockIdx.z * per, kt_hi = min(nkt_all, kt_lo + per);
    const bool vec = (a.T & 3) == 0;

    float4 ra[NA];
    float rx[NXI][7];
    const int a_m = tid & 63;
    const bool a_ok = m0 + a_m < a.M;
    auto load_tiles = [&](int kt) {
        const int bb = kt / tiles, t0 = (kt - bb * tiles) * BT;
        const float* arow = a.dY + ((size_t)bb * a.M + (a_ok ? m0 + a_m : 0)) * a.T;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int t = t0 + 4 * ((tid >> 6) + 4 * i);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a_ok) {
                if (vec) {
                    if (t < a.T) v = *(const float4*)(arow + t);
                } else {
                    if (t < a.T) v.x = arow[t];
                    if (t + 1 < a.T) v.y = arow[t + 1];
                    if (t + 2 < a.T) v.z = arow[t + 2];
                    if (t + 3 < a.T) v.w = arow[t + 3];
                }
            }
            ra[i] = v;
        }
        const float* x_item = a.X + ((size_t)bb * a.Cin + c0) * a.T;
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const int e = tid + 256 * i;
            const int c = min(e / NQ, 15), q = e - (e / NQ) * NQ;
            const float* xr = x_item + (size_t)c * a.T;
#pragma unroll
            for (int d = 0; d < 7; ++d) {
                const int pz = 4 * q + d, t = t0 - 2 + pz;           // staged position pz <-> time t0 - 2 + pz (2-halo)
                const bool ok = e < 16 * NQ && pz < BT + 4 && t >= 0 && t < a.T;
                const float v = xr[min(max(t, 0), a.T - 1)];
                rx[i][d] = ok ? v : 0.f;
            }
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int k = 4 * ((tid >> 6) + 4 * i);
            *(uint2*)&As[buf][a_m][k] = make_uint2(pack_bf16x2(ra[i].x, ra[i].y), pack_bf16x2(ra[i].z, ra[i].w));
        }
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            const int e = tid + 256 * i;
            const int c = e / NQ, q = e - c * NQ;
            if (c < 16) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    *(uint2*)&Xs[buf][r][c][4 * q] = make_uint2(pack_bf16x2(rx[i][r], rx[i][r + 1]), pack_bf16x2(rx[i][r + 2], rx[i][r + 3]));
            }
        }
    };

    // column n = 16*nt + j of the 80-wide tile -> (channel, tap); B operand of k-step s: copy kx & 3, positions 16 s + 4 kq + (kx & 4)
    int boff[5];
#pragma unroll
    for (int n = 0; n < 5; ++n) {
        const int nl = 16 * n + j, c = nl / 5, kx = nl - 5 * c;
        boff[n] = ((kx & 3) * 16 + c) * XL + 4 * kq + (kx & 4);
    }
    f32x4 acc[5];
#pragma unroll
    for (int n = 0; n < 5; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (kt_lo < kt_hi) {
        load_tiles(kt_lo);
        store_tiles(0);
    }
    __syncthreads();
    for (int kt = kt_lo; kt < kt_hi; ++kt) {
        const int buf = (kt - kt_lo) & 1;
        if (kt + 1 < kt_hi) load_tiles(kt + 1);
        const unsigned short* ap = &As[buf][16 * wave + j][4 * kq];
        const unsigned short* xp = &Xs[buf][0][0][0];
#pragma unroll
        for (int s = 0; s < BT / 16; ++s) {
            const s16x4 av = *(const s16x4*)(ap + 16 * s);
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                const s16x4 bv = *(const s16x4*)(xp + boff[n] + 16 * s);
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, bv, acc[n], 0, 0, 0);
            }
        }
        if (kt + 1 < kt_hi) store_tiles(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + 16 * wave + 4 * kq + r;
        if (m < a.M) {
            float* drow = a.Y + (size_t)blockIdx.z * a.M * CK + (size_t)m * CK + 5 * c0;   // split partials are stacked
#pragma unroll
            for (int n = 0; n < 5; ++n) drow[16 * n + j] = acc[n][r];
        }
    }
}

// Output positions per workgroup (BN = 32/48/64/80/96).  Cost model: workgroups run in rounds of one per CU; a
// workgroup's time per k-tile is its MFMA work (~BN) plus the fixed staging cost (~40 in the same units).  Small
// problems (encoder bank: 6 x 84 positions; 80-row layers) therefore take narrow tiles — more workgroups in the one
// round they need — and the big Postnet layers the width that divides T (80 for T=400: 240 workgroups).
// fixed-order sum of the K-split partials of the weight gradient
__global__ void k_conv5_dw_reduce(const float* __restrict__ part, float* __restrict__ dw, int n, int nsplit) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += part[(size_t)k * n + i];
    dw[i] = s;
}

// K-splits of the weight gradient: only when the (row tile, channel block) grid is far from filling the chip
// Round 5: with ONE workgroup per CU every k-tile waits for its own global loads (one tile of prefetch); cut along K until several
// workgroups share a CU, the launch alone runs 512->512 at B = 6, T = 400 in 83 instead of 107 us (fp32) and 62 instead of 100 us
// (bf16 tiles), at B = 16 in 208 / 137 instead of 286 / 265 us (tools/dbg/conv_dw_time.py).  INSIDE the step it pays only for
// the bf16 kernel (bf16 B = 16 step 13.84 -> 13.78 ms): behind the reverse pass four streams share the chip and already fill each
// other's stalls — the fp32 step got 0.06 ms SLOWER with the extra workgroups and reduce launches (same-box A/B against the
// round-4 tree), so the fp32 kernel keeps its rule (split only while the launch does not fill the chip)
static inline int conv5_dw_splits(int B, int Cin, int T, int Cout, bool bf16 = false) {
    const int wgs = (Cin / 16) * ((Cout + CT_BM - 1) / CT_BM);
    const int c80 = ((T + 79) / 80), c96 = ((T + 95) / 96);
    const int nkt = B * (c80 * 80 <= c96 * 96 ? c80 : c96);
    // (T2V_CONV_DW_SPLITS = target workgroups per launch, measurement override)
    static const int want_env = getenv("T2V_CONV_DW_SPLITS") ? atoi(getenv("T2V_CONV_DW_SPLITS")) : 0;
    const int want = want_env > 0 ? want_env : (bf16 ? 8 * T2V_NWG : T2V_NWG);
    int ns = 1;
    while (wgs * ns * 2 <= want && ns * 2 <= nkt && ns < 8) ns *= 2;
    return ns;
}

// scratch of the split launches: slices of one ring (a captured graph keeps the slices of its nodes; a slice comes round again
// 64 MB of partial tiles later, long after its launch has retired)
#include <atomic>
static float* conv_ks_scratch(size_t floats) {
    constexpr size_t RING = (size_t)16 << 20;       // floats (64 MB)
    static float* ring = nullptr;
    static std::atomic<size_t> pos{0};
    static std::atomic<int> state{0};
    floats = (floats + 63) & ~(size_t)63;
    if (floats > RING / 4) return nullptr;
    if (state.load(std::memory_order_acquire) != 2) {
        int expect = 0;
        if (state.compare_exchange_strong(expect, 1)) {
            float* p = nullptr;
            if (hipMalloc((void**)&p, RING * sizeof(float)) != hipSuccess) { state.store(0); return nullptr; }
            ring = p;
            state.store(2, std::memory_order_release);
        } else {
            while (state.load(std::memory_order_acquire) == 1) { }
            if (state.load() != 2) return nullptr;
        }
    }
    size_t at = pos.fetch_add(floats) % RING;
    if (at + floats > RING) { pos.store(floats); at = 0; }
    return ring + at;
}

static inline int conv5_pick_bn(int T, int B, int M) {
    if (const char* e = getenv("T2V_CONV_BN")) { const int v = atoi(e); if (v == 32 || v == 48 || v == 64 || v == 80 || v == 96) return v; }
    const int cands[5] = {32, 48, 64, 80, 96};
    int best = 80;
    long best_cost = -1;
    for (int i = 0; i < 5; ++i) {
        const int bn = cands[i];
        const long wgs = (long)B * ((T + bn - 1) / bn) * ((M + CT_BM - 1) / CT_BM);
        const long cost = ((wgs + T2V_NWG - 1) / T2V_NWG) * (bn + 40);
        if (best_cost < 0 || cost < best_cost || (cost == best_cost && bn > best)) { best = bn; best_cost = cost; }
    }
    return best;
}
// forward tiles with the optional cut of the input channels in two: the same cost model, a split launch does half the k-tiles
// per workgroup (+ ~8 columns' worth of exchange) but only pays when all its workgroups get a CU of their own
static inline int conv5_pick_bn_ks(int T, int B, int M, int Cin, int* ks) {
    static const int ksplit_on = getenv("T2V_CONV_KSPLIT") ? atoi(getenv("T2V_CONV_KSPLIT")) : 1;
    *ks = 1;
    const int bn1 = conv5_pick_bn(T, B, M);
    if (!ksplit_on || Cin < 256 || getenv("T2V_CONV_BN")) return bn1;
    const int cands[5] = {32, 48, 64, 80, 96};
    const long w1 = (long)B * ((T + bn1 - 1) / bn1) * ((M + CT_BM - 1) / CT_BM);
    long best_cost = 2 * ((w1 + T2V_NWG - 1) / T2V_NWG) * (bn1 + 40);         // (costs doubled: halves stay integers)
    int best = bn1;
    for (int i = 0; i < 5; ++i) {
        const int bn = cands[i];
        const long wgs = 2 * (long)B * ((T + bn - 1) / bn) * ((M + CT_BM - 1) / CT_BM);
        // (T2V_CONV_KSPLIT=2, measurement: also split launches that then put TWO half-size workgroups on a CU)
        if (wgs > (ksplit_on == 2 ? 2 * T2V_NWG : T2V_NWG)) continue;
        const long cost = ((wgs + T2V_NWG - 1) / T2V_NWG) * ((bn + 40) + 16);
        if (cost < best_cost) { best = bn; best_cost = cost; *ks = 2; }
    }
    return best;
}
static inline bool conv5_tiled_ok(int Cin, int KS) { return KS == 5 && Cin % 16 == 0; }
// round 6 (conv_x3.hip): fp32 forward / data gradient on the bf16 matrix cores from exactly 3-way-split operands
bool t2v_conv5_x3_ok(int B, int Cin, int T, int Cout, int KS);
int t2v_conv5_x3_stat_blocks(int B, int T);
int t2v_conv5_x3_run(const float* W, const float* X, const float* bias, float* Y, float* stat_part, int B, int Cin, int T, int M,
                     hipStream_t stream, int np);
bool t2v_conv5_planes_bf16_ok(int B, int Cin, int T, int Cout, int KS);

static void launch_conv5_fwd(const float* W, const float* X, const float* bias, float* Y, float* stat_part, int B,
                             int Cin, int T, int M, hipStream_t stream) {
    int ks = 1;
    const int BN = conv5_pick_bn_ks(T, B, M, Cin, &ks);
    ConvTiledArgs a;
    a.W = W; a.X = X; a.dY = nullptr; a.bias = bias; a.Y = Y; a.stat_part = stat_part;
    a.B = B; a.Cin = Cin; a.T = T; a.M = M; a.tiles_per_item = (T + BN - 1) / BN;
    a.prof = g_t2v_prof;
    a.ks_part = nullptr; a.ks_ctr = nullptr;
    dim3 grid(B * a.tiles_per_item, (M + CT_BM - 1) / CT_BM);
    // a launch that leaves half the chip idle and has a deep K: the input channels are cut in two (T2V_CONV_KSPLIT=0 switches
    // it off for measurements)
    const long ntile = (long)grid.x * grid.y;
    if (ks == 2) {
        const size_t pf = (size_t)2 * ntile * 256 * 4 * (BN / 16);
        a.ks_part = conv_ks_scratch(pf);
        a.ks_ctr = a.ks_part ? t2v_arrival_counters((int)ntile) : nullptr;
        if (a.ks_part && a.ks_ctr) grid.z = 2; else a.ks_part = nullptr;
    }
    if (BN == 32) k_conv5_fwd<2><<<grid, 256, 0, stream>>>(a);
    else if (BN == 48) k_conv5_fwd<3><<<grid, 256, 0, stream>>>(a);
    else if (BN == 64) k_conv5_fwd<4><<<grid, 256, 0, stream>>>(a);
    else if (BN == 80) k_conv5_fwd<5><<<grid, 256, 0, stream>>>(a);
    else k_conv5_fwd<6><<<grid, 256, 0, stream>>>(a);
}

static void launch_conv5_fwd_bf16(const float* W, int flipT, unsigned short* Wp, const float* X, const float* bias,
                                  float* Y, float* stat_part, int B, int Cin, int T, int M, int W_M, int W_Cin,
                                  hipStream_t stream) {
    const size_t n = (size_t)W_M * W_Cin * 5;
    k_conv5_pack_bf16<<<(unsigned)((n + 255) / 256), 256, 0, stream>>>(W, Wp, W_M, W_Cin, flipT);
    int ks_unused = 1;      // (the same tile width as the fp32 launch: t2v_conv1d_stat_blocks answers for both)
    const int BN = conv5_pick_bn_ks(T, B, M, Cin, &ks_unused);
    ConvBf16Args a;
    a.Wp = Wp; a.X = X; a.bias = bias; a.Y = Y; a.stat_part = stat_part;
    a.B = B; a.Cin = Cin; a.T = T; a.M = M; a.tiles_per_item = (T + BN - 1) / BN;
    dim3 grid(B * a.tiles_per_item, (M + CT_BM - 1) / CT_BM);
    static const int k32 = getenv("T2V_CONV_BF16_K32") ? atoi(getenv("T2V_CONV_BF16_K32")) : 1;      // 0: the 16-channel kernel (measurement)
    if (k32 && Cin % 32 == 0) {
        if (BN == 32) k_conv5_fwd_bf16k32<2><<<grid, 256, 0, stream>>>(a);
        else if (BN == 48) k_conv5_fwd_bf16k32<3><<<grid, 256, 0, stream>>>(a);
        else if (BN == 64) k_conv5_fwd_bf16k32<4><<<grid, 256, 0, stream>>>(a);
        else if (BN == 80) k_conv5_fwd_bf16k32<5><<<grid, 256, 0, stream>>>(a);
        else k_conv5_fwd_bf16k32<6><<<grid, 256, 0, stream>>>(a);
        return;
    }
    if (BN == 32) k_conv5_fwd_bf16<2><<<grid, 256, 0, stream>>>(a);
    else if (BN == 48) k_conv5_fwd_bf16<3><<<grid, 256, 0, stream>>>(a);
    else if (BN == 64) k_conv5_fwd_bf16<4><<<grid, 256, 0, stream>>>(a);
    else if (BN == 80) k_conv5_fwd_bf16<5><<<grid, 256, 0, stream>>>(a);
    else k_conv5_fwd_bf16<6><<<grid, 256, 0, stream>>>(a);
}

// W (M, Cin, KS) -> Wt (Cin, M, KS) with the taps flipped: conv(dY, Wt) is the data gradient
__global__ void k_conv_flip_weight(const float* __restrict__ W, float* __restrict__ Wt, int M, int Cin, int KS) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * Cin * KS) return;
    const int k = i % KS, c = (i / KS) % Cin, m = i / (KS * Cin);
    Wt[((size_t)c * M + m) * KS + (KS - 1 - k)] = W[i];
}

// Several layers' weights flipped by ONE launch (grid.y = layer): a training step issues it once, on a side stream
// during the forward pass, instead of one flip launch in front of every data-gradient convolution.
struct ConvFlipBatch { const float* W[16]; float* Wt[16]; int M[16]; int Cin[16]; int n; int KS; };
__global__ void k_conv_flip_weight_batched(ConvFlipBatch f) {
    const int l = blockIdx.y;
    const int M = f.M[l], Cin = f.Cin[l], KS = f.KS;
    const float* __restrict__ W = f.W[l];
    float* __restrict__ Wt = f.Wt[l];
    const int n = M * Cin * KS;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int k = i % KS, c = (i / KS) % Cin, m = i / (KS * Cin);
        Wt[((size_t)c * M + m) * KS + (KS - 1 - k)] = W[i];
    }
}

extern "C" int t2v_conv1d_flip_weights(const float* const* W, float* const* Wt, const int* Cout, const int* Cin, int KS,
                                       int n, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!W || !Wt || !Cout || !Cin || n < 1 || n > 16 || KS < 1) return T2V_ERR_ARG;
    ConvFlipBatch f;
    f.n = n; f.KS = KS;
    int mx = 0;
    for (int i = 0; i < n; ++i) {
        if (!W[i] || !Wt[i] || Cout[i] < 1 || Cin[i] < 1) return T2V_ERR_ARG;
        f.W[i] = W[i]; f.Wt[i] = Wt[i]; f.M[i] = Cout[i]; f.Cin[i] = Cin[i];
        mx = max(mx, Cout[i] * Cin[i] * KS);
    }
    const int bx = min(1024, (mx + 1023) / 1024);
    k_conv_flip_weight_batched<<<dim3(bx, n), 256, 0, stream>>>(f);
    return t2v_check_launch();
}

extern "C" int t2v_conv1d_fwd(const float* W, const float* X, const float* bias, float* Y, float* stat_part,
                              int B, int Cin, int T, int Cout, int KS, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!W || !X || !Y || B < 1 || Cin < 1 || T < 1 || Cout < 1 || KS < 1 || !(KS & 1)) return T2V_ERR_ARG;
    if (t2v_conv5_x3_ok(B, Cin, T, Cout, KS)) {
        const int rc = t2v_conv5_x3_run(W, X, bias, Y, stat_part, B, Cin, T, Cout, stream, 3);
        return rc != T2V_OK ? rc : t2v_check_launch();
    }
    if (conv5_tiled_ok(Cin, KS)) {
        launch_conv5_fwd(W, X, bias, Y, stat_part, B, Cin, T, Cout, stream);
        return t2v_check_launch();
    }
    ConvGemmArgs a;
    a.W = W; a.X = X; a.dY = nullptr; a.bias = bias; a.Y = Y; a.stat_part = stat_part;
    a.B = B; a.Cin = Cin; a.T = T; a.M = Cout; a.KS = KS;
    dim3 grid((B * T + CG_BN - 1) / CG_BN, (Cout + CG_BM - 1) / CG_BM);
    if (KS == 5) k_conv_gemm<0, 5><<<grid, 256, 0, stream>>>(a);
    else if (KS == 3) k_conv_gemm<0, 3><<<grid, 256, 0, stream>>>(a);
    else return T2V_ERR_DIMS;
    return t2v_check_launch();
}

extern "C" int t2v_conv1d_stat_blocks(int B, int T, int Cin, int Cout, int KS) {
    if (t2v_conv5_x3_ok(B, Cin, T, Cout, KS)) return t2v_conv5_x3_stat_blocks(B, T);
    if (conv5_tiled_ok(Cin, KS)) { int ks; const int BN = conv5_pick_bn_ks(T, B, Cout, Cin, &ks); return B * ((T + BN - 1) / BN); }
    return (B * T + CG_BN - 1) / CG_BN;
}

// ... for t2v_conv1d_fwd_bf16 (its kernels keep the tile choice of the fp32-MFMA kernels whatever the x3 mode says)
extern "C" int t2v_conv1d_stat_blocks_bf16(int B, int T, int Cin, int Cout, int KS) {
    if (t2v_conv5_planes_bf16_ok(B, Cin, T, Cout, KS)) return t2v_conv5_x3_stat_blocks(B, T);
    if (conv5_tiled_ok(Cin, KS)) { int ks; const int BN = conv5_pick_bn_ks(T, B, Cout, Cin, &ks); return B * ((T + BN - 1) / BN); }
    return (B * T + CG_BN - 1) / CG_BN;
}

extern "C" int t2v_conv1d_dw_scratch_floats(int B, int Cin, int T, int Cout, int KS) {
    if (!conv5_tiled_ok(Cin, KS)) return 0;
    const int ns = conv5_dw_splits(B, Cin, T, Cout, true);        // (the bf16 kernel splits at least as far as the fp32 one)
    return ns > 1 ? ns * Cout * Cin * 5 : 0;
}

static thread_local bool g_conv_dw_bf16 = false;     // set by t2v_conv1d_bwd_bf16 around its call of t2v_conv1d_bwd (weight gradient on bf16 MFMA)
extern "C" int t2v_conv1d_bwd(const float* W, const float* X, const float* dY, float* dX, float* dW, float* Wt_scratch,
                              float* dw_scratch, int B, int Cin, int T, int Cout, int KS, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    // W == NULL: Wt_scratch already holds the flipped, transposed weight (t2v_conv1d_flip_weights ran earlier in the step)
    if (!X || !dY || B < 1 || Cin < 1 || T < 1 || Cout < 1 || KS < 1 || !(KS & 1)) return T2V_ERR_ARG;
    if (dX) {
        if (!Wt_scratch) return T2V_ERR_ARG;
        const int n = Cout * Cin * KS;
        if (W) k_conv_flip_weight<<<(n + 255) / 256, 256, 0, stream>>>(W, Wt_scratch, Cout, Cin, KS);
        if (t2v_conv5_x3_ok(B, Cout, T, Cin, KS)) {         // the data gradient = the same convolution with the flipped, transposed weights
            const int rc = t2v_conv5_x3_run(Wt_scratch, dY, nullptr, dX, nullptr, B, Cout, T, Cin, stream, 3);
            if (rc != T2V_OK) return rc;
        } else if (conv5_tiled_ok(Cout, KS)) {
            launch_conv5_fwd(Wt_scratch, dY, nullptr, dX, nullptr, B, Cout, T, Cin, stream);
        } else {
            ConvGemmArgs a;
            a.W = Wt_scratch; a.X = dY; a.dY = nullptr; a.bias = nullptr; a.Y = dX; a.stat_part = nullptr;
            a.B = B; a.Cin = Cout; a.T = T; a.M = Cin; a.KS = KS;
            dim3 grid((B * T + CG_BN - 1) / CG_BN, (Cin + CG_BM - 1) / CG_BM);
            if (KS == 5) k_conv_gemm<0, 5><<<grid, 256, 0, stream>>>(a);
            else if (KS == 3) k_conv_gemm<0, 3><<<grid, 256, 0, stream>>>(a);
            else return T2V_ERR_DIMS;
        }
    }
    if (dW && conv5_tiled_ok(Cin, KS)) {
        const bool dw_bf16 = g_conv_dw_bf16;
        ConvTiledArgs a;
        a.W = nullptr; a.X = X; a.dY = dY; a.bias = nullptr; a.Y = dW; a.stat_part = nullptr;
        a.B = B; a.Cin = Cin; a.T = T; a.M = Cout; a.tiles_per_item = 0;
        a.prof = nullptr; a.ks_part = nullptr; a.ks_ctr = nullptr;
        const int ns = conv5_dw_splits(B, Cin, T, Cout, dw_bf16);
        if (ns > 1 && !dw_scratch) return T2V_ERR_ARG;
        if (ns > 1) a.Y = dw_scratch;
        dim3 grid(Cin / 16, (Cout + CT_BM - 1) / CT_BM, ns);
        const int c80 = ((T + 79) / 80) * 80, c96 = ((T + 95) / 96) * 96;
        if (dw_bf16) {
            if (c80 <= c96) k_conv5_dw_bf16<80><<<grid, 256, 0, stream>>>(a);
            else k_conv5_dw_bf16<96><<<grid, 256, 0, stream>>>(a);
        } else if (c80 <= c96) k_conv5_dw<80><<<grid, 256, 0, stream>>>(a);
        else k_conv5_dw<96><<<grid, 256, 0, stream>>>(a);
        if (ns > 1) {
            const int n = Cout * Cin * 5;
            k_conv5_dw_reduce<<<(n + 255) / 256, 256, 0, stream>>>(dw_scratch, dW, n, ns);
        }
    } else if (dW) {
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

// ---- bf16_run entry points: same contracts as t2v_conv1d_fwd / t2v_conv1d_bwd, plus a bf16 scratch of W's
// element count (2 bytes each).  Forward and data gradient run on bf16 MFMA, the weight gradient stays on the
// fp32 kernel.  T2V_ERR_DIMS when the shape is outside the tiled bf16 path (caller uses the fp32 entry points).
extern "C" int t2v_conv1d_fwd_bf16(const float* W, const float* X, const float* bias, float* Y, float* stat_part,
                                   void* Wp_scratch, int B, int Cin, int T, int Cout, int KS, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!W || !X || !Y || !Wp_scratch || B < 1 || Cin < 1 || T < 1 || Cout < 1) return T2V_ERR_ARG;
    if (!conv5_tiled_ok(Cin, KS)) return T2V_ERR_DIMS;
    if (t2v_conv5_planes_bf16_ok(B, Cin, T, Cout, KS)) {        // round 6: pre-rounded planes + the LDS-DMA kernel (conv_x3.hip, one plane)
        const int rc = t2v_conv5_x3_run(W, X, bias, Y, stat_part, B, Cin, T, Cout, stream, 1);
        return rc != T2V_OK ? rc : t2v_check_launch();
    }
    launch_conv5_fwd_bf16(W, 0, (unsigned short*)Wp_scratch, X, bias, Y, stat_part, B, Cin, T, Cout, Cout, Cin, stream);
    return t2v_check_launch();
}

extern "C" int t2v_conv1d_bwd_bf16(const float* W, const float* X, const float* dY, float* dX, float* dW,
                                   void* Wp_scratch, float* dw_scratch, int B, int Cin, int T, int Cout, int KS,
                                   void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!X || !dY || B < 1 || Cin < 1 || T < 1 || Cout < 1) return T2V_ERR_ARG;
    if (!conv5_tiled_ok(Cin, KS) || (dX && !conv5_tiled_ok(Cout, KS))) return T2V_ERR_DIMS;
    if (dX) {
        if (!W || !Wp_scratch) return T2V_ERR_ARG;
        bool done = false;
        if (t2v_conv5_planes_bf16_ok(B, Cout, T, Cin, KS)) {
            // round 6: the data gradient = the same convolution with the flipped, transposed weights, on the one-plane form of
            // conv_x3.hip (the flipped fp32 copy lives in the library's ring: the caller's scratch is sized for bf16)
            float* wt = conv_ks_scratch((size_t)Cout * Cin * KS);
            if (wt) {
                k_conv_flip_weight<<<(Cout * Cin * KS + 255) / 256, 256, 0, stream>>>(W, wt, Cout, Cin, KS);
                const int rc = t2v_conv5_x3_run(wt, dY, nullptr, dX, nullptr, B, Cout, T, Cin, stream, 1);
                if (rc != T2V_OK) return rc;
                done = true;
            }
        }
        if (!done) launch_conv5_fwd_bf16(W, 1, (unsigned short*)Wp_scratch, dY, nullptr, dX, nullptr, B, Cout, T, Cin, Cout, Cin, stream);
    }
    if (dW) {           // (round 5) the weight gradient on bf16 MFMA as well: same tiling / K-splits as the fp32 kernel
        g_conv_dw_bf16 = true;
        const int rc = t2v_conv1d_bwd(W, X, dY, nullptr, dW, nullptr, dw_scratch, B, Cin, T, Cout, KS, stream_);
        g_conv_dw_bf16 = false;
        return rc;
    }
    return t2v_check_launch();
}
