// Round 6: the k = 5 Conv1d forward / data gradient in fp32 on the bf16 matrix cores ("x3", see gemm.hip: every fp32 operand is cut
// EXACTLY into three bf16 values a = a0 + a1 + a2 and a product block is accumulated in fp32 from six bf16 MFMAs — a0b0 + a0b1 + a1b0 +
// a1b1 + a0b2 + a2b0; the terms left out are <= 2^-25 |a b|, below the rounding of one fp32 product).  Encoder conv bank
// (reference model.py:159-177) and Postnet (model.py:110-148), forward and data gradient:
//      Y[b][m][t] = sum_tap sum_c W[m][c][tap] * X[b][c][t + tap - 2]     = five GEMMs whose B operand is shifted by the tap.
// Three launches:
//  (1) k_cx3_split_w: the weights once per call — planes [tap][plane][channel group of 8][row m] of 16-byte words (an MFMA A operand),
//      rows padded to 128, channels to 16;
//  (2) k_cx3_split_x: the activations once — planes [item][plane][channel group][position slot] with slot = t + 2 (two zero halo slots
//      on either side, zeros behind T), so the B operand of tap k at output position t is slot t + k: ONE staged tile serves all taps;
//  (3) k_conv5_x3: 128 output channels x 128 positions per workgroup (tiles never straddle utterances), 2x2 waves x (2x2)
//      accumulators of v_mfma_f32_32x32x16_bf16.  A step = (16 input channels, one tap) = 24 MFMAs per wave; the weight tile of the next
//      step and the activation tile of the next 16 channels arrive by LDS-DMA while this step multiplies (both double-buffered, ONE
//      barrier per step, 51 KB of LDS, < 168 registers: three workgroups per CU).  Launches that leave CUs idle are cut over the input
//      channels (gridDim.z), raw tiles to scratch, the last arriver adds them in fixed order (deterministic) and runs the epilogue:
//      bias, store, and the per-channel partial sums / sums of squares BatchNorm needs (fixed order as well).
#include <stdlib.h>
#include <atomic>
#include "t2v_common.h"
#include "t2v_kernels.h"
#include "t2v_coop.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 cx_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 cx_bf16x2 __attribute__((ext_vector_type(2)));
#define CX_BM 128
#define CX_BN 128
#define CX_XS 136                   // activation slots per tile and channel group: 128 positions + 4 halo (+ 4: the DMA's last piece is 8 lanes)

__device__ __forceinline__ unsigned cx_pack(float lo, float hi) {
    cx_bf16x2 p = {(__bf16)lo, (__bf16)hi};
    return *(unsigned*)&p;
}
__device__ __forceinline__ void cx_split8(const float (&v)[8], uint4& p0, uint4& p1, uint4& p2) {
    unsigned q0[4], q1[4], q2[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float x = v[2 * c], y = v[2 * c + 1];
        const unsigned h = cx_pack(x, y);
        const float r1x = x - __uint_as_float(h << 16), r1y = y - __uint_as_float(h & 0xffff0000u);
        const unsigned m = cx_pack(r1x, r1y);
        const float r2x = r1x - __uint_as_float(m << 16), r2y = r1y - __uint_as_float(m & 0xffff0000u);
        q0[c] = h; q1[c] = m; q2[c] = cx_pack(r2x, r2y);
    }
    p0 = make_uint4(q0[0], q0[1], q0[2], q0[3]);
    p1 = make_uint4(q1[0], q1[1], q1[2], q1[3]);
    p2 = make_uint4(q2[0], q2[1], q2[2], q2[3]);
}

// NP = 3: the x3 planes of an fp32 operand; NP = 1: ONE plane, the operand rounded to bf16 (bf16_run)
__device__ __forceinline__ uint4 cx_round8(const float (&v)[8]) {
    return make_uint4(cx_pack(v[0], v[1]), cx_pack(v[2], v[3]), cx_pack(v[4], v[5]), cx_pack(v[6], v[7]));
}
// W (M, Cin, 5) fp32 -> Wp[tap][plane][g][Mp]: thread = (row m, channel group g), all five taps
template <int NP>
__global__ __launch_bounds__(256) void k_cx3_split_w(const float* __restrict__ W, uint4* __restrict__ Wp, int M, int Cin, int Mp, int G) {
    const int m = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
    if (m >= Mp) return;
    const bool min_ = m < M;
    const float* row = W + ((size_t)min(m, M - 1) * Cin + 8 * g) * 5;
    float w[8][5];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const bool ok = min_ && 8 * g + c < Cin;
            const float v = row[ok ? 5 * c + k : 0];
            w[c][k] = ok ? v : 0.f;
        }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = w[c][k];
        if (NP == 3) {
            uint4 p0, p1, p2;
            cx_split8(v, p0, p1, p2);
            Wp[((size_t)(k * NP + 0) * G + g) * Mp + m] = p0;
            Wp[((size_t)(k * NP + (NP > 1 ? 1 : 0)) * G + g) * Mp + m] = p1;
            Wp[((size_t)(k * NP + (NP > 2 ? 2 : 0)) * G + g) * Mp + m] = p2;
        } else {
            Wp[((size_t)(k * NP) * G + g) * Mp + m] = cx_round8(v);
        }
    }
}
// X (B, Cin, T) fp32 -> Xp[b][plane][g][Tp], slot s <-> position s - 2: thread = (slot s, group g, item b)
template <int NP>
__global__ __launch_bounds__(256) void k_cx3_split_x(const float* __restrict__ X, uint4* __restrict__ Xp, int Cin, int T, int Tp, int G) {
    const int s = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y, b = blockIdx.z;
    if (s >= Tp) return;
    const int t = s - 2;
    const bool tin = t >= 0 && t < T;
    const float* col = X + ((size_t)b * Cin + 8 * g) * T + min(max(t, 0), T - 1);
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const bool ok = tin && 8 * g + c < Cin;
        const float x = col[ok ? (size_t)c * T : 0];
        v[c] = ok ? x : 0.f;
    }
    uint4* dst = Xp + (((size_t)b * NP) * G + g) * Tp + s;
    if (NP == 3) {
        uint4 p0, p1, p2;
        cx_split8(v, p0, p1, p2);
        dst[0] = p0;
        dst[(size_t)(NP > 1 ? 1 : 0) * G * Tp] = p1;
        dst[(size_t)(NP > 2 ? 2 : 0) * G * Tp] = p2;
    } else {
        dst[0] = cx_round8(v);
    }
}

struct ConvX3Args {
    const uint4* Wp; const uint4* Xp;
    const float* bias; float* Y; float* stat_part;
    int B, M, T, Mp, Tp, G, tiles_per_item;
    int st_chunk;               // stages (8 NG channels) per blockIdx.z
    float* part; unsigned* tile_ctr;
};

// 16 bytes per lane global -> LDS without a destination register (lane i lands at lds_addr + 16 i; lds_addr wave-uniform, in an SGPR).
// Inline asm on purpose: hipcc counts the builtin form as an LDS write and puts `s_waitcnt vmcnt(0)` in front of EVERY later ds_read —
// the prefetch issued at the top of a step was waited for before the step's own MFMAs (first version of these kernels: 1.9 us per step
// of 0.35 us of MFMA work).  The asm form is invisible to its bookkeeping; the waits are counted by hand below.
__device__ __forceinline__ void cx_dma16(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}
#define CX_NW 4                     // weight tiles in flight + 1: the tile of step i + 3 is requested during step i
template <int N>
__device__ __forceinline__ void cx_wait() {
    if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else static_assert(N == 6, "add the literal");
}
// NP planes per operand, NG channel groups (of 8) per stage: <3, 2> = fp32 operands cut into three bf16 planes (six MFMAs per 16
// channels and tap, 24 per wave and step); <1, 4> = bf16_run on operands rounded once (8 MFMAs per wave and step of 32 channels)
template <int NP, int NG>
__global__ __launch_bounds__(256, 2) void k_conv5_x3(ConvX3Args a) {
    // one carve: Ws[CX_NW][NP][NG][128] | Xs[2][NP][NG][136] (the epilogue's row sums reuse the front of it)
    constexpr int WSL = NP * NG * CX_BM, XSL = NP * NG * CX_XS;         // 16-byte slots per weight / activation tile
    constexpr int CARVE = CX_NW * WSL + 2 * XSL < 2112 ? 2112 : CX_NW * WSL + 2 * XSL;      // (>= 128 x 65 floats for the row sums)
    __shared__ uint4 lds_[CARVE];
    uint4 (*Ws)[NP][NG][CX_BM] = (uint4 (*)[NP][NG][CX_BM])&lds_[0];
    uint4 (*Xs)[NP][NG][CX_XS] = (uint4 (*)[NP][NG][CX_XS])&lds_[CX_NW * WSL];
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)&lds_[0];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int bb = blockIdx.x / a.tiles_per_item, t0 = (blockIdx.x - bb * a.tiles_per_item) * CX_BN;
    const int i0 = blockIdx.y * CX_BM;
    const int nst_all = a.G / NG;
    const int s0 = blockIdx.z * a.st_chunk, s1 = min(nst_all, s0 + a.st_chunk);
    // weight tile of (stage, tap): 2 NP NG pieces of 1 KB = NP planes x NG channel groups x 2 row halves; wave w issues pieces w, w + 4, ...
    // A request past the end of this workgroup's range re-reads its last tile into a buffer nobody reads any more: the number of
    // requests per step stays constant, which is what the counted waits below rely on
    constexpr int WPW = NP * NG / 2;        // weight requests per wave and step
    constexpr int XPW = (NP * NG * 3) / 4;  // activation requests per wave and stage, at least (waves 0, 1 may have one more)
    auto dma_w = [&](int st, int tap, int buf) {
        st = min(st, s1 - 1);
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            const int q = wave + 4 * i, p = q / (2 * NG), g = (q >> 1) % NG, half = q & 1;
            const uint4* src = a.Wp + ((size_t)(tap * NP + p) * a.G + NG * st + g) * a.Mp + i0 + 64 * half + lane;
            cx_dma16(src, lds0 + 16u * (unsigned)(((buf * NP + p) * NG + g) * CX_BM + 64 * half));
        }
    };
    // activation tile of a stage: NP planes x NG channel groups x (64 + 64 + 8 slots): 3 NP NG pieces, the short ones with 8 lanes
    auto dma_x = [&](int st, int buf) {
        st = min(st, s1 - 1);
#pragma unroll
        for (int i = 0; i < (3 * NP * NG + 3) / 4; ++i) {
            const int q = wave + 4 * i;
            if (q < 3 * NP * NG) {
                const int pg = q / 3, piece = q - 3 * pg, p = pg / NG, g = pg % NG;
                const uint4* src = a.Xp + (((size_t)bb * NP + p) * a.G + NG * st + g) * a.Tp + t0 + 64 * piece + lane;
                const unsigned dst = lds0 + 16u * (unsigned)(CX_NW * WSL + ((buf * NP + p) * NG + g) * CX_XS + 64 * piece);
                if (piece < 2 || lane < 8) cx_dma16(src, dst);
            }
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
    const int am = 64 * wm + (lane & 31), bn = 64 * wn + (lane & 31), kq = lane >> 5;
#define CX_MFMA(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const cx_bf16x8*)&(A_), *(const cx_bf16x8*)&(B_), C_, 0, 0, 0)
    // prologue: the activation tile of the first stage and the weight tiles of steps 0, 1, 2
    dma_x(s0, 0);
    dma_w(s0, 0, 0);
    dma_w(s0, 1, 1);
    dma_w(s0, 2, 2);
    cx_wait<2 * WPW>();         // everything but the tiles of steps 1 and 2
    __syncthreads();
    int idx = 0;
    for (int st = s0; st < s1; ++st) {
        const int xb = (st - s0) & 1;
#pragma unroll
        for (int tap = 0; tap < 5; ++tap, ++idx) {
            const int wb = idx & (CX_NW - 1);
            // requests of this step, in this order: the weight tile of step idx + 3 (into the buffer everybody left at the last barrier),
            // and at tap 0 the activation tile of the next stage
            if (tap < 2) dma_w(st, tap + 3, (idx + 3) & (CX_NW - 1));
            else dma_w(st + 1, tap - 2, (idx + 3) & (CX_NW - 1));
            if (tap == 0) dma_x(st + 1, xb ^ 1);
#define CX_ALL(PA, PB)                                      \
            CX_MFMA(av[PA][0], bv[PB][0], acc[0][0]); CX_MFMA(av[PA][0], bv[PB][1], acc[0][1]); \
            CX_MFMA(av[PA][1], bv[PB][0], acc[1][0]); CX_MFMA(av[PA][1], bv[PB][1], acc[1][1])
#pragma unroll
            for (int ks = 0; ks < NG / 2; ++ks) {       // blocks of 16 channels of this step
                uint4 av[NP][2], bv[NP][2];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    av[p][0] = Ws[wb][p][2 * ks + kq][am];
                    av[p][1] = Ws[wb][p][2 * ks + kq][am + 32];
                    bv[p][0] = Xs[xb][p][2 * ks + kq][bn + tap];
                    bv[p][1] = Xs[xb][p][2 * ks + kq][bn + 32 + tap];
                }
                if constexpr (NP == 3) { CX_ALL(2, 0); CX_ALL(0, 2); CX_ALL(1, 1); CX_ALL(1, 0); CX_ALL(0, 1); CX_ALL(0, 0); }
                else { CX_ALL(0, 0); }
            }
#undef CX_ALL
            // the weight tile of step idx + 1 (requested two steps ago) must have landed before anybody passes the barrier; requests
            // come back in order, so it has once at most the younger ones are outstanding: the weight requests of the last two steps
            // (WPW each per wave) and — until tap 3 — this stage's activation request (>= XPW per wave), which is younger as well
            if (tap < 3) cx_wait<2 * WPW + XPW>();
            else cx_wait<2 * WPW>();
            __syncthreads();    // ... and this step's LDS reads are done (the next step overwrites nothing that is still read: CX_NW = 4)
        }
    }
#undef CX_MFMA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the surplus requests of the last steps: the epilogue reuses the carve
    __syncthreads();
    const size_t tiles = (size_t)gridDim.x * gridDim.y, tile = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    if (gridDim.z > 1) {
        // channel split: raw accumulators to scratch in accumulator order (write-through), the workgroup that arrives last at its
        // tile's counter adds the partials in the fixed order z = 0, 1, ... and runs the epilogue (see gemm.hip)
        typedef unsigned cx_u32x4 __attribute__((ext_vector_type(4)));
        {
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.part + (blockIdx.z * tiles + tile) * (CX_BM * CX_BN), 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        cx_u32x4 v;
                        v.x = __float_as_uint(acc[x][y][4 * q]); v.y = __float_as_uint(acc[x][y][4 * q + 1]);
                        v.z = __float_as_uint(acc[x][y][4 * q + 2]); v.w = __float_as_uint(acc[x][y][4 * q + 3]);
                        __builtin_amdgcn_raw_buffer_store_b128(v, rs, ((((x * 2 + y) * 4 + q) * 256) + tid) * 16, 0, 16);
                    }
        }
        __shared__ unsigned last_;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* ctr = a.tile_ctr + tile;
        if (tid == 0) last_ = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.z - 1 ? 1u : 0u;
        __syncthreads();
        if (!last_) return;
        if (tid == 0) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int nz = (int)gridDim.z;
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
        for (int z0 = 0; z0 < nz; z0 += 2) {
            cx_u32x4 v[2][16];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int z = min(z0 + u, nz - 1);
                __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.part + (z * tiles + tile) * (CX_BM * CX_BN), 0, 0x7fffffff, 0x00020000);
#pragma unroll
                for (int g = 0; g < 16; ++g) v[u][g] = __builtin_amdgcn_raw_buffer_load_b128(rs, (g * 256 + tid) * 16, 0, 16);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (z0 + u < nz) {
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        acc[g >> 3][(g >> 2) & 1][4 * (g & 3)] += __uint_as_float(v[u][g].x);
                        acc[g >> 3][(g >> 2) & 1][4 * (g & 3) + 1] += __uint_as_float(v[u][g].y);
                        acc[g >> 3][(g >> 2) & 1][4 * (g & 3) + 2] += __uint_as_float(v[u][g].z);
                        acc[g >> 3][(g >> 2) & 1][4 * (g & 3) + 3] += __uint_as_float(v[u][g].w);
                    }
                }
        }
    }
    // epilogue: lane holds rows i0 + 64 wm + 32 x + (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of column t0 + 64 wn + 32 y + (lane & 31)
    float* red = (float*)&lds_[0];                   // [128 rows][65] floats = 33 KB of the carve: per-row partial sums of the 64 (wn, lane & 31) column slots
    constexpr int RS = 65;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {          // pass 0: sums (+ the stores), pass 1: sums of squares
        if (pass == 1 && !a.stat_part) break;
        __syncthreads();
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int il = 64 * wm + 32 * x + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), i = i0 + il;
                const float bv = (a.bias && i < a.M) ? a.bias[i] : 0.f;
                float s = 0.f;
#pragma unroll
                for (int y = 0; y < 2; ++y) {
                    const int t = t0 + 64 * wn + 32 * y + (lane & 31);
                    const float v = acc[x][y][r] + bv;
                    if (i < a.M && t < a.T) {
                        if (pass == 0) a.Y[((size_t)bb * a.M + i) * a.T + t] = v;
                        s += pass == 0 ? v : v * v;
                    }
                }
                if (a.stat_part) red[il * RS + 32 * wn + (lane & 31)] = s;
            }
        if (!a.stat_part) break;
        __syncthreads();
        if (tid < CX_BM && i0 + tid < a.M) {
            float s = 0.f;
#pragma unroll 8
            for (int c = 0; c < 64; ++c) s += red[tid * RS + c];
            a.stat_part[((size_t)blockIdx.x * a.M + i0 + tid) * 2 + pass] = s;
        }
    }
}

// scratch of the x3 convolutions (weight planes, activation planes, raw tiles of the channel split): slices of one ring, handed out
// in call order.  A step's sixteen convolutions take ~330 MB; a slice comes round again 1 GB later — long after the launches that
// used it have retired (convolutions of one stream run in order, a training step ends with a join of its streams).  A captured graph
// keeps the slices of its nodes.
static float* cx_scratch(size_t floats) {
    constexpr size_t RING = (size_t)256 << 20;       // floats (1 GB)
    static float* ring = nullptr;
    static std::atomic<size_t> pos{0};
    static std::atomic<int> state{0};
    floats = (floats + 63) & ~(size_t)63;
    if (floats > RING / 4) return nullptr;
    if (state.load(std::memory_order_acquire) != 2) {
        int expect = 0;
        if (state.compare_exchange_strong(expect, 1)) {
            float* p = nullptr;
            if (hipMalloc((void**)&p, RING * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); state.store(0); return nullptr; }
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

extern "C" int t2v_gemm_f32_set_mode(int x3);
// 0 = never, 1 = whenever the shape allows it, 2 = launches of >= 192 tiles (default; T2V_CONV_X3 presets it); -1 queries
static int g_cx3_mode = -1;
extern "C" int t2v_conv1d_x3_set_mode(int mode) {
    if (g_cx3_mode < 0) g_cx3_mode = getenv("T2V_CONV_X3") ? atoi(getenv("T2V_CONV_X3")) : 2;
    const int prev = g_cx3_mode;
    if (mode >= 0 && mode <= 2) g_cx3_mode = mode;
    return prev;
}
// does the fp32 k = 5 convolution (Cin -> Cout channels) take the x3 path?  (t2v_conv1d_stat_blocks must give the same answer)
bool t2v_conv5_x3_ok(int B, int Cin, int T, int Cout, int KS) {
    // t2v_conv1d_x3_set_mode / T2V_CONV_X3: 0 = never, 1 = whenever the shape allows it (tests, measurement), default 2: launches of >= 192 tiles only.  Measured on the
    // step's shapes (tools/dbg/conv_x3_time.py, T2V_CX3_DBG): at B = 6 the 512 -> 512 Postnet layer is 96 tiles; cut four ways over
    // its input channels the kernel's loop takes 45 us, but the two split passes (15 us), the raw tiles of the channel split going to
    // scratch and back and the epilogue add 46 us — 85-91 us against 80 for k_conv5_fwd<5>, and the fp32 step got 0.33 ms SLOWER with it
    // (11.32 vs 10.98 ms).  At B = 16 (256 tiles, no channel split) it runs the layer in 152 us against 212: that is where it is used.
    const int mode = t2v_conv1d_x3_set_mode(-1);
    if (!mode || KS != 5 || Cin % 16 || Cin < 64 || Cout < 64 || B < 1 || T < 1) return false;
    if (mode == 2 && (long)B * ((T + CX_BN - 1) / CX_BN) * ((Cout + CX_BM - 1) / CX_BM) < 192) return false;
    return t2v_gemm_f32_set_mode(-1) != 0;
}
int t2v_conv5_x3_stat_blocks(int B, int T) { return B * ((T + CX_BN - 1) / CX_BN); }

// channel splits: whole rounds of 768 resident workgroups (three per CU), a fixed cost per unit (prologue, epilogue, raw tile to scratch
// and back) — the cost model of gemm.hip's x3 kernel; at least two stages (32 channels) per split
static int cx_splits(long tiles, int nst) {
    static const int forced = getenv("T2V_CONV_X3_SPLITS") ? atoi(getenv("T2V_CONV_X3_SPLITS")) : 0;
    if (forced > 0) return forced > nst ? nst : forced;
    if (tiles >= 192) return 1;         // (measured at 256 tiles, one-plane form: 60 / 65 / 73 us with 1 / 2 / 4 splits)
    int best = 1;
    double best_w = 1e30;
    for (int ns = 1; ns <= 8 && (ns == 1 || nst / ns >= 2); ++ns) {
        const double rounds = (double)((tiles * ns + 767) / 768);
        const double w = rounds * (1.0 / ns + 0.08 * 30.0 / nst) + 0.08 * (ns - 1);
        if (w < best_w - 1e-9) { best_w = w; best = ns; }
    }
    return best;
}

// bf16_run: does the k = 5 convolution (t2v_conv1d_fwd_bf16 / the data gradient of t2v_conv1d_bwd_bf16) take the one-plane form of the
// kernels above (operands rounded to bf16 once by the split passes, LDS-DMA, 32 channels per step)?  The same switch as the fp32 form
// (t2v_conv1d_x3_set_mode: 0 never, 1 every eligible shape, 2 launches of >= 192 tiles); t2v_conv1d_stat_blocks_bf16 gives the
// matching answer
bool t2v_conv5_planes_bf16_ok(int B, int Cin, int T, int Cout, int KS) {
    const int mode = t2v_conv1d_x3_set_mode(-1);
    if (!mode || KS != 5 || Cin % 16 || Cin < 64 || Cout < 64 || B < 1 || T < 1) return false;
    if (mode == 2 && (long)B * ((T + CX_BN - 1) / CX_BN) * ((Cout + CX_BM - 1) / CX_BM) < 192) return false;
    return true;
}
// W: (M, Cin, 5) weights of the convolution to run (the data gradient passes the flipped, transposed weights); np = 3: fp32 operands
// as three bf16 planes, np = 1: bf16_run
int t2v_conv5_x3_run(const float* W, const float* X, const float* bias, float* Y, float* stat_part, int B, int Cin, int T, int M,
                     hipStream_t stream, int np) {
    const int NG = np == 3 ? 2 : 4;
    const int Mp = (M + CX_BM - 1) / CX_BM * CX_BM, G = (Cin + 8 * NG - 1) / (8 * NG) * NG, ntile = (T + CX_BN - 1) / CX_BN;
    const int Tp = (ntile * CX_BN + CX_XS - CX_BN + 63) / 64 * 64;          // the last tile's DMA reads slots up to ntile * 128 + 8
    const size_t w_slots = (size_t)5 * np * G * Mp, x_slots = (size_t)B * np * G * Tp;
    const long tiles = (long)B * ntile * (Mp / CX_BM);
    const int nst = G / NG, ns = cx_splits(tiles, nst);
    const size_t part_floats = ns > 1 ? (size_t)ns * tiles * CX_BM * CX_BN : 0;
    float* scr = cx_scratch(4 * (w_slots + x_slots) + part_floats);
    if (!scr) return T2V_ERR_LAUNCH;
    uint4* Wp = (uint4*)scr;
    uint4* Xp = Wp + w_slots;
    if (np == 3) {
        k_cx3_split_w<3><<<dim3((Mp + 255) / 256, G), 256, 0, stream>>>(W, Wp, M, Cin, Mp, G);
        k_cx3_split_x<3><<<dim3((Tp + 255) / 256, G, B), 256, 0, stream>>>(X, Xp, Cin, T, Tp, G);
    } else {
        k_cx3_split_w<1><<<dim3((Mp + 255) / 256, G), 256, 0, stream>>>(W, Wp, M, Cin, Mp, G);
        k_cx3_split_x<1><<<dim3((Tp + 255) / 256, G, B), 256, 0, stream>>>(X, Xp, Cin, T, Tp, G);
    }
    ConvX3Args a;
    a.Wp = Wp; a.Xp = Xp; a.bias = bias; a.Y = Y; a.stat_part = stat_part;
    a.B = B; a.M = M; a.T = T; a.Mp = Mp; a.Tp = Tp; a.G = G; a.tiles_per_item = ntile;
    a.st_chunk = nst; a.part = nullptr; a.tile_ctr = nullptr;
    dim3 grid(B * ntile, Mp / CX_BM, 1);
    if (ns > 1) {
        a.st_chunk = (nst + ns - 1) / ns;
        grid.z = (unsigned)((nst + a.st_chunk - 1) / a.st_chunk);
        if (grid.z > 1) {
            a.part = scr + 4 * (w_slots + x_slots);
            a.tile_ctr = t2v_arrival_counters((int)(grid.x * grid.y));
            if (!a.tile_ctr) return T2V_ERR_LAUNCH;
        } else {
            a.st_chunk = nst;
        }
    }
    if (np == 3) k_conv5_x3<3, 2><<<grid, 256, 0, stream>>>(a);
    else k_conv5_x3<1, 4><<<grid, 256, 0, stream>>>(a);
    return T2V_OK;
}
