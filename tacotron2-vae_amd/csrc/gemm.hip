// Generic fp32 GEMM on v_mfma_f32_32x32x2_f32 for the time-batched dense layers of the decoder
// (Prenet model.py:91-102, memory_layer model.py:37/290, the hoisted attention_rnn input term,
// linear_projection + gate_layer model.py:237-243 as one 81-column tile) and their gradients:
//      C[i][j] (+)= sum_k A(i,k) * B(j,k)  (+ bias[j]) (relu / dropout epilogue optional)
// A(i,k) = A[i*sAi + k*sAk], B(j,k) = B[j*sBj + k*sBk]: any of the NT / NN / TN forms by strides.
// 64x64x32 block tile, 256 threads (2x2 waves, one 32x32 accumulator each), operands staged k-major in
// LDS, next tile's global loads in flight during the MFMAs.  Threads run along whichever index of an
// operand is contiguous in memory so the staging loads are coalesced.
#include <stdlib.h>
#include <string.h>
#include "t2v_common.h"
#include "t2v_kernels.h"
#include "t2v_coop.h"

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
    int kz_chunk;          // split-K: k range per blockIdx.z (multiple of GM_BK), 0 = whole K in one block
    float* part;           // split-K partial tiles (gridDim.z, M, N) or NULL
    int nbatch;            // > 1: blockIdx.z = independent product (no split-K), operands advance by the batch strides
    long sAb, sBb, sCb;
    int nsub;              // > 1: product z covers k-chunk z % nsub of batch item z / nsub (A, B advance by sAs, sBs per chunk)
    long sAs, sBs;
    int bias_row;          // bias indexed by the output ROW (a convolution's output channel) instead of the column
    unsigned* tile_ctr;    // split-K: one arrival counter per output tile (zero before and after the launch)
};

// epilogue of the 64x64 kernels (fp32 and bf16 operands share the accumulator layout: wave (wm, wn) owns a 32x32 patch)
__device__ __forceinline__ void gemm64_epilogue(const GemmArgs& a, f32x16& acc, int i0, int j0, int wm, int wn, int lane, int tid) {
    const int j = j0 + 32 * wn + (lane & 31);
    if (a.part) {
        // split-K: the raw partial tile goes to scratch; the workgroup that arrives LAST at its tile's counter (round 4: no
        // second launch) adds the gridDim.z partials in the fixed order z = 0, 1, ... — the same sum, bit for bit, whichever
        // workgroup happens to do it — and applies the epilogue.  The counter is left at zero for the next launch.
        if (j < a.N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (i < a.M) st_sc1(a.part + ((size_t)blockIdx.z * a.M + i) * a.N + j, acc[r]);
            }
        }
        // (no __threadfence(): at agent scope it writes back and invalidates the XCD's whole L2 — with a dozen split-K and
        // column-sum launches per step that cost 1.3 ms of cache misses in the kernels running next to them.  The partials
        // are write-through stores (sc1) and are read back with sc1 loads, like the exchange rows of the persistent kernels;
        // the arrival is counted once this wave's stores have been acknowledged.)
        __shared__ unsigned last_;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned* ctr = a.tile_ctr + blockIdx.y * gridDim.x + blockIdx.x;
        if (tid == 0) last_ = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.z - 1 ? 1u : 0u;
        __syncthreads();
        if (!last_) return;
        if (tid == 0) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int nz = (int)gridDim.z;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (j < a.N) {
            // four partial tiles (64 loads per thread) are requested before the first add: one memory round trip per four
            // splits instead of one per split; the adds keep the order z = 0, 1, 2, ...
            for (int z0 = 0; z0 < nz; z0 += 4) {
                float v[4][16];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int z = min(z0 + u, nz - 1);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int i = min(i0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), a.M - 1);
                        v[u][r] = ld_sc1(a.part + ((size_t)z * a.M + i) * a.N + j);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (z0 + u < nz) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[r] += v[u][r];
                    }
            }
        }
    }
    if (j < a.N) {
        const float bv = (a.bias && !a.bias_row) ? a.bias[j] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (i < a.M) {
                const size_t idx = (size_t)i * a.ldc + j;
                float v = acc[r] + ((a.bias && a.bias_row) ? a.bias[i] : bv);
                if (a.accumulate) v += a.C[idx];
                if (a.relu) v = fmaxf(v, 0.f);
                if (a.p_drop > 0.f) v *= t2v_drop_scale(t2v_step_seed(a.seed, a.step), a.rng_stream, a.rng_t, (uint32_t)idx, a.p_drop);
                a.C[idx] = v;
            }
        }
    }
}

template <bool A_KC, bool B_KC>   // operand contiguous along k?
__global__ __launch_bounds__(256) void k_gemm_f32(GemmArgs a) {
    __shared__ float As[2][GM_BK][GM_BM + 1];
    __shared__ float Bs[2][GM_BK][GM_BN + 1];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int i0 = blockIdx.y * GM_BM, j0 = blockIdx.x * GM_BN;
    if (a.nbatch > 1) {
        const int zb = a.nsub > 1 ? blockIdx.z / a.nsub : blockIdx.z, zs = a.nsub > 1 ? blockIdx.z - zb * a.nsub : 0;
        a.A += zb * a.sAb + zs * a.sAs; a.B += zb * a.sBb + zs * a.sBs; a.C += blockIdx.z * a.sCb;
    }
    const int kbeg = a.kz_chunk ? blockIdx.z * a.kz_chunk : 0, kend = a.kz_chunk ? min(a.K, kbeg + a.kz_chunk) : a.K;
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
                ra[e8] = (i < a.M && k < kend) ? a.A[(long)i * a.sAi + (long)k * a.sAk] : 0.f;
            }
            {
                const int kk = B_KC ? (e & (GM_BK - 1)) : (e >> 6);
                const int jj = B_KC ? (e >> 5) : (e & (GM_BN - 1));
                const int j = j0 + jj, k = k0 + kk;
                rb[e8] = (j < a.N && k < kend) ? a.B[(long)j * a.sBj + (long)k * a.sBk] : 0.f;
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
    const int nkt = (kend - kbeg + GM_BK - 1) / GM_BK;
    load_tiles(kbeg);
    store_tiles(0);
    __syncthreads();
    const int ai = 32 * wm + (lane & 31), bj = 32 * wn + (lane & 31), kh = lane >> 5;
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) load_tiles(kbeg + (kt + 1) * GM_BK);
#pragma unroll
        for (int s = 0; s < GM_BK / 2; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[buf][2 * s + kh][ai], Bs[buf][2 * s + kh][bj], acc, 0, 0, 0);
        if (kt + 1 < nkt) store_tiles(buf ^ 1);
        __syncthreads();
    }
    gemm64_epilogue(a, acc, i0, j0, wm, wn, lane, tid);
}

// ---- split-K for the skinny deep-K products (weight gradients of the Prenet / projection / GRU / VAE head: a handful of
// 64x64 tiles with K = T*B = 2400): the k range is cut over gridDim.z, partial tiles are summed in a fixed order
// (deterministic) together with the bias / ReLU / dropout / accumulate epilogue.
// arrival counters of the split-K tiles (and of colsum.hip's column blocks): slices of one zero-initialised ring (every launch takes the next `tiles` words and
// leaves them at zero; a slice comes round again 64 K tiles later, long after its launch has retired; a captured graph keeps
// the slices of its nodes, and the launches of different graphs / eager steps of one engine never run at the same time)
#include <atomic>
#define GM_CTR_RING (1 << 16)
unsigned* t2v_arrival_counters(int tiles) {
    static unsigned* ring = nullptr;
    static std::atomic<unsigned> pos{0};
    static std::atomic<int> state{0};
    if (tiles > GM_CTR_RING) return nullptr;
    int st = state.load(std::memory_order_acquire);
    if (st != 2) {
        int expect = 0;
        if (state.compare_exchange_strong(expect, 1)) {
            unsigned* p = nullptr;
            if (hipMalloc((void**)&p, GM_CTR_RING * sizeof(unsigned)) != hipSuccess || hipMemset(p, 0, GM_CTR_RING * sizeof(unsigned)) != hipSuccess
                || hipDeviceSynchronize() != hipSuccess) { state.store(0); return nullptr; }
            ring = p;
            state.store(2, std::memory_order_release);
        } else {
            while (state.load(std::memory_order_acquire) == 1) { }
            if (state.load() != 2) return nullptr;
        }
    }
    unsigned at = pos.fetch_add((unsigned)tiles);
    at %= GM_CTR_RING;
    if (at + (unsigned)tiles > GM_CTR_RING) {         // would wrap inside the slice: start over at the top
        pos.store((unsigned)tiles);
        at = 0;
    }
    return ring + at;
}
// number of k-splits for the 64x64 kernel: deep-K products whose tiles leave CUs idle (r3 step trace: the 2400x256,
// K = 4096 data gradient of the Prenet ran 174 us on 152 workgroups, the 2400x81, K = 1536 projection 61 us on 76) are
// split until ~2 workgroups per CU are in flight, with >= 8 k-tiles left per split
static int gemm_splits(int M, int N, int K) {
    const long tiles = (long)((M + GM_BM - 1) / GM_BM) * ((N + GM_BN - 1) / GM_BN);
    // round 5: also launches of up to one workgroup per CU, from 4 k-tiles per split on — the Prenet weight gradient (256 tiles,
    // K = T*B), the BiLSTM weight gradients (128 tiles, K = B*T_in = 504): 11.34 -> 11.29 ms per fp32 step, two alternating pairs
    // (T2V_GEMM_SPLIT_POLICY=0: the round-4 rule below)
    static const int wide = getenv("T2V_GEMM_SPLIT_POLICY") ? atoi(getenv("T2V_GEMM_SPLIT_POLICY")) : 1;
    if (wide) {
        if (tiles > 256 || K < 8 * GM_BK) return 1;
        long ns2 = (512 + tiles - 1) / tiles;
        const long maxk2 = K / (4 * GM_BK);
        if (ns2 > maxk2) ns2 = maxk2;
        if (ns2 > 32) ns2 = 32;
        return ns2 < 2 ? 1 : (int)ns2;
    }
    if (tiles >= 256 || K < 16 * GM_BK) return 1;
    long ns = 512 / tiles;
    const long maxk = K / (8 * GM_BK);
    if (ns > maxk) ns = maxk;
    if (ns > 32) ns = 32;
    return ns < 2 ? 1 : (int)ns;
}
static bool gemm_x3_shape_ok(int M, int N, int K);
static long gemm_x3_scratch_floats(int M, int N, int K, int np);
static int gemm_x3_mode();
extern "C" long t2v_gemm_splitk_scratch_floats(int M, int N, int K) {
    if (M < 1 || N < 1 || K < 1) return 0;
    // round 6: the x3 path takes the large products — its scratch holds the bf16 planes of both operands (+ split-K partial tiles)
    if (gemm_x3_mode() && gemm_x3_shape_ok(M, N, K)) return gemm_x3_scratch_floats(M, N, K, 3);
    const int ns = gemm_splits(M, N, K);
    return ns > 1 ? (long)ns * M * N : 0;
}

// ---- large-tile variant for the big time-batched GEMMs (the deferred LSTM weight gradients DGA^T·X / DGD^T·X with
// K = T·B, the hoisted attention_rnn input term, the data gradients of the Prenet): 128x128x16 block tile, 256 threads =
// 2x2 waves, each wave a 64x64 patch = 2x2 accumulators of v_mfma_f32_32x32x2_f32 (one LDS read per operand per MFMA:
// 4 ds_read_b32 feed 4 MFMAs = 256 matrix-core cycles), 16-byte global loads from clamped addresses (no divergent
// branch around a load), register-staged double buffering with ONE barrier per k-tile.  LDS rows are 160 floats so
// that the two k-rows an MFMA reads sit in disjoint bank halves.
#define GB_BM 128
#define GB_BK 32
#define GB_LD 160
template <bool A_KC, bool B_KC, int BN>     // BN = 128 or 64 columns per block (the narrower tile evens out the last wave of tiles)
__global__ __launch_bounds__(256) void k_gemm_f32_big(GemmArgs a) {
    constexpr int NY = BN / 64;                      // 32-column accumulators per wave
    constexpr int NA = GB_BM * GB_BK / 4 / 256;      // float4 of the A tile per thread (4)
    constexpr int NB = BN * GB_BK / 4 / 256;         // float4 of the B tile per thread (2 or 4)
    constexpr int KQ = GB_BK / 4;                    // float4 per row of a k-contiguous operand (8)
    __shared__ __attribute__((aligned(16))) float As[2][GB_BK][GB_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][GB_BK][BN + 32];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so the tiles of ONE XCD
    // are made a contiguous run of the row-major tile order — a band of M-tile rows whose A rows (and the B columns that
    // sweep past them) are re-used inside that XCD's L2 instead of being fetched by all eight.  (Bijective for any tile
    // count: the first nb % 8 XCDs take one tile more.)
    int by_ = blockIdx.y, bx_ = blockIdx.x;
    {
        const int nb = gridDim.x * gridDim.y, bid = blockIdx.y * gridDim.x + blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
        const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        by_ = lin / gridDim.x;
        bx_ = lin - by_ * gridDim.x;
    }
    const int i0 = by_ * GB_BM, j0 = bx_ * BN;
    float4 ra[NA], rb[NB];
    // thread -> element of the operand tile: k-contiguous operand: (row = tid/KQ + (256/KQ) e, k4 = tid%KQ);
    //                                        row-contiguous operand: (k = tid/(rows/4) + .. e, row4 = tid%(rows/4))
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int e = 0; e < NA; ++e) {
            if (A_KC) {
                const int i = min(i0 + tid / KQ + (256 / KQ) * e, a.M - 1), k = min(k0 + 4 * (tid % KQ), a.K - 4);
                ra[e] = *(const float4*)(a.A + (long)i * a.sAi + k);
            } else {
                const int k = min(k0 + (tid >> 5) + 8 * e, a.K - 1), i = min(i0 + 4 * (tid & 31), a.M - 4);
                ra[e] = *(const float4*)(a.A + (long)k * a.sAk + i);
            }
        }
#pragma unroll
        for (int e = 0; e < NB; ++e) {
            if (B_KC) {
                const int j = min(j0 + tid / KQ + (256 / KQ) * e, a.N - 1), k = min(k0 + 4 * (tid % KQ), a.K - 4);
                rb[e] = *(const float4*)(a.B + (long)j * a.sBj + k);
            } else {
                constexpr int RQ = BN / 4;       // float4 per k-row
                const int k = min(k0 + tid / RQ + (256 / RQ) * e, a.K - 1), j = min(j0 + 4 * (tid % RQ), a.N - 4);
                rb[e] = *(const float4*)(a.B + (long)k * a.sBk + j);
            }
        }
    };
    // rows beyond K contribute zero: the clamped loads above fetched valid (finite) memory, the k mask zeroes it
    auto store_tiles = [&](int buf, int k0) {
#pragma unroll
        for (int e = 0; e < NA; ++e) {
            if (A_KC) {
                const int m = tid / KQ + (256 / KQ) * e, kk = 4 * (tid % KQ);
                const float z = k0 + kk < a.K ? 1.f : 0.f;        // K % 4 == 0: a float4 is all in or all out
                As[buf][kk + 0][m] = ra[e].x * z; As[buf][kk + 1][m] = ra[e].y * z;
                As[buf][kk + 2][m] = ra[e].z * z; As[buf][kk + 3][m] = ra[e].w * z;
            } else {
                const int kk = (tid >> 5) + 8 * e;
                const float z = k0 + kk < a.K ? 1.f : 0.f;
                *(float4*)&As[buf][kk][4 * (tid & 31)] = make_float4(ra[e].x * z, ra[e].y * z, ra[e].z * z, ra[e].w * z);
            }
        }
#pragma unroll
        for (int e = 0; e < NB; ++e) {
            if (B_KC) {
                const int n = tid / KQ + (256 / KQ) * e, kk = 4 * (tid % KQ);
                const float z = k0 + kk < a.K ? 1.f : 0.f;
                Bs[buf][kk + 0][n] = rb[e].x * z; Bs[buf][kk + 1][n] = rb[e].y * z;
                Bs[buf][kk + 2][n] = rb[e].z * z; Bs[buf][kk + 3][n] = rb[e].w * z;
            } else {
                constexpr int RQ = BN / 4;
                const int kk = tid / RQ + (256 / RQ) * e;
                const float z = k0 + kk < a.K ? 1.f : 0.f;
                *(float4*)&Bs[buf][kk][4 * (tid % RQ)] = make_float4(rb[e].x * z, rb[e].y * z, rb[e].z * z, rb[e].w * z);
            }
        }
    };
    f32x16 acc[2][NY];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < NY; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
    const int nkt = (a.K + GB_BK - 1) / GB_BK;
    load_tiles(0);
    store_tiles(0, 0);
    __syncthreads();
    const int am = 64 * wm + (lane & 31), bn = (BN / 2) * wn + (lane & 31), kh = lane >> 5;
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) load_tiles((kt + 1) * GB_BK);
        // LDS operands are requested half a k-tile ahead of their MFMAs (the compiler otherwise waits for each step's
        // reads right before its MFMAs: one LDS latency per 2-4 MFMAs); the MFMAs then drain the queue in order
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float av[GB_BK / 4][2], bw[GB_BK / 4][NY];
#pragma unroll
            for (int s = 0; s < GB_BK / 4; ++s) {
                const int kr = GB_BK / 2 * h + 2 * s + kh;
                av[s][0] = As[buf][kr][am];
                av[s][1] = As[buf][kr][am + 32];
#pragma unroll
                for (int y = 0; y < NY; ++y) bw[s][y] = Bs[buf][kr][bn + 32 * y];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < GB_BK / 4; ++s) {
#pragma unroll
                for (int y = 0; y < NY; ++y) {
                    acc[0][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][0], bw[s][y], acc[0][y], 0, 0, 0);
                    acc[1][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s][1], bw[s][y], acc[1][y], 0, 0, 0);
                }
            }
        }
        if (kt + 1 < nkt) store_tiles(buf ^ 1, (kt + 1) * GB_BK);
        __syncthreads();
    }
    const uint64_t seed = t2v_step_seed(a.seed, a.step);
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < NY; ++y) {
            const int j = j0 + (BN / 2) * wn + 32 * y + (lane & 31);
            if (j < a.N) {
                const float bvs = a.bias ? a.bias[j] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = i0 + 64 * wm + 32 * x + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (i < a.M) {
                        const size_t idx = (size_t)i * a.ldc + j;
                        float v = acc[x][y][r] + bvs;
                        if (a.accumulate) v += a.C[idx];
                        if (a.relu) v = fmaxf(v, 0.f);
                        if (a.p_drop > 0.f) v *= t2v_drop_scale(seed, a.rng_stream, a.rng_t, (uint32_t)idx, a.p_drop);
                        a.C[idx] = v;
                    }
                }
            }
        }
}

// rounds of resident tiles the chip needs (256 CUs, 2 blocks per CU) relative to the ideal: picks the column width
static double gemm_big_waste(int M, int N, int BN) {
    const double tiles = (double)((M + GB_BM - 1) / GB_BM) * ((N + BN - 1) / BN) * (BN / 64.0);   // in 128x64 units of work
    const double per_cu = tiles / 256.0;
    const double unit = BN / 64.0;
    return (double)((long)((per_cu + unit - 1e-9) / unit) + 0) * unit / per_cu;      // ceil to whole tiles per CU
}

// the large-tile kernel needs 16-byte-aligned float4 runs along each operand's contiguous index
static bool gemm_big_ok(const GemmArgs& a) {
    if (a.M < GB_BM || a.N < 128 || a.K < 4) return false;
    // the tiles must still fill the chip: mid-size GEMMs (Prenet, BiLSTM projections: < 256 tiles of 128x64) keep the 64x64 kernel
    if ((long)((a.M + GB_BM - 1) / GB_BM) * ((a.N + 63) / 64) < 256) return false;
    if (((uintptr_t)a.A | (uintptr_t)a.B) & 15) return false;
    const bool akc = a.sAk == 1, bkc = a.sBk == 1;
    if (akc ? ((a.K & 3) || (a.sAi & 3)) : (a.sAi != 1 || (a.M & 3) || (a.sAk & 3))) return false;
    if (bkc ? ((a.K & 3) || (a.sBj & 3)) : (a.sBj != 1 || (a.N & 3) || (a.sBk & 3))) return false;
    return true;
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
    // split-K (round 5, as in k_gemm_f32): blockIdx.z takes the k range [kbeg, kend)
    const int kbeg = a.kz_chunk ? blockIdx.z * a.kz_chunk : 0, kend = a.kz_chunk ? min(a.K, kbeg + a.kz_chunk) : a.K;
    // a k-contiguous operand whose rows start 16-byte aligned is read as two float4 per thread (round 5: the eight strided
    // scalar loads made the projection GEMM of the B = 16 step — 6400 x 81, K = 1536 — a 200 us launch)
    const bool a_vec = A_KC && a.sAk == 1 && !(a.sAi & 3) && !((uintptr_t)a.A & 15);
    const bool b_vec = B_KC && a.sBk == 1 && !(a.sBj & 3) && !((uintptr_t)a.B & 15);
    auto load_tiles = [&](int k0) {
        if (a_vec && k0 + a_k + 8 <= kend) {
            const float4 lo = *(const float4*)(a_p + k0 + a_k), hi = *(const float4*)(a_p + k0 + a_k + 4);
            const float z = a_rok ? 1.f : 0.f;
            ra[0] = lo.x * z; ra[1] = lo.y * z; ra[2] = lo.z * z; ra[3] = lo.w * z;
            ra[4] = hi.x * z; ra[5] = hi.y * z; ra[6] = hi.z * z; ra[7] = hi.w * z;
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ka = k0 + a_k + u;
                const float va = a_p[(long)min(ka, kend - 1) * a.sAk];
                ra[u] = (a_rok && ka < kend) ? va : 0.f;
            }
        }
        if (b_vec && k0 + b_k + 8 <= kend) {
            const float4 lo = *(const float4*)(b_p + k0 + b_k), hi = *(const float4*)(b_p + k0 + b_k + 4);
            const float z = b_rok ? 1.f : 0.f;
            rb[0] = lo.x * z; rb[1] = lo.y * z; rb[2] = lo.z * z; rb[3] = lo.w * z;
            rb[4] = hi.x * z; rb[5] = hi.y * z; rb[6] = hi.z * z; rb[7] = hi.w * z;
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int kb = k0 + b_k + u;
                const float vb = b_p[(long)min(kb, kend - 1) * a.sBk];
                rb[u] = (b_rok && kb < kend) ? vb : 0.f;
            }
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
    const int nkt = (kend - kbeg + GM_BK - 1) / GM_BK;
    load_tiles(kbeg);
    store_tiles(0);
    __syncthreads();
    const int ai = 32 * wm + (lane & 31), bj = 32 * wn + (lane & 31), kh = lane >> 5;
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) load_tiles(kbeg + (kt + 1) * GM_BK);
#pragma unroll
        for (int s4 = 0; s4 < GM_BK / 8; ++s4) {
            const s16x4 av = *(const s16x4*)&As[buf][ai][8 * s4 + 4 * kh];
            const s16x4 bv = *(const s16x4*)&Bs[buf][bj][8 * s4 + 4 * kh];
            acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(av, bv, acc, 0, 0, 0);
        }
        if (kt + 1 < nkt) store_tiles(buf ^ 1);
        __syncthreads();
    }
    gemm64_epilogue(a, acc, i0, j0, wm, wn, lane, tid);
}

// Backward of the relu / dropout epilogue: the stored output y already carries relu and the 1/(1-p) scaling, so
// ---- large-tile bf16 variant for the deferred LSTM weight gradients under bf16_run (DGA^T·X, DGD^T·X: K = T·B, both
// operands stored k-major, i.e. contiguous along their row index): 128x128x32 block tile, 2x2 waves x (2x2) accumulators
// of v_mfma_f32_32x32x16_bf16.  fp32 operands are rounded to bf16 (RNE) while they are staged; a thread stages 4 rows x
// 8 consecutive k of ONE operand (threads 0..127 the A tile, 128..255 the B tile: eight 16-byte global loads along the
// rows) and writes each row's eight bf16 as one 16-byte LDS word [k-group][row] — exactly what a lane reads back as an
// MFMA operand (conflict-free ds_read_b128), register-staged double buffering, one barrier per k-tile.
#define GBB_BM 128
#define GBB_BN 128
#define GBB_BK 32
// Round 5: an operand may also be k-contiguous (A_KC / B_KC: activations stored (rows, K) row-major — the Prenet output in
// front of the hoisted attention_rnn input term, the gate gradients in front of the Prenet data gradient); its staging
// threads then run along k (4 lanes cover 32 consecutive k of a row: 128 contiguous bytes) and read two float4 per row.
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void k_gemm_bf16_big_rr(GemmArgs a) {
    __shared__ uint4 As[2][GBB_BK / 8][GBB_BM];
    __shared__ uint4 Bs[2][GBB_BK / 8][GBB_BN];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int i0 = blockIdx.y * GBB_BM, j0 = blockIdx.x * GBB_BN;
    const bool isB = tid >= 128;
    const bool kc = isB ? B_KC : A_KC;                  // wave-uniform (waves 0, 1 stage A, waves 2, 3 stage B)
    const int t7 = tid & 127;
    const int r4 = kc ? t7 >> 2 : t7 & 31, kg = kc ? t7 & 3 : t7 >> 5;
    const float* P = isB ? a.B : a.A;
    const long sk = isB ? a.sBk : a.sAk, si = isB ? a.sBj : a.sAi;
    const int lim = isB ? a.N : a.M, row0 = min((isB ? j0 : i0) + 4 * r4, lim - 4);
    const float z_row = (isB ? j0 : i0) + 4 * r4 < lim ? 1.f : 0.f;      // M, N are multiples of 4: a row quad is all in or all out
    // split-K (round 5): blockIdx.z takes the k range [kbeg, kend) — the deferred LSTM weight gradients are 64 .. 384 tiles with
    // K = T*B = 6400, and a workgroup with ONE tile of prefetch is bound by the memory round trip per k-tile (200 k-tiles x
    // ~1.3 us = the 260-420 us these launches took), so the k range is cut until ~3 workgroups per CU are in flight
    const int kbeg = a.kz_chunk ? blockIdx.z * a.kz_chunk : 0, kend = a.kz_chunk ? min(a.K, kbeg + a.kz_chunk) : a.K;
    float4 rg[8];
    auto load_tiles = [&](int k0) {
        if (kc) {           // rows row0 .. row0 + 3, k = k0 + 8 kg .. + 7 (K and the split are multiples of 32 on this path)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const float* q = P + (long)(row0 + rr) * si + k0 + 8 * kg;
                rg[2 * rr] = *(const float4*)q;
                rg[2 * rr + 1] = *(const float4*)(q + 4);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) rg[u] = *(const float4*)(P + (long)min(k0 + 8 * kg + u, kend - 1) * sk + row0);
        }
    };
    auto store_tiles = [&](int buf, int k0) {
        uint4* dst = (isB ? &Bs[buf][kg][0] : &As[buf][kg][0]) + 4 * r4;
        if (kc) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const float4 lo = rg[2 * rr], hi = rg[2 * rr + 1];
                dst[rr] = make_uint4(gemm_pack_bf16x2(lo.x * z_row, lo.y * z_row), gemm_pack_bf16x2(lo.z * z_row, lo.w * z_row),
                                     gemm_pack_bf16x2(hi.x * z_row, hi.y * z_row), gemm_pack_bf16x2(hi.z * z_row, hi.w * z_row));
            }
            return;
        }
        float zk[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) zk[u] = (k0 + 8 * kg + u < kend ? 1.f : 0.f) * z_row;
        dst[0] = make_uint4(gemm_pack_bf16x2(rg[0].x * zk[0], rg[1].x * zk[1]), gemm_pack_bf16x2(rg[2].x * zk[2], rg[3].x * zk[3]),
                            gemm_pack_bf16x2(rg[4].x * zk[4], rg[5].x * zk[5]), gemm_pack_bf16x2(rg[6].x * zk[6], rg[7].x * zk[7]));
        dst[1] = make_uint4(gemm_pack_bf16x2(rg[0].y * zk[0], rg[1].y * zk[1]), gemm_pack_bf16x2(rg[2].y * zk[2], rg[3].y * zk[3]),
                            gemm_pack_bf16x2(rg[4].y * zk[4], rg[5].y * zk[5]), gemm_pack_bf16x2(rg[6].y * zk[6], rg[7].y * zk[7]));
        dst[2] = make_uint4(gemm_pack_bf16x2(rg[0].z * zk[0], rg[1].z * zk[1]), gemm_pack_bf16x2(rg[2].z * zk[2], rg[3].z * zk[3]),
                            gemm_pack_bf16x2(rg[4].z * zk[4], rg[5].z * zk[5]), gemm_pack_bf16x2(rg[6].z * zk[6], rg[7].z * zk[7]));
        dst[3] = make_uint4(gemm_pack_bf16x2(rg[0].w * zk[0], rg[1].w * zk[1]), gemm_pack_bf16x2(rg[2].w * zk[2], rg[3].w * zk[3]),
                            gemm_pack_bf16x2(rg[4].w * zk[4], rg[5].w * zk[5]), gemm_pack_bf16x2(rg[6].w * zk[6], rg[7].w * zk[7]));
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;
    const int nkt = (kend - kbeg + GBB_BK - 1) / GBB_BK;
    load_tiles(kbeg);
    store_tiles(0, kbeg);
    __syncthreads();
    const int am = 64 * wm + (lane & 31), bn = 64 * wn + (lane & 31), kq = lane >> 5;
    typedef __bf16 gbb_bf16x8 __attribute__((ext_vector_type(8)));
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) load_tiles(kbeg + (kt + 1) * GBB_BK);
        uint4 av[2][2], bv[2][2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            av[s2][0] = As[buf][2 * s2 + kq][am];
            av[s2][1] = As[buf][2 * s2 + kq][am + 32];
            bv[s2][0] = Bs[buf][2 * s2 + kq][bn];
            bv[s2][1] = Bs[buf][2 * s2 + kq][bn + 32];
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const gbb_bf16x8*)&av[s2][x], *(const gbb_bf16x8*)&bv[s2][y], acc[x][y], 0, 0, 0);
        if (kt + 1 < nkt) store_tiles(buf ^ 1, kbeg + (kt + 1) * GBB_BK);
        __syncthreads();
    }
    if (a.part) {
        // the raw accumulators go to scratch in ACCUMULATOR order ([split][tile][16-byte group g = 0..15][thread]: a wave's
        // store / load is 1 KB contiguous), write-through; the workgroup that arrives LAST at its tile's counter adds the
        // partials in the fixed order z = 0, 1, ... (bit-identical whichever workgroup does it) and runs the epilogue
        typedef unsigned gbb_u32x4 __attribute__((ext_vector_type(4)));
        const size_t tiles = (size_t)gridDim.x * gridDim.y, tile = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        {
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.part + (blockIdx.z * tiles + tile) * (GBB_BM * GBB_BN), 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        gbb_u32x4 v;
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
        for (int z0 = 0; z0 < nz; z0 += 2) {            // two partial tiles (32 loads per thread) requested before the first add
            gbb_u32x4 v[2][16];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int z = min(z0 + u, nz - 1);
                __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.part + (z * tiles + tile) * (GBB_BM * GBB_BN), 0, 0x7fffffff, 0x00020000);
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
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int j = j0 + 64 * wn + 32 * y + (lane & 31);
            if (j < a.N) {
                const float bvs = a.bias ? a.bias[j] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = i0 + 64 * wm + 32 * x + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (i < a.M) {
                        const size_t idx = (size_t)i * a.ldc + j;
                        float v = acc[x][y][r] + bvs;
                        if (a.accumulate) v += a.C[idx];
                        a.C[idx] = v;
                    }
                }
            }
        }
}
// an operand is staged by 16-byte loads either along its rows (row-contiguous: stride 1 between rows, a multiple of 4 between k)
// or along k (k-contiguous: the other way round, and K a multiple of the k-tile)
static bool gemm_bf16_big_operand_ok(long s_row, long s_k, int K, bool* kc) {
    if (s_row == 1 && !(s_k & 3)) { *kc = false; return true; }
    if (s_k == 1 && !(s_row & 3) && !(K % GBB_BK)) { *kc = true; return true; }
    return false;
}
static bool gemm_bf16_big_ok(const GemmArgs& a, bool* akc, bool* bkc) {
    if (a.relu || a.p_drop > 0.f) return false;
    if (!gemm_bf16_big_operand_ok(a.sAi, a.sAk, a.K, akc) || !gemm_bf16_big_operand_ok(a.sBj, a.sBk, a.K, bkc)) return false;
    if (a.M < GBB_BM || a.N < GBB_BN || (a.M & 3) || (a.N & 3)) return false;
    if (((uintptr_t)a.A | (uintptr_t)a.B) & 15) return false;
    // (round 5: from 64 tiles on — the 64x64 kernel it would fall to runs these shapes at 20-35 TFLOP/s)
    return (long)((a.M + GBB_BM - 1) / GBB_BM) * ((a.N + GBB_BN - 1) / GBB_BN) >= 64;
}
// k-splits of the 128x128 bf16 kernel: until ~3 workgroups per CU (its occupancy) are in flight, >= 16 k-tiles per split, <= 8
// partial tiles for the last workgroup of a tile to add
static int gemm_bf16_big_splits(int M, int N, int K) {
    const long tiles = (long)((M + GBB_BM - 1) / GBB_BM) * ((N + GBB_BN - 1) / GBB_BN);
    const int nkt = (K + GBB_BK - 1) / GBB_BK;
    if (tiles < 64 || tiles >= 512 || nkt < 32) return 1;
    static const int forced = getenv("T2V_GEMM_BF16_SPLITS") ? atoi(getenv("T2V_GEMM_BF16_SPLITS")) : 0;     // measurement
    long ns = forced > 0 ? forced : (768 + tiles - 1) / tiles;
    if (ns > nkt / 16) ns = nkt / 16;
    if (ns > 8) ns = 8;
    return ns < 2 ? 1 : (int)ns;
}
// (whether the 128x128 or the 64x64 kernel takes a product also depends on its strides: the scratch covers either)
static bool gemm_bf16_planes_ok(int M, int N, int K);
static long gemm_x3_scratch_floats(int M, int N, int K, int np);
extern "C" long t2v_gemm_bf16_splitk_scratch_floats(int M, int N, int K) {
    if (M < 1 || N < 1 || K < 1) return 0;
    if (gemm_bf16_planes_ok(M, N, K)) return gemm_x3_scratch_floats(M, N, K, 1);
    long big = 0;
    if (M >= GBB_BM && N >= GBB_BN && !(M & 3) && !(N & 3)) {
        const int ns = gemm_bf16_big_splits(M, N, K);
        const long tiles = (long)((M + GBB_BM - 1) / GBB_BM) * ((N + GBB_BN - 1) / GBB_BN);
        big = ns > 1 ? (long)ns * tiles * GBB_BM * GBB_BN : 0;
    }
    const int ns64 = gemm_splits(M, N, K);
    const long small = ns64 > 1 ? (long)ns64 * M * N : 0;
    return big > small ? big : small;
}

// ---- Round 6: fp32 GEMM on the bf16 matrix cores — "x3": every fp32 operand is cut, EXACTLY, into three bf16 values
//      a = a0 + a1 + a2   (a0 = RNE_bf16(a), a1 = RNE_bf16(a - a0), a2 = a - a0 - a1: three 8-bit significands = the 24 bits of a float;
//      both subtractions are exact in fp32, a2 is exactly representable in bf16)
// and the product is accumulated in fp32 from SIX bf16 MFMAs per k-block:  a0b0 + a0b1 + a1b0 + a1b1 + a0b2 + a2b0.
// The three terms left out (a1b2, a2b1, a2b2) are <= 2^-25 |a b| each (|a1| <= 2^-9 |a|, |a2| <= 2^-17 |a|, random sign) — below the
// rounding of ONE fp32 product (2^-24), so the result is fp32-class: the error against an fp64 reference is SMALLER than that of the
// fp32 MFMA kernel above (v_mfma_f32_32x32x2_f32 rounds the running sum once per k, i.e. K times; here it is rounded six times per
// SIXTEEN k) — measured 0.3x .. 0.8x, tests/test_gemm_gpu.py::test_x3_gemm_is_fp32_class holds both to the same bound.  Why: gfx950 has
// no TF32 path and its fp32 MFMA runs at the vector rate (157 TFLOP/s), bf16 MFMA at 16x that — six bf16 MFMAs per fp32 product block
// are 2.7x the fp32 matrix peak.  T2V_F32_GEMM=native (or t2v_gemm_f32_set_mode(0)) takes the fp32-MFMA kernels instead.
//
// Two launches per operand pair.  (1) k_x3_split: each operand ONCE — split, zero-padded to whole tiles and written K-major as the three
// planes [plane][k-group of 8][row] of 16-byte words (= a lane's MFMA operand), whatever its strides were.  A first version split
// inside the GEMM while staging: every 128x128 tile split its operand rows again (24x redundant at the LSTM weight-gradient shapes:
// 400 VALU-cycles per MFMA-cycle... 183 us of split + LDS stores next to 120 us of MFMA work) and reached 120 TFLOP/s; the split pass
// moves 10 B per element once.  (2) k_gemm_x3p: 128x128 tile, 256 threads = 2x2 waves x (2x2) accumulators of
// v_mfma_f32_32x32x16_bf16, stages of 16 k; the planes go global -> LDS by LDS-DMA (1 KB per wave instruction, no staging registers, no
// ds_write pass), double-buffered per stage: 48 KB of LDS and < 168 registers, so three workgroups share a CU and fill each other's
// barrier / DMA waits.
#define GX_BM 128
#define GX_BN 128
#define GX_SK 16                    // k per stage (two k-groups of 8)
typedef __bf16 gx_bf16x8 __attribute__((ext_vector_type(8)));
// 8 consecutive-k fp32 values of one operand row -> the row's 16-byte word in each of the three planes
__device__ __forceinline__ void gx_split8(const float (&v)[8], uint4& p0, uint4& p1, uint4& p2) {
    unsigned q0[4], q1[4], q2[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float x = v[2 * c], y = v[2 * c + 1];
        const unsigned h = gemm_pack_bf16x2(x, y);
        const float r1x = x - __uint_as_float(h << 16), r1y = y - __uint_as_float(h & 0xffff0000u);
        const unsigned m = gemm_pack_bf16x2(r1x, r1y);
        const float r2x = r1x - __uint_as_float(m << 16), r2y = r1y - __uint_as_float(m & 0xffff0000u);
        q0[c] = h; q1[c] = m; q2[c] = gemm_pack_bf16x2(r2x, r2y);
    }
    p0 = make_uint4(q0[0], q0[1], q0[2], q0[3]);
    p1 = make_uint4(q1[0], q1[1], q1[2], q1[3]);
    p2 = make_uint4(q2[0], q2[1], q2[2], q2[3]);
}
// planes of an operand with `rows` rows and K columns: Rp = rows rounded up to 128, G = k-groups rounded up to 4 (32 k);
// plane p, k-group g, row r -> 16-byte slot (p * G + g) * Rp + r.  Rows >= rows and k >= K are zero.
static inline long gx_rp(int rows) { return ((long)rows + 127) / 128 * 128; }
// np = 3: the x3 planes of an fp32 product (stages of 16 k); np = 1: ONE plane, the bf16-rounded operand of a bf16_run product (stages of 64 k)
static inline long gx_groups(int K, int np = 3) { return np == 3 ? ((long)K + 31) / 32 * 4 : ((long)K + 63) / 64 * 8; }      // (multiples of 4 / 8 k-groups)
static inline long gx_plane_slots(int rows, int K, int np = 3) { return np * gx_groups(K, np) * gx_rp(rows); }
// grid (Rp / 64, G / 4): a workgroup splits 64 rows x 32 k.  KC: the operand is contiguous along k (two float4 per row and k-group when
// aligned), else along its rows (or neither: scalar loads either way).  Every word crosses LDS once so that the plane stores run along
// the rows (1 KB contiguous per wave) whichever way the loads ran.
template <bool KC, int NP>
__global__ __launch_bounds__(256) void k_x3_split(const float* __restrict__ src, long s_row, long s_k, int rows, int K, uint4* __restrict__ dst,
                                                  long Rp, long G) {
    __shared__ uint4 sm[NP][4][64];
    const int tid = threadIdx.x;
    const int row0 = blockIdx.x * 64, g0 = blockIdx.y * 4;
    const int r = KC ? tid >> 2 : tid & 63, g = KC ? tid & 3 : tid >> 6;
    const int row = row0 + r, k0 = 8 * (g0 + g);
    float v[8];
    const bool rin = row < rows;
    const float* base = src + (long)min(row, rows - 1) * s_row;
    if (KC && s_k == 1 && !(s_row & 3) && !((uintptr_t)src & 15) && k0 + 8 <= K) {
        const float4 lo = *(const float4*)(base + k0), hi = *(const float4*)(base + k0 + 4);
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = rin ? v[u] : 0.f;
    } else {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float x = base[(long)min(k0 + u, K - 1) * s_k];
            v[u] = (rin && k0 + u < K) ? x : 0.f;
        }
    }
    if (NP == 3) {
        uint4 p0, p1, p2;
        gx_split8(v, p0, p1, p2);
        sm[0][g][r] = p0; sm[NP > 1 ? 1 : 0][g][r] = p1; sm[NP > 2 ? 2 : 0][g][r] = p2;
    } else {        // bf16_run: the operand rounded to bf16 (RNE), nothing else
        sm[0][g][r] = make_uint4(gemm_pack_bf16x2(v[0], v[1]), gemm_pack_bf16x2(v[2], v[3]), gemm_pack_bf16x2(v[4], v[5]), gemm_pack_bf16x2(v[6], v[7]));
    }
    __syncthreads();
    const int g2 = tid >> 6, r2 = tid & 63;
    if (row0 + r2 < Rp && g0 + g2 < G) {
#pragma unroll
        for (int p = 0; p < NP; ++p) dst[(p * G + g0 + g2) * Rp + row0 + r2] = sm[p][g2][r2];
    }
}
// One launch may cover up to two products that share M and K (t2v_gemm_f32_grouped: the decoder's LSTM weight gradients, DGA^T·[x...]
// and DGD^T·[x...]): the tiles of product 1 follow those of product 0 in the linear tile order.  The columns of a product's result go
// to up to three destinations (the [prenet | h | ctx] column blocks of one gate-gradient product are the gradients of different tensors).
struct GemmX3Prod {
    const uint4* Ap; const uint4* Bp;      // planes (gx_plane_slots)
    long RpA, RpB;
    int N, tiles_x, tile0, nseg;           // columns, column tiles, first linear tile, destinations
    int seg_col[3]; float* segC[3]; int seg_ldc[3];     // destination s takes columns [seg_col[s], seg_col[s + 1]) (multiples of 128)
};
struct GemmX3Args {
    GemmX3Prod pr[2];
    int nprod, ntiles;
    long G;
    const float* bias;
    int M, relu, accumulate;
    float p_drop; uint64_t seed; uint32_t rng_stream, rng_t;
    const t2v_step_params* step;
    int st_chunk;           // split-K: stages per blockIdx.z (0 = all)
    float* part; unsigned* tile_ctr;
};
// 16 bytes per lane global -> LDS without a destination register (lane i lands at lds_addr + 16 i; lds_addr wave-uniform, in an SGPR).
// Inline asm on purpose: hipcc counts the builtin form as an LDS write and puts `s_waitcnt vmcnt(0)` in front of EVERY later ds_read —
// the prefetch issued at the top of a stage was waited for before the stage's own MFMAs.  The asm form is invisible to its
// bookkeeping; the waits are counted by hand below.
__device__ __forceinline__ void gx_dma16(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}
#ifndef GX_NB
#define GX_NB 2                     // stage buffers: the planes of stage i + GX_NB - 1 are requested during stage i.  Two (48 KB, three
                                    // workgroups per CU) and three (72 KB, two per CU) measure the same alone and in the step (10.82 .. 10.95 ms):
                                    // what bounds the kernel is what a CU can pull in, ~13 B/clk (24 KB per stage against 768 MFMA cycles per SIMD
                                    // = the measured 0.41 MFMA-busy), not the depth of the prefetch
#endif
#ifndef GX_LDS_PAD
#define GX_LDS_PAD 0                // unused 16-byte slots on top (measurement: how many workgroups / how much free LDS a CU keeps)
#endif
#ifndef GX_NG1
#define GX_NG1 4                   // k-groups per stage of the one-plane (bf16_run) form (2 / 4 / 8 measured: 432 / 453 / 423 TFLOP/s on 4096 x 2560 x 6400)
#endif
#define GX_STR2(x) #x
#define GX_STR(x) GX_STR2(x)
// wait until at most N of this wave's memory requests are outstanding (N a compile-time constant <= 63)
template <int N>
__device__ __forceinline__ void gx_wait_stage() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else static_assert(N == 0, "add the literal");
}
// NP planes per operand, NG k-groups (of 8) per stage: <3, 2> = the x3 form of an fp32 product (six MFMAs per k-block of 16, 24 per
// wave and stage); <1, GX_NG1> = a bf16_run product on pre-rounded operands (one MFMA per k-block of 16)
template <int NP, int NG>
__global__ __launch_bounds__(256, 2) void k_gemm_x3p(GemmX3Args a) {
    // [buffer][A | B][plane][k-group][row]: 24 / 32 KB per buffer
    __shared__ uint4 lds_[GX_NB * 2 * NP * NG * GX_BM + GX_LDS_PAD];
    uint4 (*Ls)[2][NP][NG][GX_BM] = (uint4 (*)[2][NP][NG][GX_BM])&lds_[0];
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)&lds_[0];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order (as in k_gemm_f32_big): the tiles of one XCD are a contiguous run of the linear (product, row-major) tile order
    int lin;
    {
        const int nb = a.ntiles, bid = blockIdx.x;
        const int q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const GemmX3Prod& P = a.pr[(a.nprod > 1 && lin >= a.pr[1].tile0) ? 1 : 0];
    const int by_ = (lin - P.tile0) / P.tiles_x, bx_ = (lin - P.tile0) - by_ * P.tiles_x;
    const int i0 = by_ * GX_BM, j0 = bx_ * GX_BN;
    const int nst_all = (int)(a.G / NG);
    const int st0 = a.st_chunk ? blockIdx.z * a.st_chunk : 0, st1 = a.st_chunk ? min(nst_all, st0 + a.st_chunk) : nst_all;
    // DMA plan of a stage: 4 NP NG pieces of 1 KB = {A, B} x NP planes x NG k-groups x 2 row halves; wave w issues pieces w, w + 4, ...
    // A request past the end re-reads the last stage into a buffer nobody reads any more: NP NG requests per wave and stage, always —
    // what the counted wait below relies on
    constexpr int NPIECE = NP * NG;         // per wave
    auto stage_dma = [&](int st, int buf) {
        st = min(st, st1 - 1);
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) {
            const int q = wave + 4 * i;
            const int op = q / (2 * NP * NG), rem = q - (2 * NP * NG) * op, p = rem / (2 * NG), g = (rem >> 1) % NG, half = rem & 1;
            const uint4* src = op ? P.Bp + (p * a.G + NG * st + g) * P.RpB + j0 + 64 * half + lane
                                  : P.Ap + (p * a.G + NG * st + g) * P.RpA + i0 + 64 * half + lane;
            gx_dma16(src, lds0 + 16u * (unsigned)(((((buf * 2 + op) * NP + p) * NG + g) * GX_BM) + 64 * half));
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
#define GX_MFMA(A_, B_, C_) C_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const gx_bf16x8*)&(A_), *(const gx_bf16x8*)&(B_), C_, 0, 0, 0)
#pragma unroll
    for (int d = 0; d < GX_NB - 1; ++d) stage_dma(st0 + d, d);
    gx_wait_stage<NPIECE * (GX_NB - 2)>();          // the first stage has landed, the later ones may still be on their way
    __syncthreads();
    int buf = 0;
    for (int st = st0; st < st1; ++st) {
        // the stage GX_NB - 1 ahead goes into the buffer everybody left at the last barrier
        const int nb2 = buf == 0 ? GX_NB - 1 : buf - 1;     // (buf + GX_NB - 1) % GX_NB
        stage_dma(st + GX_NB - 1, nb2);
#define GX_ALL(PA, PB)                                      \
        GX_MFMA(av[PA][0], bv[PB][0], acc[0][0]); GX_MFMA(av[PA][0], bv[PB][1], acc[0][1]); \
        GX_MFMA(av[PA][1], bv[PB][0], acc[1][0]); GX_MFMA(av[PA][1], bv[PB][1], acc[1][1])
#pragma unroll
        for (int ks = 0; ks < NG / 2; ++ks) {           // k-blocks of 16 of this stage
            uint4 av[NP][2], bv[NP][2];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                av[p][0] = Ls[buf][0][p][2 * ks + kq][am];
                av[p][1] = Ls[buf][0][p][2 * ks + kq][am + 32];
                bv[p][0] = Ls[buf][1][p][2 * ks + kq][bn];
                bv[p][1] = Ls[buf][1][p][2 * ks + kq][bn + 32];
            }
            if constexpr (NP == 3) {
                // small terms first; product-major, so that consecutive MFMAs go to different accumulators
                GX_ALL(2, 0); GX_ALL(0, 2); GX_ALL(1, 1); GX_ALL(1, 0); GX_ALL(0, 1); GX_ALL(0, 0);
            } else {
                GX_ALL(0, 0);
            }
        }
#undef GX_ALL
        // stage st + 1 must have landed before anybody passes the barrier: requests come back in order, so it has once only the
        // requests of the GX_NB - 2 stages behind it (six per wave and stage) are outstanding
        gx_wait_stage<NPIECE * (GX_NB - 2)>();
        __syncthreads();        // ... and this stage's LDS reads are done
        buf = buf == GX_NB - 1 ? 0 : buf + 1;
    }
#undef GX_MFMA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the surplus requests of the last stages
    if (a.part) {
        // split-K exactly as in k_gemm_bf16_big_rr: raw accumulators to scratch in accumulator order (write-through), the workgroup
        // that arrives last at its tile's counter adds the partials in the fixed order z = 0, 1, ... and runs the epilogue
        typedef unsigned gx_u32x4 __attribute__((ext_vector_type(4)));
        const size_t tiles = (size_t)a.ntiles, tile = (size_t)lin;
        {
            __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.part + (blockIdx.z * tiles + tile) * (GX_BM * GX_BN), 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        gx_u32x4 v;
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
            gx_u32x4 v[2][16];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int z = min(z0 + u, nz - 1);
                __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.part + (z * tiles + tile) * (GX_BM * GX_BN), 0, 0x7fffffff, 0x00020000);
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
    const uint64_t seed = t2v_step_seed(a.seed, a.step);
    // destination of this tile's columns (a tile never straddles two: their boundaries are multiples of 128)
    int sg = 0;
    if (P.nseg > 1 && j0 >= P.seg_col[1]) sg = 1;
    if (P.nseg > 2 && j0 >= P.seg_col[2]) sg = 2;
    float* const Cd = P.segC[sg];
    const int ldc = P.seg_ldc[sg], jc0 = j0 - P.seg_col[sg];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int jl = 64 * wn + 32 * y + (lane & 31), j = j0 + jl;
            if (j < P.N) {
                const float bvs = a.bias ? a.bias[j] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = i0 + 64 * wm + 32 * x + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (i < a.M) {
                        const size_t idx = (size_t)i * ldc + jc0 + jl;
                        float v = acc[x][y][r] + bvs;
                        if (a.accumulate) v += Cd[idx];
                        if (a.relu) v = fmaxf(v, 0.f);
                        if (a.p_drop > 0.f) v *= t2v_drop_scale(seed, a.rng_stream, a.rng_t, (uint32_t)idx, a.p_drop);
                        Cd[idx] = v;
                    }
                }
            }
        }
}
// 1 = the x3 kernel takes the large fp32 products (default), 0 = the fp32-MFMA kernels only
static int g_f32_gemm_x3 = -1;
static int gemm_x3_mode() {
    if (g_f32_gemm_x3 < 0) {
        const char* e = getenv("T2V_F32_GEMM");
        g_f32_gemm_x3 = (e && (!strcmp(e, "native") || !strcmp(e, "0"))) ? 0 : 1;
    }
    return g_f32_gemm_x3;
}
extern "C" int t2v_gemm_f32_set_mode(int x3) {
    const int prev = gemm_x3_mode();
    if (x3 >= 0) g_f32_gemm_x3 = x3 ? 1 : 0;
    return prev;
}
// the x3 path takes a product from 64 tiles of 128x128 on, whatever its strides (the split pass reads any layout)
static bool gemm_x3_shape_ok(int M, int N, int K) {
    if (M < GX_BM || N < GX_BN || K < 32) return false;
    static const int min_tiles = getenv("T2V_X3_MIN_TILES") ? atoi(getenv("T2V_X3_MIN_TILES")) : 64;
    return (long)((M + GX_BM - 1) / GX_BM) * ((N + GX_BN - 1) / GX_BN) >= min_tiles;
}
// k-splits of the x3 kernel.  Three of its workgroups share a CU and hide each other's barrier / DMA waits, so a launch wants whole
// rounds of 768 units; a unit costs its share of K plus a fixed part (prologue, epilogue, raw tile to scratch and back), a split a
// little on top.  cost(ns) = ceil(tiles * ns / 768) * (1 / ns + 0.08) + 0.03 (ns - 1) reproduces the measured order of ns = 1..4 on
// 128 / 256 / 384 / 640 tiles at K = 2400 (tools/dbg/x3_time.py with T2V_GEMM_X3_SPLITS: 165 96 83 71 | 171 122 113 136 | 201 152 173
// 164 | 269 257 271 291 us)
static int gemm_x3_splits(int M, int N, int K, int np = 3) {
    const long tiles = (long)((M + GX_BM - 1) / GX_BM) * ((N + GX_BN - 1) / GX_BN);
    const int nst = (int)(gx_groups(K, np) / (np == 3 ? 2 : GX_NG1));
    static const int forced = getenv("T2V_GEMM_X3_SPLITS") ? atoi(getenv("T2V_GEMM_X3_SPLITS")) : 0;     // measurement
    if (forced > 0) return forced > nst ? nst : forced;
    int best = 1;
    double best_w = 1e30;
    for (int ns = 1; ns <= 6 && (ns == 1 || nst / ns >= 16); ++ns) {
        const double rounds = (double)((tiles * ns + 767) / 768);
        const double w = rounds * (1.0 / ns + 0.08) + 0.03 * (ns - 1);
        if (w < best_w - 1e-9) { best_w = w; best = ns; }
    }
    return best;
}
// floats of caller scratch the x3 path needs for (M, N, K): the planes of both operands + the split-K partial tiles
static long gemm_x3_plane_floats(int M, int N, int K, int np = 3) { return 4 * (gx_plane_slots(M, K, np) + gx_plane_slots(N, K, np)); }
static long gemm_x3_scratch_floats(int M, int N, int K, int np) {
    const int ns = gemm_x3_splits(M, N, K, np);
    const long tiles = (long)((M + GX_BM - 1) / GX_BM) * ((N + GX_BN - 1) / GX_BN);
    return gemm_x3_plane_floats(M, N, K, np) + (ns > 1 ? (long)ns * tiles * GX_BM * GX_BN : 0);
}
// bf16_run: large products on pre-rounded bf16 planes + the LDS-DMA kernel (round 6; T2V_BF16_GEMM_PLANES=0: k_gemm_bf16_big_rr)
static bool gemm_bf16_planes_ok(int M, int N, int K) {
    static const int on = getenv("T2V_BF16_GEMM_PLANES") ? atoi(getenv("T2V_BF16_GEMM_PLANES")) : 1;
    return on && K >= 64 && gemm_x3_shape_ok(M, N, K);
}
static void gx_split_launch(const float* src, long s_row, long s_k, int rows, int K, uint4* planes, long Rp, long G, long row_off, hipStream_t stream,
                            int np = 3) {
    // (planes + row_off: the operand's rows start at slot row_off of every k-group — column blocks of one B operand gathered from
    //  different tensors; row_off is a multiple of 128 and the operand's padded rows end at or before Rp)
    const unsigned gx = (unsigned)((rows + 127) / 128 * 2);
    const dim3 grid(gx, (unsigned)(G / 4));
    if (np == 3) {
        if (s_k == 1) k_x3_split<true, 3><<<grid, 256, 0, stream>>>(src, s_row, s_k, rows, K, planes + row_off, Rp, G);
        else k_x3_split<false, 3><<<grid, 256, 0, stream>>>(src, s_row, s_k, rows, K, planes + row_off, Rp, G);
    } else {
        if (s_k == 1) k_x3_split<true, 1><<<grid, 256, 0, stream>>>(src, s_row, s_k, rows, K, planes + row_off, Rp, G);
        else k_x3_split<false, 1><<<grid, 256, 0, stream>>>(src, s_row, s_k, rows, K, planes + row_off, Rp, G);
    }
}
static int gemm_x3_run(const GemmArgs& g, float* scratch, hipStream_t stream, int np = 3) {
    const int M = g.M, N = g.N, K = g.K;
    uint4* Ap = (uint4*)scratch;
    uint4* Bp = Ap + gx_plane_slots(M, K, np);
    const long G = gx_groups(K, np), RpA = gx_rp(M), RpB = gx_rp(N);
    gx_split_launch(g.A, g.sAi, g.sAk, M, K, Ap, RpA, G, 0, stream, np);
    gx_split_launch(g.B, g.sBj, g.sBk, N, K, Bp, RpB, G, 0, stream, np);
    GemmX3Args a;
    a.nprod = 1;
    GemmX3Prod& P = a.pr[0];
    P.Ap = Ap; P.Bp = Bp; P.RpA = RpA; P.RpB = RpB; P.N = N; P.tiles_x = (N + GX_BN - 1) / GX_BN; P.tile0 = 0; P.nseg = 1;
    P.seg_col[0] = 0; P.segC[0] = g.C; P.seg_ldc[0] = g.ldc;
    a.pr[1] = a.pr[0];
    a.ntiles = P.tiles_x * ((M + GX_BM - 1) / GX_BM);
    a.G = G; a.bias = g.bias; a.M = M; a.relu = g.relu; a.accumulate = g.accumulate;
    a.p_drop = g.p_drop; a.seed = g.seed; a.rng_stream = g.rng_stream; a.rng_t = g.rng_t; a.step = g.step;
    a.st_chunk = 0; a.part = nullptr; a.tile_ctr = nullptr;
    dim3 gb(a.ntiles, 1, 1);
    const int ns = gemm_x3_splits(M, N, K, np);
    if (ns > 1) {
        const int nst = (int)(G / (np == 3 ? 2 : GX_NG1));
        a.st_chunk = (nst + ns - 1) / ns;
        gb.z = (unsigned)((nst + a.st_chunk - 1) / a.st_chunk);         // no empty split
        a.part = scratch + gemm_x3_plane_floats(M, N, K, np);
        a.tile_ctr = t2v_arrival_counters(a.ntiles);
        if (!a.tile_ctr) return T2V_ERR_LAUNCH;
        if (gb.z < 2) { a.st_chunk = 0; a.part = nullptr; a.tile_ctr = nullptr; gb.z = 1; }
    }
    if (np == 3) k_gemm_x3p<3, 2><<<gb, 256, 0, stream>>>(a);
    else k_gemm_x3p<1, GX_NG1><<<gb, 256, 0, stream>>>(a);
    return t2v_check_launch();
}

// ---- grouped form: for every group g and part p   C[g][p] (+)= A_g · B_{g,p}^T,   all A_g (M x K), B_{g,p} (N[g][p] x K).
// The decoder's LSTM weight gradients are two such groups (A = the gate gradients of a cell, parts = the column blocks of its input
// [prenet | h_att | ctx] / [h_att | ctx | h_dec]): five products, 1 024 tiles.  Issued one by one they split the gate gradients
// three and two times, and none of the five launches fills whole rounds of the chip (64 .. 384 tiles on 512 slots, two of them cut over
// k with their raw tiles going through scratch); as ONE launch every operand is split once and the tiles make two full rounds.
static long gemm_grouped_scratch(const t2v_gemm_group* gr, int ngroups, int M, int K, int np) {
    if (!gr || ngroups < 1 || ngroups > 2 || M < 1 || K < 1) return 0;
    long slots = 0;
    for (int g = 0; g < ngroups; ++g) {
        long n = 0;
        for (int p = 0; p < gr[g].nb; ++p) n += gx_rp(gr[g].N[p]);
        slots += gx_plane_slots(M, K, np) + np * gx_groups(K, np) * n;
    }
    return 4 * slots;
}
extern "C" long t2v_gemm_f32_grouped_scratch_floats(const t2v_gemm_group* gr, int ngroups, int M, int K) {
    return gemm_grouped_scratch(gr, ngroups, M, K, 3);
}
extern "C" long t2v_gemm_bf16_grouped_scratch_floats(const t2v_gemm_group* gr, int ngroups, int M, int K) {
    return gemm_grouped_scratch(gr, ngroups, M, K, 1);
}
static bool gemm_grouped_x3_ok(const t2v_gemm_group* gr, int ngroups, int M, int K, int np) {
    if ((np == 3 && !gemm_x3_mode()) || M < GX_BM || K < 32) return false;
    long tiles = 0;
    for (int g = 0; g < ngroups; ++g)
        for (int p = 0; p < gr[g].nb; ++p) {
            if (gr[g].N[p] < GX_BN || (gr[g].N[p] % GX_BN)) return false;
            tiles += (long)(gr[g].N[p] / GX_BN) * ((M + GX_BM - 1) / GX_BM);
        }
    return tiles >= 64;
}
static int gemm_grouped_impl(const t2v_gemm_group* gr, int ngroups, int M, int K, int accumulate, float* scratch, void* stream_, int np) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!gr || ngroups < 1 || ngroups > 2 || M < 1 || K < 1) return T2V_ERR_ARG;
    for (int g = 0; g < ngroups; ++g) {
        if (!gr[g].A || gr[g].nb < 1 || gr[g].nb > 3) return T2V_ERR_ARG;
        for (int p = 0; p < gr[g].nb; ++p)
            if (!gr[g].B[p] || !gr[g].C[p] || gr[g].N[p] < 1 || gr[g].ldc[p] < gr[g].N[p]) return T2V_ERR_ARG;
    }
    if (!scratch || ((uintptr_t)scratch & 15) || !gemm_grouped_x3_ok(gr, ngroups, M, K, np)) {
        // (fp32-MFMA mode, or column blocks that are not whole tiles: the products one by one)
        for (int g = 0; g < ngroups; ++g)
            for (int p = 0; p < gr[g].nb; ++p) {
                const int rc = (np == 3 ? t2v_gemm_f32 : t2v_gemm_bf16)(gr[g].A, gr[g].sAi, gr[g].sAk, gr[g].B[p], gr[g].sBj[p], gr[g].sBk[p], nullptr, gr[g].C[p], gr[g].ldc[p],
                                            M, gr[g].N[p], K, 0, accumulate, 0.f, 0, 0, 0, stream_);
                if (rc != T2V_OK) return rc;
            }
        return T2V_OK;
    }
    const long G = gx_groups(K, np), RpA = gx_rp(M);
    GemmX3Args a;
    a.nprod = ngroups;
    uint4* at = (uint4*)scratch;
    int tile0 = 0;
    for (int g = 0; g < ngroups; ++g) {
        GemmX3Prod& P = a.pr[g];
        long n = 0;
        for (int p = 0; p < gr[g].nb; ++p) n += gr[g].N[p];
        P.Ap = at; at += gx_plane_slots(M, K, np);
        P.Bp = at; at += np * G * n;
        P.RpA = RpA; P.RpB = n; P.N = (int)n; P.tiles_x = (int)(n / GX_BN); P.tile0 = tile0; P.nseg = gr[g].nb;
        gx_split_launch(gr[g].A, gr[g].sAi, gr[g].sAk, M, K, (uint4*)P.Ap, RpA, G, 0, stream, np);
        long col = 0;
        for (int p = 0; p < 3; ++p) {
            const int q = p < gr[g].nb ? p : gr[g].nb - 1;
            P.seg_col[p] = p < gr[g].nb ? (int)col : 0x7fffffff; P.segC[p] = gr[g].C[q]; P.seg_ldc[p] = gr[g].ldc[q];
            if (p < gr[g].nb) {
                gx_split_launch(gr[g].B[p], gr[g].sBj[p], gr[g].sBk[p], gr[g].N[p], K, (uint4*)P.Bp, n, G, col, stream, np);
                col += gr[g].N[p];
            }
        }
        tile0 += P.tiles_x * (int)((M + GX_BM - 1) / GX_BM);
    }
    if (ngroups == 1) a.pr[1] = a.pr[0];
    a.ntiles = tile0;
    a.G = G; a.bias = nullptr; a.M = M; a.relu = 0; a.accumulate = accumulate;
    a.p_drop = 0.f; a.seed = 0; a.rng_stream = 0; a.rng_t = 0; a.step = t2v_step_for(stream);
    a.st_chunk = 0; a.part = nullptr; a.tile_ctr = nullptr;
    if (np == 3) k_gemm_x3p<3, 2><<<dim3(a.ntiles, 1, 1), 256, 0, stream>>>(a);
    else k_gemm_x3p<1, GX_NG1><<<dim3(a.ntiles, 1, 1), 256, 0, stream>>>(a);
    return t2v_check_launch();
}
extern "C" int t2v_gemm_f32_grouped(const t2v_gemm_group* gr, int ngroups, int M, int K, int accumulate, float* scratch, void* stream_) {
    return gemm_grouped_impl(gr, ngroups, M, K, accumulate, scratch, stream_, 3);
}
// bf16_run: the same launch structure on ONE plane per operand — the operands rounded to bf16 (RNE) once by the split pass, fp32
// accumulation: the arithmetic of k_gemm_bf16_big_rr (which rounds while staging), half its bytes into the CU
extern "C" int t2v_gemm_bf16_grouped(const t2v_gemm_group* gr, int ngroups, int M, int K, int accumulate, float* scratch, void* stream_) {
    return gemm_grouped_impl(gr, ngroups, M, K, accumulate, scratch, stream_, 1);
}

// d(pre-activation) = dy * [y != 0] * scale (reference Prenet, model.py:96-99: F.dropout(F.relu(linear(x)), p=0.5)).
__global__ __launch_bounds__(256) void k_epilogue_bwd(const float4* __restrict__ dy, const float4* __restrict__ y,
                                                      float4* __restrict__ out, size_t n4, float scale) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 g = dy[i], v = y[i];
    float4 o;
    o.x = v.x != 0.f ? g.x * scale : 0.f; o.y = v.y != 0.f ? g.y * scale : 0.f;
    o.z = v.z != 0.f ? g.z * scale : 0.f; o.w = v.w != 0.f ? g.w * scale : 0.f;
    out[i] = o;
}
__global__ __launch_bounds__(256) void k_epilogue_bwd_tail(const float* __restrict__ dy, const float* __restrict__ y,
                                                           float* __restrict__ out, size_t lo, size_t n, float scale) {
    const size_t i = lo + threadIdx.x;
    if (i < n) out[i] = y[i] != 0.f ? dy[i] * scale : 0.f;
}

extern "C" int t2v_gemm_epilogue_bwd(const float* dy, const float* y, float* out, size_t n, float scale, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!dy || !y || !out || n < 1) return T2V_ERR_ARG;
    if (((uintptr_t)dy | (uintptr_t)y | (uintptr_t)out) & 15) return T2V_ERR_ARG;
    const size_t n4 = n / 4;
    if (n4) k_epilogue_bwd<<<(unsigned)((n4 + 255) / 256), 256, 0, stream>>>((const float4*)dy, (const float4*)y, (float4*)out, n4, scale);
    if (n & 3) k_epilogue_bwd_tail<<<1, 256, 0, stream>>>(dy, y, out, 4 * n4, n, scale);
    return t2v_check_launch();
}

extern "C" int t2v_gemm_f32(const float* A, long sAi, long sAk, const float* B, long sBj, long sBk, const float* bias,
                            float* C, int ldc, int M, int N, int K, int relu, int accumulate, float p_drop,
                            uint64_t seed, uint32_t rng_stream, uint32_t rng_t, void* stream_);
static int gemm_bf16_impl(const float* A, long sAi, long sAk, const float* B, long sBj, long sBk, const float* bias,
                          float* C, int ldc, int M, int N, int K, int relu, int accumulate, float p_drop,
                          uint64_t seed, uint32_t rng_stream, uint32_t rng_t, float* splitk_scratch, void* stream_);
extern "C" int t2v_gemm_bf16(const float* A, long sAi, long sAk, const float* B, long sBj, long sBk, const float* bias,
                             float* C, int ldc, int M, int N, int K, int relu, int accumulate, float p_drop,
                             uint64_t seed, uint32_t rng_stream, uint32_t rng_t, void* stream_) {
    return gemm_bf16_impl(A, sAi, sAk, B, sBj, sBk, bias, C, ldc, M, N, K, relu, accumulate, p_drop, seed, rng_stream, rng_t, nullptr, stream_);
}
extern "C" int t2v_gemm_bf16_splitk(const float* A, long sAi, long sAk, const float* B, long sBj, long sBk, const float* bias,
                                    float* C, int ldc, int M, int N, int K, int relu, int accumulate, float p_drop,
                                    uint64_t seed, uint32_t rng_stream, uint32_t rng_t, float* splitk_scratch, void* stream_) {
    return gemm_bf16_impl(A, sAi, sAk, B, sBj, sBk, bias, C, ldc, M, N, K, relu, accumulate, p_drop, seed, rng_stream, rng_t, splitk_scratch, stream_);
}
static int gemm_bf16_impl(const float* A, long sAi, long sAk, const float* B, long sBj, long sBk, const float* bias,
                          float* C, int ldc, int M, int N, int K, int relu, int accumulate, float p_drop,
                          uint64_t seed, uint32_t rng_stream, uint32_t rng_t, float* splitk_scratch, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!A || !B || !C || M < 1 || N < 1 || K < 1 || ldc < N) return T2V_ERR_ARG;
    GemmArgs a;
    a.A = A; a.B = B; a.bias = bias; a.C = C; a.sAi = sAi; a.sAk = sAk; a.sBj = sBj; a.sBk = sBk;
    a.M = M; a.N = N; a.K = K; a.ldc = ldc; a.relu = relu; a.accumulate = accumulate;
    a.p_drop = p_drop; a.seed = seed; a.rng_stream = rng_stream; a.rng_t = rng_t; a.step = t2v_step_for(stream);
    a.kz_chunk = 0; a.part = nullptr; a.nbatch = 1; a.sAb = a.sBb = a.sCb = 0; a.nsub = 1; a.sAs = a.sBs = 0; a.bias_row = 0; a.tile_ctr = nullptr;
    if (splitk_scratch && !((uintptr_t)splitk_scratch & 15) && gemm_bf16_planes_ok(M, N, K))
        return gemm_x3_run(a, splitk_scratch, stream, 1);
    bool big_akc = false, big_bkc = false;
    if (gemm_bf16_big_ok(a, &big_akc, &big_bkc)) {
        const int ns = splitk_scratch ? gemm_bf16_big_splits(M, N, K) : 1;
        dim3 gb((N + GBB_BN - 1) / GBB_BN, (M + GBB_BM - 1) / GBB_BM, ns);
        if (ns > 1) {
            const int tiles_k = (K + GBB_BK - 1) / GBB_BK;
            a.kz_chunk = ((tiles_k + ns - 1) / ns) * GBB_BK;
            gb.z = (unsigned)((K + a.kz_chunk - 1) / a.kz_chunk);      // no empty split
            a.part = splitk_scratch;
            a.tile_ctr = t2v_arrival_counters((int)(gb.x * gb.y));
            if (!a.tile_ctr) return T2V_ERR_LAUNCH;
            if (gb.z < 2) { a.kz_chunk = 0; a.part = nullptr; a.tile_ctr = nullptr; gb.z = 1; }
        }
        if (big_akc && big_bkc) k_gemm_bf16_big_rr<true, true><<<gb, 256, 0, stream>>>(a);
        else if (big_akc) k_gemm_bf16_big_rr<true, false><<<gb, 256, 0, stream>>>(a);
        else if (big_bkc) k_gemm_bf16_big_rr<false, true><<<gb, 256, 0, stream>>>(a);
        else k_gemm_bf16_big_rr<false, false><<<gb, 256, 0, stream>>>(a);
        return t2v_check_launch();
    }
    // large products in an operand form the 128x128 bf16 kernel does not take (the hoisted attention_rnn input term and the
    // Prenet data gradient at B = 16: 13.4 GFLOP each): the 64x64 bf16 kernel ran them at ~35 TFLOP/s (361 / 399 us), the
    // large-tile FP32 kernel does ~100 — bf16_run takes whichever is faster, fp32 operands lose no accuracy
    if (gemm_big_ok(a))
        return t2v_gemm_f32(A, sAi, sAk, B, sBj, sBk, bias, C, ldc, M, N, K, relu, accumulate, p_drop, seed, rng_stream, rng_t, stream_);
    // the 64x64 kernel: one tile of prefetch per workgroup, so a grid of less than one workgroup per CU with a deep K is bound
    // by the memory round trip per k-tile (the BiLSTM data gradients at B = 16: 168 workgroups, 32 k-tiles, 90-120 us for
    // 1.4 GFLOP) — split over k like the fp32 kernel
    const int ns = splitk_scratch ? gemm_splits(M, N, K) : 1;
    dim3 grid((N + GM_BN - 1) / GM_BN, (M + GM_BM - 1) / GM_BM, ns);
    if (ns > 1) {
        const int tiles_k = (K + GM_BK - 1) / GM_BK;
        a.kz_chunk = ((tiles_k + ns - 1) / ns) * GM_BK;
        grid.z = (unsigned)((K + a.kz_chunk - 1) / a.kz_chunk);
        a.part = splitk_scratch;
        a.tile_ctr = t2v_arrival_counters((int)(grid.x * grid.y));
        if (!a.tile_ctr) return T2V_ERR_LAUNCH;
        if (grid.z < 2) { a.kz_chunk = 0; a.part = nullptr; a.tile_ctr = nullptr; grid.z = 1; }
    }
    const bool akc = sAk == 1, bkc = sBk == 1;
    if (akc && bkc) k_gemm_bf16<true, true><<<grid, 256, 0, stream>>>(a);
    else if (akc) k_gemm_bf16<true, false><<<grid, 256, 0, stream>>>(a);
    else if (bkc) k_gemm_bf16<false, true><<<grid, 256, 0, stream>>>(a);
    else k_gemm_bf16<false, false><<<grid, 256, 0, stream>>>(a);
    return t2v_check_launch();
}

static int gemm_f32_impl(const float* A, long sAi, long sAk, const float* B, long sBj, long sBk, const float* bias,
                         float* C, int ldc, int M, int N, int K, int relu, int accumulate, float p_drop,
                         uint64_t seed, uint32_t rng_stream, uint32_t rng_t, float* splitk_scratch, void* stream_);
extern "C" int t2v_gemm_f32(const float* A, long sAi, long sAk, const float* B, long sBj, long sBk, const float* bias,
                            float* C, int ldc, int M, int N, int K, int relu, int accumulate, float p_drop,
                            uint64_t seed, uint32_t rng_stream, uint32_t rng_t, void* stream_) {
    return gemm_f32_impl(A, sAi, sAk, B, sBj, sBk, bias, C, ldc, M, N, K, relu, accumulate, p_drop, seed, rng_stream, rng_t,
                         nullptr, stream_);
}
extern "C" int t2v_gemm_f32_splitk(const float* A, long sAi, long sAk, const float* B, long sBj, long sBk, const float* bias,
                                   float* C, int ldc, int M, int N, int K, int relu, int accumulate, float p_drop,
                                   uint64_t seed, uint32_t rng_stream, uint32_t rng_t, float* splitk_scratch, void* stream_) {
    return gemm_f32_impl(A, sAi, sAk, B, sBj, sBk, bias, C, ldc, M, N, K, relu, accumulate, p_drop, seed, rng_stream, rng_t,
                         splitk_scratch, stream_);
}
static int gemm_f32_impl(const float* A, long sAi, long sAk, const float* B, long sBj, long sBk, const float* bias,
                         float* C, int ldc, int M, int N, int K, int relu, int accumulate, float p_drop,
                         uint64_t seed, uint32_t rng_stream, uint32_t rng_t, float* splitk_scratch, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!A || !B || !C || M < 1 || N < 1 || K < 1 || ldc < N) return T2V_ERR_ARG;
    GemmArgs a;
    a.A = A; a.B = B; a.bias = bias; a.C = C; a.sAi = sAi; a.sAk = sAk; a.sBj = sBj; a.sBk = sBk;
    a.M = M; a.N = N; a.K = K; a.ldc = ldc; a.relu = relu; a.accumulate = accumulate;
    a.p_drop = p_drop; a.seed = seed; a.rng_stream = rng_stream; a.rng_t = rng_t; a.step = t2v_step_for(stream);
    a.kz_chunk = 0; a.part = nullptr; a.nbatch = 1; a.sAb = a.sBb = a.sCb = 0; a.nsub = 1; a.sAs = a.sBs = 0; a.bias_row = 0; a.tile_ctr = nullptr;
    const bool akc = sAk == 1, bkc = sBk == 1;
    if (splitk_scratch && gemm_x3_mode() && gemm_x3_shape_ok(M, N, K) && !((uintptr_t)splitk_scratch & 15))
        return gemm_x3_run(a, splitk_scratch, stream);
    if (gemm_big_ok(a)) {
        // (T2V_GEMM_NARROW=1, measurement: always the 128x64 tile — twice the tiles, so a launch that shares CUs with long
        // small-grid kernels balances itself instead of waiting for its slowest single-tile workgroup)
        static const int narrow = getenv("T2V_GEMM_NARROW") ? atoi(getenv("T2V_GEMM_NARROW")) : 0;
        const bool wide = !narrow && gemm_big_waste(M, N, 128) <= gemm_big_waste(M, N, 64) + 1e-6;
        const int BN = wide ? 128 : 64;
        dim3 gb((N + BN - 1) / BN, (M + GB_BM - 1) / GB_BM);
#define T2V_BIG(AK, BK)                                                             \
        if (wide) k_gemm_f32_big<AK, BK, 128><<<gb, 256, 0, stream>>>(a);           \
        else k_gemm_f32_big<AK, BK, 64><<<gb, 256, 0, stream>>>(a)
        if (akc && bkc) { T2V_BIG(true, true); }
        else if (akc) { T2V_BIG(true, false); }
        else if (bkc) { T2V_BIG(false, true); }
        else { T2V_BIG(false, false); }
#undef T2V_BIG
        return t2v_check_launch();
    }
    const int ns = splitk_scratch ? gemm_splits(M, N, K) : 1;
    dim3 grid((N + GM_BN - 1) / GM_BN, (M + GM_BM - 1) / GM_BM, ns);
    if (ns > 1) {
        const int tiles_k = (K + GM_BK - 1) / GM_BK;
        a.kz_chunk = ((tiles_k + ns - 1) / ns) * GM_BK;
        a.part = splitk_scratch;
        a.tile_ctr = t2v_arrival_counters((int)(grid.x * grid.y));
        if (!a.tile_ctr) return T2V_ERR_LAUNCH;
    }
    if (akc && bkc) k_gemm_f32<true, true><<<grid, 256, 0, stream>>>(a);
    else if (akc) k_gemm_f32<true, false><<<grid, 256, 0, stream>>>(a);
    else if (bkc) k_gemm_f32<false, true><<<grid, 256, 0, stream>>>(a);
    else k_gemm_f32<false, false><<<grid, 256, 0, stream>>>(a);
    return t2v_check_launch();
}

// nbatch independent products C_z = A_z · B_z^T in ONE launch of the 64x64 kernel (blockIdx.z = z): the per-item
// d_memory[b] = alignments_b^T · d_ctx_b of the decoder's reverse pass (model.py:84-85 under autograd) were B launches
// of 16 workgroups each.
// internal form (refenc.hip: the stride-2 convolutions of the reference encoder as batched GEMMs): per-row bias, and
// nsub k-chunks per batch item, each an independent product with its own output block
int t2v_gemm_f32_batched_ex(const float* A, long sAb, long sAs, long sAi, long sAk, const float* B, long sBb, long sBs, long sBj, long sBk,
                            const float* bias_row, float* C, long sCb, int ldc, int nbatch, int nsub, int M, int N, int K, hipStream_t stream) {
    if (!A || !B || !C || nbatch < 1 || nsub < 1 || (long)nbatch * nsub > 65535 || M < 1 || N < 1 || K < 1 || ldc < N) return T2V_ERR_ARG;
    GemmArgs a;
    a.A = A; a.B = B; a.bias = bias_row; a.C = C; a.sAi = sAi; a.sAk = sAk; a.sBj = sBj; a.sBk = sBk;
    a.M = M; a.N = N; a.K = K; a.ldc = ldc; a.relu = 0; a.accumulate = 0;
    a.p_drop = 0.f; a.seed = 0; a.rng_stream = 0; a.rng_t = 0; a.step = t2v_step_for(stream);
    a.kz_chunk = 0; a.part = nullptr; a.nbatch = nbatch * nsub > 1 ? nbatch * nsub : 1; a.sAb = sAb; a.sBb = sBb; a.sCb = sCb;
    a.nsub = nsub; a.sAs = sAs; a.sBs = sBs; a.bias_row = bias_row ? 1 : 0; a.tile_ctr = nullptr;
    const bool akc = sAk == 1, bkc = sBk == 1;
    dim3 grid((N + GM_BN - 1) / GM_BN, (M + GM_BM - 1) / GM_BM, nbatch * nsub);
    if (akc && bkc) k_gemm_f32<true, true><<<grid, 256, 0, stream>>>(a);
    else if (akc) k_gemm_f32<true, false><<<grid, 256, 0, stream>>>(a);
    else if (bkc) k_gemm_f32<false, true><<<grid, 256, 0, stream>>>(a);
    else k_gemm_f32<false, false><<<grid, 256, 0, stream>>>(a);
    return t2v_check_launch();
}
extern "C" int t2v_gemm_f32_batched(const float* A, long sAb, long sAi, long sAk, const float* B, long sBb, long sBj, long sBk,
                                    float* C, long sCb, int ldc, int nbatch, int M, int N, int K, void* stream_) {
    return t2v_gemm_f32_batched_ex(A, sAb, 0, sAi, sAk, B, sBb, 0, sBj, sBk, nullptr, C, sCb, ldc, nbatch, 1, M, N, K, (hipStream_t)stream_);
}
