// Column sums of a row-major (M, N) matrix: out[j] = sum_i A[i][j].  These are the bias gradients of every nn.Linear /
// LSTM cell on the path (reference model.py:171-190 encoder LSTM, model.py:200-236 decoder cells: autograd's
// `grad.sum(0)`) and the sum over decoder steps of the processed-memory gradient (model.py:60-64 memory_layer).
// Deterministic: the rows are cut into fixed slices, every slice is summed in a fixed order, the slices are added in
// ascending order — no atomics.
#include "t2v_common.h"
#include "t2v_kernels.h"
#include "t2v_coop.h"

#define CS_COLS 64      // columns per workgroup: one 256-byte row segment
#define CS_ROWL 16      // row lanes per workgroup

static int colsum_slices_(long M, long N) {
    const long cb = (N + CS_COLS - 1) / CS_COLS;
    long rs = (1024 + cb - 1) / cb;
    const long cap = (M + CS_ROWL - 1) / CS_ROWL;
    if (rs > cap) rs = cap;
    if (rs > 64) rs = 64;
    return rs < 1 ? 1 : (int)rs;
}

// grid = (ceil(N/64), RS), block = 256: thread = (column quad cq = tid & 15, row lane rl = tid >> 4)
// (round 4: with more than one row slice, the slice that arrives LAST at its column block's counter adds the gridDim.y
// partial rows in ascending order and writes the result — one launch, the same fixed-order sum as the former finishing launch)
template <bool VEC>
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ A, long lda, int M, int N, int rows_per_slice,
                                                float* __restrict__ dst, float* __restrict__ out, unsigned* __restrict__ ctr) {
    const int tid = threadIdx.x, cq = tid & 15, rl = tid >> 4;
    const int j0 = blockIdx.x * CS_COLS + 4 * cq;
    const int r0 = blockIdx.y * rows_per_slice, r1 = min(M, r0 + rows_per_slice);
    __shared__ float4 red[CS_ROWL][16];
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    if (VEC) {
        if (j0 < N) {
            const float* p = A + (size_t)j0;
            int i = r0 + rl;
            for (; i + 3 * CS_ROWL < r1; i += 4 * CS_ROWL) {      // four independent row loads in flight
                const float4 v0 = *(const float4*)(p + (size_t)i * lda);
                const float4 v1 = *(const float4*)(p + (size_t)(i + CS_ROWL) * lda);
                const float4 v2 = *(const float4*)(p + (size_t)(i + 2 * CS_ROWL) * lda);
                const float4 v3 = *(const float4*)(p + (size_t)(i + 3 * CS_ROWL) * lda);
                acc.x += (v0.x + v1.x) + (v2.x + v3.x); acc.y += (v0.y + v1.y) + (v2.y + v3.y);
                acc.z += (v0.z + v1.z) + (v2.z + v3.z); acc.w += (v0.w + v1.w) + (v2.w + v3.w);
            }
            for (; i < r1; i += CS_ROWL) {
                const float4 v = *(const float4*)(p + (size_t)i * lda);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
    } else {
        for (int i = r0 + rl; i < r1; i += CS_ROWL) {
            const float* p = A + (size_t)i * lda + j0;
            if (j0 + 0 < N) acc.x += p[0];
            if (j0 + 1 < N) acc.y += p[1];
            if (j0 + 2 < N) acc.z += p[2];
            if (j0 + 3 < N) acc.w += p[3];
        }
    }
    red[rl][cq] = acc;
    __syncthreads();
    if (tid < CS_COLS) {
        const int j = blockIdx.x * CS_COLS + tid;
        const float* rf = (const float*)&red[0][0];
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < CS_ROWL; ++r) s += rf[r * CS_COLS + tid];
        if (j < N) { if (ctr) st_sc1(dst + (size_t)blockIdx.y * N + j, s); else dst[(size_t)blockIdx.y * N + j] = s; }
    }
    if (!ctr) return;
    // (write-through partials + sc1 loads instead of a fence: see the split-K epilogue in gemm.hip)
    __shared__ unsigned last_;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) last_ = __hip_atomic_fetch_add(ctr + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.y - 1 ? 1u : 0u;
    __syncthreads();
    if (!last_) return;
    if (tid == 0) __hip_atomic_store(ctr + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < CS_COLS) {
        const int j = blockIdx.x * CS_COLS + tid;
        if (j < N) {
            float s = 0.f;
            const int RS = (int)gridDim.y;
            for (int r = 0; r < RS; ++r) s += ld_sc1(dst + (size_t)r * N + j);
            out[j] = s;
        }
    }
}

extern "C" long t2v_colsum_scratch_floats(long M, long N) {
    if (M < 1 || N < 1) return 0;
    const int rs = colsum_slices_(M, N);
    return rs > 1 ? (long)rs * N : 0;
}

extern "C" int t2v_colsum(const float* A, long lda, long M, long N, float* scratch, float* out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!A || !out || M < 1 || N < 1 || lda < N || M > 0x7fffffff || N > 0x7fffffff) return T2V_ERR_ARG;
    const int rs = colsum_slices_(M, N);
    if (rs > 1 && !scratch) return T2V_ERR_ARG;
    const int rows = (int)((M + rs - 1) / rs);
    const bool vec = !(N & 3) && !(lda & 3) && !((uintptr_t)A & 15);
    const dim3 grid((unsigned)((N + CS_COLS - 1) / CS_COLS), (unsigned)rs);
    float* dst = rs > 1 ? scratch : out;
    unsigned* ctr = nullptr;
    if (rs > 1) {
        if (grid.x > 4096) return T2V_ERR_ARG;
        ctr = t2v_arrival_counters((int)grid.x);
        if (!ctr) return T2V_ERR_LAUNCH;
    }
    if (vec) k_colsum<true><<<grid, 256, 0, stream>>>(A, lda, (int)M, (int)N, rows, dst, out, ctr);
    else k_colsum<false><<<grid, 256, 0, stream>>>(A, lda, (int)M, (int)N, rows, dst, out, ctr);
    return t2v_check_launch();
}
