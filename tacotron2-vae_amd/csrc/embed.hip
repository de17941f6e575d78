// Symbol embedding of the text encoder (reference model.py:474-482 nn.Embedding(80, 512), used at model.py:528 as
// `self.transcript_embedding(text).transpose(1, 2)`): the gather writes the (B, C, T) channel-major layout the first
// encoder convolution reads (no transpose copy), the backward is a deterministic per-symbol accumulation (one workgroup
// per symbol walks the B*T ids in ascending order: fixed summation order, no atomics).
#include "t2v_common.h"
#include "t2v_kernels.h"

// grid = (ceil(T/64), B, channel slices), block = 256: thread = (position t = 64 bx + (tid & 63), channel group cg = tid >> 6)
__global__ __launch_bounds__(256) void k_embed_fwd(const long long* __restrict__ ids, const float* __restrict__ W,
                                                   float* __restrict__ out, int T, int C, int nsym) {
    const int b = blockIdx.y, t = blockIdx.x * 64 + (threadIdx.x & 63);
    if (t >= T) return;
    long long id = ids[(size_t)b * T + t];
    id = id < 0 ? 0 : (id >= nsym ? nsym - 1 : id);
    const float* w = W + (size_t)id * C;
    // grid.z slices the channels (12 workgroups for the whole table took 21 us: one long serial loop per thread)
    const int cper = (C + gridDim.z - 1) / gridDim.z, c_lo = blockIdx.z * cper, c_hi = min(C, c_lo + cper);
    for (int c = c_lo + (threadIdx.x >> 6); c < c_hi; c += 4) out[((size_t)b * C + c) * T + t] = w[c];      // coalesced along t
}

// grid = nsym, block = 256 (thread = channels tid, tid + 256, ..): dW[v][c] = sum over positions with id v of dy[b][c][t].
// The positions holding symbol v are compacted 256 at a time in ascending order (wave ballot + prefix over the four
// waves), so the summation order is fixed.
__global__ __launch_bounds__(256) void k_embed_bwd(const long long* __restrict__ ids, const float* __restrict__ dy,
                                                   float* __restrict__ dW, int B, int T, int C) {
    const int v = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    __shared__ int hits[256];
    __shared__ int wcount[4];
    const int n = B * T;
    for (int c0 = 0; c0 < C; c0 += 512) {
        float acc0 = 0.f, acc1 = 0.f;
        const int c = c0 + tid;
        for (int p0 = 0; p0 < n; p0 += 256) {
            __syncthreads();                                  // hits / wcount of the previous chunk fully consumed
            const int p = p0 + tid;
            const bool hit = p < n && ids[p] == v;
            const unsigned long long m = __ballot(hit);
            if (lane == 0) wcount[wave] = __popcll(m);
            __syncthreads();
            int base = 0;
            for (int w = 0; w < wave; ++w) base += wcount[w];
            if (hit) hits[base + __popcll(m & ((1ull << lane) - 1ull))] = p;
            const int nhit = wcount[0] + wcount[1] + wcount[2] + wcount[3];
            __syncthreads();
            for (int i = 0; i < nhit; ++i) {
                const int q = hits[i], b = q / T, t = q - b * T;
                if (c < C) acc0 += dy[((size_t)b * C + c) * T + t];
                if (c + 256 < C) acc1 += dy[((size_t)b * C + c + 256) * T + t];
            }
        }
        if (c < C) dW[(size_t)v * C + c] = acc0;
        if (c + 256 < C) dW[(size_t)v * C + c + 256] = acc1;
    }
}

extern "C" int t2v_embedding_fwd(const long long* ids, const float* W, float* out_bct, int B, int T, int C, int n_symbols,
                                 void* stream_) {
    if (!ids || !W || !out_bct || B < 1 || T < 1 || C < 1 || n_symbols < 1) return T2V_ERR_ARG;
    k_embed_fwd<<<dim3((T + 63) / 64, B, C >= 64 ? 16 : 1), 256, 0, (hipStream_t)stream_>>>(ids, W, out_bct, T, C, n_symbols);
    return t2v_check_launch();
}

extern "C" int t2v_embedding_bwd(const long long* ids, const float* dy_bct, float* dW, int B, int T, int C, int n_symbols,
                                 void* stream_) {
    if (!ids || !dy_bct || !dW || B < 1 || T < 1 || C < 1 || n_symbols < 1) return T2V_ERR_ARG;
    k_embed_bwd<<<n_symbols, 256, 0, (hipStream_t)stream_>>>(ids, dy_bct, dW, B, T, C);
    return t2v_check_launch();
}
