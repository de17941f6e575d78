// Symbol embedding of the text encoder (reference model.py:474-482 nn.Embedding(80, 512), used at model.py:528 as
// `self.transcript_embedding(text).transpose(1, 2)`): the gather writes the (B, C, T) channel-major layout the first
// encoder convolution reads (no transpose copy), the backward is a deterministic per-symbol accumulation (one workgroup
// per symbol walks the B*T ids in ascending order: fixed summation order, no atomics).
#include "t2v_common.h"
#include "t2v_kernels.h"

// grid = (ceil(T/64), B), block = 256: thread = (position t = 64 bx + (tid & 63), channel group cg = tid >> 6)
__global__ __launch_bounds__(256) void k_embed_fwd(const long long* __restrict__ ids, const float* __restrict__ W,
                                                   float* __restrict__ out, int T, int C, int nsym) {
    const int b = blockIdx.y, t = blockIdx.x * 64 + (threadIdx.x & 63);
    if (t >= T) return;
    long long id = ids[(size_t)b * T + t];
    id = id < 0 ? 0 : (id >= nsym ? nsym - 1 : id);
    const float* w = W + (size_t)id * C;
    for (int c = threadIdx.x >> 6; c < C; c += 4) out[((size_t)b * C + c) * T + t] = w[c];      // coalesced along t
}

// grid = nsym, block = 256 (thread = channels tid, tid + 256, ..): dW[v][c] = sum over positions with id v of dy[b][c][t]
__global__ __launch_bounds__(256) void k_embed_bwd(const long long* __restrict__ ids, const float* __restrict__ dy,
                                                   float* __restrict__ dW, int B, int T, int C) {
    const int v = blockIdx.x;
    __shared__ int hits[1024];
    __shared__ int nhit;
    for (int c0 = 0; c0 < C; c0 += 512) {
        float acc0 = 0.f, acc1 = 0.f;
        for (int p0 = 0; p0 < B * T; p0 += 1024) {       // compact the matching positions of this chunk (ascending order)
            __syncthreads();
            if (threadIdx.x == 0) {
                int n = 0;
                const int hi = min(B * T, p0 + 1024);
                for (int p = p0; p < hi; ++p)
                    if (ids[p] == v) hits[n++] = p;
                nhit = n;
            }
            __syncthreads();
            for (int i = 0; i < nhit; ++i) {
                const int p = hits[i], b = p / T, t = p - b * T;
                const int c = c0 + threadIdx.x;
                if (c < C) acc0 += dy[((size_t)b * C + c) * T + t];
                if (c + 256 < C) acc1 += dy[((size_t)b * C + c + 256) * T + t];
            }
        }
        const int c = c0 + threadIdx.x;
        if (c < C) dW[(size_t)v * C + c] = acc0;
        if (c + 256 < C) dW[(size_t)v * C + c + 256] = acc1;
    }
}

extern "C" int t2v_embedding_fwd(const long long* ids, const float* W, float* out_bct, int B, int T, int C, int n_symbols,
                                 void* stream_) {
    if (!ids || !W || !out_bct || B < 1 || T < 1 || C < 1 || n_symbols < 1) return T2V_ERR_ARG;
    k_embed_fwd<<<dim3((T + 63) / 64, B), 256, 0, (hipStream_t)stream_>>>(ids, W, out_bct, T, C, n_symbols);
    return t2v_check_launch();
}

extern "C" int t2v_embedding_bwd(const long long* ids, const float* dy_bct, float* dW, int B, int T, int C, int n_symbols,
                                 void* stream_) {
    if (!ids || !dy_bct || !dW || B < 1 || T < 1 || C < 1 || n_symbols < 1) return T2V_ERR_ARG;
    k_embed_bwd<<<n_symbols, 256, 0, (hipStream_t)stream_>>>(ids, dy_bct, dW, B, T, C);
    return t2v_check_launch();
}
