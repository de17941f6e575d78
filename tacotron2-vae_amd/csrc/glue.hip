// Small fused epilogues that replace chains of one-line tensor ops on the critical path of a training step (round 4: the
// step was ~75 such launches, each 4-6 us inside a replayed graph):
//  * t2v_mask_outputs   — Tacotron2.parse_output (reference model.py:509-520): mel / post-net mel := 0, gate := 1e3 past each
//                         item's length; was arange + lt + bitwise_not + three masked_fill_ launches
//  * t2v_reparam_fwd/bwd — VAE_GST.reparameterize (reference modules.py:74-81): z = eps * exp(0.5 logvar) + mu and its
//                         gradient; was mul, exp, mul, add forward and five launches backward
//  * t2v_gather_words   — the error ledger (t2v_hip._err_note): up to 16 four-byte words copied in one launch
#include "t2v_common.h"
#include "t2v_kernels.h"

__global__ __launch_bounds__(256) void k_mask_outputs(float* __restrict__ mel, float* __restrict__ mel_post, float* __restrict__ gate,
                                                      const int* __restrict__ lengths, int C, int T, float gate_fill) {
    // grid = (rows of T per item: C mel + C post + 1 gate, B); only the tail past lengths[b] is touched
    const int b = blockIdx.y, r = blockIdx.x, len = lengths[b];
    if (len >= T) return;
    float* row; float v = 0.f;
    if (r < C) row = mel + ((size_t)b * C + r) * T;
    else if (r < 2 * C) row = mel_post + ((size_t)b * C + (r - C)) * T;
    else { row = gate + (size_t)b * T; v = gate_fill; }
    for (int t = max(len, 0) + threadIdx.x; t < T; t += 256) row[t] = v;
}
extern "C" int t2v_mask_outputs(float* mel, float* mel_post, float* gate, const int* lengths, int B, int C, int T, float gate_fill,
                                void* stream_) {
    if (!mel || !mel_post || !gate || !lengths || B < 1 || C < 1 || T < 1 || B > 65535) return T2V_ERR_ARG;
    k_mask_outputs<<<dim3((unsigned)(2 * C + 1), (unsigned)B), 256, 0, (hipStream_t)stream_>>>(mel, mel_post, gate, lengths, C, T, gate_fill);
    return t2v_check_launch();
}

__global__ void k_reparam_fwd(const float* __restrict__ eps, const float* __restrict__ mu, const float* __restrict__ logvar,
                              float* __restrict__ z, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) z[i] = eps[i] * expf(0.5f * logvar[i]) + mu[i];      // same operation order as eps * exp(0.5 * logvar) + mu
}
__global__ void k_reparam_bwd(const float* __restrict__ dz, const float* __restrict__ eps, const float* __restrict__ logvar,
                              float* __restrict__ dlogvar, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dlogvar[i] = ((dz[i] * eps[i]) * expf(0.5f * logvar[i])) * 0.5f;     // autograd's order: MulBackward, ExpBackward, MulBackward
}
extern "C" int t2v_reparam_fwd(const float* eps, const float* mu, const float* logvar, float* z, long n, void* stream_) {
    if (!eps || !mu || !logvar || !z || n < 1 || n > 0x7fffffff) return T2V_ERR_ARG;
    k_reparam_fwd<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream_>>>(eps, mu, logvar, z, (int)n);
    return t2v_check_launch();
}
extern "C" int t2v_reparam_bwd(const float* dz, const float* eps, const float* logvar, float* dlogvar, long n, void* stream_) {
    if (!dz || !eps || !logvar || !dlogvar || n < 1 || n > 0x7fffffff) return T2V_ERR_ARG;
    k_reparam_bwd<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream_>>>(dz, eps, logvar, dlogvar, (int)n);
    return t2v_check_launch();
}

struct GatherWords { const unsigned* src[16]; unsigned* dst[16]; int n; };
__global__ void k_gather_words(GatherWords g) {
    const int i = threadIdx.x;
    if (i < g.n) *g.dst[i] = *g.src[i];
}
extern "C" int t2v_gather_words(const void* const* src, void* const* dst, int n, void* stream_) {
    if (!src || !dst || n < 0) return T2V_ERR_ARG;
    for (int i0 = 0; i0 < n; i0 += 16) {
        GatherWords g;
        g.n = n - i0 < 16 ? n - i0 : 16;
        for (int i = 0; i < 16; ++i) {
            g.src[i] = i < g.n ? (const unsigned*)src[i0 + i] : nullptr;
            g.dst[i] = i < g.n ? (unsigned*)dst[i0 + i] : nullptr;
            if (i < g.n && (!g.src[i] || !g.dst[i])) return T2V_ERR_ARG;
        }
        k_gather_words<<<1, 64, 0, (hipStream_t)stream_>>>(g);
    }
    return t2v_check_launch();
}

// out[r] = [a[r][0..na) | b[r][0..nb)] for `rows` rows (row strides lda / ldb floats; na, nb, strides multiples of 4, 16-byte
// aligned): the (h_dec(t), context(t)) input rows of linear_projection / gate_layer (reference model.py:385-388 `torch.cat`)
// gathered from the decoder arena — a generic concatenation launch took 23 us for these 14.7 MB
__global__ __launch_bounds__(256) void k_concat2_rows(const float4* __restrict__ a, long lda4, int na4, const float4* __restrict__ b, long ldb4,
                                                      int nb4, float4* __restrict__ out, long rows) {
    const int w4 = na4 + nb4;
    const long n = rows * w4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / w4;
        const int c = (int)(i - r * w4);
        out[i] = c < na4 ? a[r * lda4 + c] : b[r * ldb4 + (c - na4)];
    }
}
extern "C" int t2v_concat2_rows(const float* a, long lda, int na, const float* b, long ldb, int nb, float* out, long rows, void* stream_) {
    if (!a || !b || !out || rows < 1 || na < 4 || nb < 4 || ((na | nb) & 3) || ((lda | ldb) & 3) ||
        (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15)) return T2V_ERR_ARG;
    const long n = rows * ((na + nb) / 4);
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    k_concat2_rows<<<(unsigned)blocks, 256, 0, (hipStream_t)stream_>>>((const float4*)a, lda / 4, na / 4, (const float4*)b, ldb / 4, nb / 4,
                                                                         (float4*)out, rows);
    return t2v_check_launch();
}
