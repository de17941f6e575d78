// BatchNorm1d (+ tanh / ReLU / identity) (+ dropout) around the implicit-GEMM conv, forward and backward.
// One workgroup per channel: a channel's B*T values are contiguous runs of T floats, the statistics are
// block reductions in a fixed order (deterministic), nothing is atomically accumulated.
// Reference semantics: nn.BatchNorm1d in train mode = biased batch variance over (B,T) INCLUDING padded
// positions, eps 1e-5, running stats momentum 0.1 with unbiased variance;
// F.dropout(act(bn(conv(x))), 0.5, training) (model.py:143-148, 175-177).
#include "t2v_common.h"
#include "t2v_kernels.h"

enum { ACT_NONE = 0, ACT_TANH = 1, ACT_RELU = 2 };

__device__ __forceinline__ float block_sum_256(float v, float* scr) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scr[threadIdx.x >> 6] = v;
    __syncthreads();
    return (scr[0] + scr[1]) + (scr[2] + scr[3]);
}

struct BnFwdArgs {
    const float* y;          // conv output (B,M,T)
    const float* stat_part;  // (nblk,M,2) from the conv epilogue
    int nblk;
    const float* gamma; const float* beta;
    float* running_mean; float* running_var;
    float* mean_out; float* rstd_out;     // (M) saved for backward
    float* out;              // (B,M,T)
    int B, M, T, act, training;
    float p_drop, momentum, eps;
    uint64_t seed; uint32_t rng_stream, rng_t;
    const t2v_step_params* step;
    int chunk;               // gridDim.y > 1: elements of a channel per workgroup (blockIdx.y takes [y*chunk, (y+1)*chunk))
};

// Partial statistics of wide channels (round 5): the reference encoder's first two layers have 32 channels of B*H*W = 48 000
// (B = 6) .. 128 000 (B = 16) values; one workgroup per channel made k_bn_act_fwd a 46 .. 105 us launch on 32 CUs, on the chain
// that bounds the start of the decoder.  Workgroup (m, s) sums elements [s*chunk, (s+1)*chunk) of channel m; k_bn_act_fwd
// finalises them like a convolution's partials and applies the normalisation with the same (m, s) grid.
__global__ __launch_bounds__(256) void k_bn_stat_part(const float* __restrict__ y, float* __restrict__ part, int B, int M, int T, int chunk) {
    const int m = blockIdx.x, tid = threadIdx.x;
    const int nel = B * T, lo = blockIdx.y * chunk, hi = min(nel, lo + chunk);
    __shared__ float scr[4];
    float ls = 0.f, lq = 0.f;
    for (int i0 = lo; i0 < hi; i0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int i = i0 + tid + 256 * e, ic = min(i, hi - 1);
            const int b = ic / T, t = ic - b * T;
            v[e] = y[((size_t)b * M + m) * T + t];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (i0 + tid + 256 * e < hi) { ls += v[e]; lq = fmaf(v[e], v[e], lq); }
    }
    const float s = block_sum_256(ls, scr), q = block_sum_256(lq, scr);
    if (tid == 0) {
        part[((size_t)blockIdx.y * M + m) * 2] = s;
        part[((size_t)blockIdx.y * M + m) * 2 + 1] = q;
    }
}

#define BN_NE 12      // fast path: channels of up to 256*12 values are held in registers
#define BN_UN 8       // larger channels: loads in flight per thread and pass

__global__ __launch_bounds__(256) void k_bn_act_fwd(BnFwdArgs a) {
    const int m = blockIdx.x, tid = threadIdx.x;
    float mean, rstd;
    __shared__ float scr[4];
    if (a.training) {
        double s = 0.0, q = 0.0;
        if (a.stat_part) {
            // finalize the statistics: the conv epilogue's partials are fetched in parallel (one per thread) and
            // added by a fixed tree in double precision (a serial loop here cost ~0.5 us of load latency per partial)
            __shared__ double dscr[2][256];
            double ls = 0.0, lq = 0.0;
            for (int i = tid; i < a.nblk; i += 256) {
                ls += (double)a.stat_part[((size_t)i * a.M + m) * 2];
                lq += (double)a.stat_part[((size_t)i * a.M + m) * 2 + 1];
            }
            dscr[0][tid] = ls;
            dscr[1][tid] = lq;
            __syncthreads();
            for (int st = 128; st > 0; st >>= 1) {
                if (tid < st) { dscr[0][tid] += dscr[0][tid + st]; dscr[1][tid] += dscr[1][tid + st]; }
                __syncthreads();
            }
            s = dscr[0][0];
            q = dscr[1][0];
        } else {
            // no partials (direct-form conv2d of the reference encoder): reduce the channel here
            // (eight loads in flight per thread: one load per iteration made the 48 000-value channels of the reference
            // encoder's first layer a 190-deep chain of memory latencies — 86 us for one launch)
            float ls = 0.f, lq = 0.f;
            const int nel = a.B * a.T;
            for (int i0 = 0; i0 < nel; i0 += 256 * BN_UN) {
                float v[BN_UN];
#pragma unroll
                for (int e = 0; e < BN_UN; ++e) {
                    const int i = i0 + tid + 256 * e, ic = min(i, nel - 1);
                    const int b = ic / a.T, t = ic - b * a.T;
                    v[e] = a.y[((size_t)b * a.M + m) * a.T + t];
                }
#pragma unroll
                for (int e = 0; e < BN_UN; ++e)
                    if (i0 + tid + 256 * e < nel) { ls += v[e]; lq = fmaf(v[e], v[e], lq); }
            }
            s = (double)block_sum_256(ls, scr);
            q = (double)block_sum_256(lq, scr);
        }
        const double n = (double)a.B * a.T;
        const double mu = s / n;
        double var = q / n - mu * mu;
        if (var < 0.0) var = 0.0;
        mean = (float)mu;
        rstd = (float)(1.0 / sqrt(var + (double)a.eps));
        if (tid == 0 && blockIdx.y == 0) {
            a.mean_out[m] = mean;
            a.rstd_out[m] = rstd;
            a.running_mean[m] = (1.f - a.momentum) * a.running_mean[m] + a.momentum * mean;
            a.running_var[m] = (1.f - a.momentum) * a.running_var[m] + a.momentum * (float)(var * n / (n - 1.0));
        }
    } else {
        mean = a.running_mean[m];
        rstd = 1.0f / sqrtf(a.running_var[m] + a.eps);
    }
    const float g = a.gamma[m] * rstd, bt = a.beta[m] - mean * a.gamma[m] * rstd;
    if (gridDim.y > 1) {        // a slice of a wide channel
        const int nel = a.B * a.T, lo = blockIdx.y * a.chunk, hi = min(nel, lo + a.chunk);
        for (int i0 = lo; i0 < hi; i0 += 256 * BN_UN) {
            size_t off[BN_UN];
            float yv[BN_UN];
#pragma unroll
            for (int e = 0; e < BN_UN; ++e) {
                const int i = i0 + tid + 256 * e, ic = min(i, hi - 1);
                const int b = ic / a.T, t = ic - b * a.T;
                off[e] = ((size_t)b * a.M + m) * a.T + t;
                yv[e] = a.y[off[e]];
            }
#pragma unroll
            for (int e = 0; e < BN_UN; ++e) {
                if (i0 + tid + 256 * e < hi) {
                    float z = fmaf(yv[e], g, bt);
                    if (a.act == ACT_TANH) z = tanhf_(z);
                    else if (a.act == ACT_RELU) z = fmaxf(z, 0.f);
                    if (a.training && a.p_drop > 0.f) z *= t2v_drop_scale(t2v_step_seed(a.seed, a.step), a.rng_stream, a.rng_t, (uint32_t)off[e], a.p_drop);
                    a.out[off[e]] = z;
                }
            }
        }
        return;
    }
    const int n = a.B * a.T;
    if (n <= 256 * BN_NE) {
        // a channel is only a few thousand values: every thread requests ALL of its elements before touching any
        // (the per-element loop below pays one memory latency per element)
        size_t off[BN_NE];
        float yv[BN_NE];
#pragma unroll
        for (int e = 0; e < BN_NE; ++e) {
            const int i = tid + 256 * e, ic = min(i, n - 1);
            const int b = ic / a.T, t = ic - b * a.T;
            off[e] = ((size_t)b * a.M + m) * a.T + t;
            yv[e] = a.y[off[e]];
        }
#pragma unroll
        for (int e = 0; e < BN_NE; ++e) {
            if (tid + 256 * e < n) {
                float z = fmaf(yv[e], g, bt);
                if (a.act == ACT_TANH) z = tanhf_(z);
                else if (a.act == ACT_RELU) z = fmaxf(z, 0.f);
                if (a.training && a.p_drop > 0.f) z *= t2v_drop_scale(t2v_step_seed(a.seed, a.step), a.rng_stream, a.rng_t, (uint32_t)off[e], a.p_drop);
                a.out[off[e]] = z;
            }
        }
        return;
    }
    for (int i0 = 0; i0 < n; i0 += 256 * BN_UN) {
        size_t off[BN_UN];
        float yv[BN_UN];
#pragma unroll
        for (int e = 0; e < BN_UN; ++e) {
            const int i = i0 + tid + 256 * e, ic = min(i, n - 1);
            const int b = ic / a.T, t = ic - b * a.T;
            off[e] = ((size_t)b * a.M + m) * a.T + t;
            yv[e] = a.y[off[e]];
        }
#pragma unroll
        for (int e = 0; e < BN_UN; ++e) {
            if (i0 + tid + 256 * e < n) {
                float z = fmaf(yv[e], g, bt);
                if (a.act == ACT_TANH) z = tanhf_(z);
                else if (a.act == ACT_RELU) z = fmaxf(z, 0.f);
                if (a.training && a.p_drop > 0.f) z *= t2v_drop_scale(t2v_step_seed(a.seed, a.step), a.rng_stream, a.rng_t, (uint32_t)off[e], a.p_drop);
                a.out[off[e]] = z;
            }
        }
    }
}

struct BnBwdArgs {
    const float* y;          // conv output (B,M,T)
    const float* dout;       // grad wrt the block output (B,M,T)
    const float* mean; const float* rstd; const float* gamma; const float* beta;
    float* dy;               // grad wrt the conv output (B,M,T)
    float* dgamma; float* dbeta;   // (M)
    float* dconv_bias;             // (M) or NULL: receives zeros (see t2v_bn_act_bwd)
    int B, M, T, act;
    float p_drop;
    uint64_t seed; uint32_t rng_stream, rng_t;
    const t2v_step_params* step;
    int eval_mode;           // running statistics (constants): no batch-statistic terms, the conv bias gets a gradient
};

__device__ __forceinline__ float bn_dz_v(const BnBwdArgs& a, size_t idx, float yv, float d, float g, float bt, float& xhat, float mean, float rstd) {
    xhat = (yv - mean) * rstd;
    if (a.p_drop > 0.f) d *= t2v_drop_scale(t2v_step_seed(a.seed, a.step), a.rng_stream, a.rng_t, (uint32_t)idx, a.p_drop);
    const float z = fmaf(yv, g, bt);
    if (a.act == ACT_TANH) { const float th = tanhf_(z); d *= 1.0f - th * th; }
    else if (a.act == ACT_RELU) { d = z > 0.f ? d : 0.f; }
    return d;
}

__global__ __launch_bounds__(256) void k_bn_act_bwd(BnBwdArgs a) {
    __shared__ float scr[4];
    const int m = blockIdx.x, tid = threadIdx.x;
    const float mean = a.mean[m], rstd = a.rstd[m], gam = a.gamma[m];
    const float g = gam * rstd, bt = a.beta[m] - mean * gam * rstd;
    float s1 = 0.f, s2 = 0.f;
    const int nel = a.B * a.T;
    if (nel <= 256 * BN_NE) {
        // one pass over memory: y and dout of this thread's elements are requested up front, dz / xhat stay in
        // registers across the block reduction
        size_t off[BN_NE];
        float yv[BN_NE], dv[BN_NE];
#pragma unroll
        for (int e = 0; e < BN_NE; ++e) {
            const int i = tid + 256 * e, ic = min(i, nel - 1);
            const int b = ic / a.T, t = ic - b * a.T;
            off[e] = ((size_t)b * a.M + m) * a.T + t;
            yv[e] = a.y[off[e]];
            dv[e] = a.dout[off[e]];
        }
        float dzv[BN_NE], xh[BN_NE];
#pragma unroll
        for (int e = 0; e < BN_NE; ++e) {
            float d = dv[e];
            xh[e] = (yv[e] - mean) * rstd;
            if (a.p_drop > 0.f) d *= t2v_drop_scale(t2v_step_seed(a.seed, a.step), a.rng_stream, a.rng_t, (uint32_t)off[e], a.p_drop);
            const float z = fmaf(yv[e], g, bt);
            if (a.act == ACT_TANH) { const float th = tanhf_(z); d *= 1.0f - th * th; }
            else if (a.act == ACT_RELU) { d = z > 0.f ? d : 0.f; }
            if (tid + 256 * e >= nel) d = 0.f;
            dzv[e] = d;
            s1 += d;
            s2 = fmaf(d, xh[e], s2);
        }
        const float S1 = block_sum_256(s1, scr);
        const float S2 = block_sum_256(s2, scr);
        if (tid == 0) { a.dbeta[m] = S1; a.dgamma[m] = S2; if (a.dconv_bias) a.dconv_bias[m] = a.eval_mode ? g * S1 : 0.f; }
        const float n = (float)a.B * (float)a.T;
        const float m1 = a.eval_mode ? 0.f : S1 / n, m2 = a.eval_mode ? 0.f : S2 / n;
#pragma unroll
        for (int e = 0; e < BN_NE; ++e)
            if (tid + 256 * e < nel) a.dy[off[e]] = g * (dzv[e] - m1 - xh[e] * m2);
        return;
    }
    for (int i0 = 0; i0 < nel; i0 += 256 * BN_UN) {
        float yv[BN_UN], dv[BN_UN];
        size_t off[BN_UN];
#pragma unroll
        for (int e = 0; e < BN_UN; ++e) {
            const int i = i0 + tid + 256 * e, ic = min(i, nel - 1);
            const int b = ic / a.T, t = ic - b * a.T;
            off[e] = ((size_t)b * a.M + m) * a.T + t;
            yv[e] = a.y[off[e]];
            dv[e] = a.dout[off[e]];
        }
#pragma unroll
        for (int e = 0; e < BN_UN; ++e) {
            if (i0 + tid + 256 * e < nel) {
                float xhat;
                const float dz = bn_dz_v(a, off[e], yv[e], dv[e], g, bt, xhat, mean, rstd);
                s1 += dz;
                s2 = fmaf(dz, xhat, s2);
            }
        }
    }
    const float S1 = block_sum_256(s1, scr);
    const float S2 = block_sum_256(s2, scr);
    if (tid == 0) { a.dbeta[m] = S1; a.dgamma[m] = S2; if (a.dconv_bias) a.dconv_bias[m] = a.eval_mode ? g * S1 : 0.f; }
    const float n = (float)a.B * (float)a.T;
    const float m1 = a.eval_mode ? 0.f : S1 / n, m2 = a.eval_mode ? 0.f : S2 / n;
    for (int i0 = 0; i0 < nel; i0 += 256 * BN_UN) {
        float yv[BN_UN], dv[BN_UN];
        size_t off[BN_UN];
#pragma unroll
        for (int e = 0; e < BN_UN; ++e) {
            const int i = i0 + tid + 256 * e, ic = min(i, nel - 1);
            const int b = ic / a.T, t = ic - b * a.T;
            off[e] = ((size_t)b * a.M + m) * a.T + t;
            yv[e] = a.y[off[e]];
            dv[e] = a.dout[off[e]];
        }
#pragma unroll
        for (int e = 0; e < BN_UN; ++e) {
            if (i0 + tid + 256 * e < nel) {
                float xhat;
                const float dz = bn_dz_v(a, off[e], yv[e], dv[e], g, bt, xhat, mean, rstd);
                a.dy[off[e]] = g * (dz - m1 - xhat * m2);
            }
        }
    }
}

// Backward of few WIDE channels (round 6; the reference encoder's first layers: 32 channels of B*H*W = 48 000 / 12 000 values at
// B = 6, 128 000 / 32 000 at B = 16): one workgroup per channel made k_bn_act_bwd an 86 us launch on 32 CUs — twice — on the chain that
// bounds the tail of the step (171-410 us at B = 16).  Two launches over a (channel, slice) grid instead: k_bn_bwd_part sums dz and
// dz * xhat of its slice, k_bn_bwd_apply adds the slices' partials in the fixed order s = 0, 1, ... (every workgroup of a channel gets
// the same bits) and writes its slice of dy; slice 0 writes dgamma / dbeta.
__global__ __launch_bounds__(256) void k_bn_bwd_part(BnBwdArgs a, float* __restrict__ part, int chunk) {
    __shared__ float scr[4];
    const int m = blockIdx.x, tid = threadIdx.x;
    const float mean = a.mean[m], rstd = a.rstd[m], gam = a.gamma[m];
    const float g = gam * rstd, bt = a.beta[m] - mean * gam * rstd;
    const int nel = a.B * a.T, lo = blockIdx.y * chunk, hi = min(nel, lo + chunk);
    float s1 = 0.f, s2 = 0.f;
    for (int i0 = lo; i0 < hi; i0 += 256 * BN_UN) {
        float yv[BN_UN], dv[BN_UN];
        size_t off[BN_UN];
#pragma unroll
        for (int e = 0; e < BN_UN; ++e) {
            const int i = i0 + tid + 256 * e, ic = min(i, hi - 1);
            const int b = ic / a.T, t = ic - b * a.T;
            off[e] = ((size_t)b * a.M + m) * a.T + t;
            yv[e] = a.y[off[e]];
            dv[e] = a.dout[off[e]];
        }
#pragma unroll
        for (int e = 0; e < BN_UN; ++e) {
            if (i0 + tid + 256 * e < hi) {
                float xhat;
                const float dz = bn_dz_v(a, off[e], yv[e], dv[e], g, bt, xhat, mean, rstd);
                s1 += dz;
                s2 = fmaf(dz, xhat, s2);
            }
        }
    }
    const float S1 = block_sum_256(s1, scr);
    const float S2 = block_sum_256(s2, scr);
    if (tid == 0) {
        part[((size_t)blockIdx.y * a.M + m) * 2] = S1;
        part[((size_t)blockIdx.y * a.M + m) * 2 + 1] = S2;
    }
}
__global__ __launch_bounds__(256) void k_bn_bwd_apply(BnBwdArgs a, const float* __restrict__ part, int chunk) {
    const int m = blockIdx.x, tid = threadIdx.x;
    const float mean = a.mean[m], rstd = a.rstd[m], gam = a.gamma[m];
    const float g = gam * rstd, bt = a.beta[m] - mean * gam * rstd;
    const int nel = a.B * a.T, lo = blockIdx.y * chunk, hi = min(nel, lo + chunk);
    float S1 = 0.f, S2 = 0.f;
    for (int s = 0; s < (int)gridDim.y; ++s) {          // (uniform loads, <= 64 slices, the same order in every workgroup)
        S1 += part[((size_t)s * a.M + m) * 2];
        S2 += part[((size_t)s * a.M + m) * 2 + 1];
    }
    if (tid == 0 && blockIdx.y == 0) { a.dbeta[m] = S1; a.dgamma[m] = S2; if (a.dconv_bias) a.dconv_bias[m] = a.eval_mode ? g * S1 : 0.f; }
    const float n = (float)a.B * (float)a.T;
    const float m1 = a.eval_mode ? 0.f : S1 / n, m2 = a.eval_mode ? 0.f : S2 / n;
    for (int i0 = lo; i0 < hi; i0 += 256 * BN_UN) {
        float yv[BN_UN], dv[BN_UN];
        size_t off[BN_UN];
#pragma unroll
        for (int e = 0; e < BN_UN; ++e) {
            const int i = i0 + tid + 256 * e, ic = min(i, hi - 1);
            const int b = ic / a.T, t = ic - b * a.T;
            off[e] = ((size_t)b * a.M + m) * a.T + t;
            yv[e] = a.y[off[e]];
            dv[e] = a.dout[off[e]];
        }
#pragma unroll
        for (int e = 0; e < BN_UN; ++e) {
            if (i0 + tid + 256 * e < hi) {
                float xhat;
                const float dz = bn_dz_v(a, off[e], yv[e], dv[e], g, bt, xhat, mean, rstd);
                a.dy[off[e]] = g * (dz - m1 - xhat * m2);
            }
        }
    }
}
static float* bn_part_scratch(size_t floats);
// launches the backward of `a`: sliced for few wide channels, else one workgroup per channel
static void bn_bwd_launch(BnBwdArgs& a, hipStream_t stream) {
    static const int split_on = getenv("T2V_BN_SPLIT") ? atoi(getenv("T2V_BN_SPLIT")) : 1;
    const long nel = (long)a.B * a.T;
    if (split_on && a.M <= 128 && nel >= 8192) {
        long S = (nel + 2047) / 2048;                   // >= 2048 values per workgroup ...
        if (S * a.M > 1024) S = 1024 / a.M;             // ... and at most ~4 workgroups per CU
        if (S > 64) S = 64;
        if (S >= 2) {
            float* part = bn_part_scratch((size_t)S * a.M * 2);
            if (part) {
                const int chunk = (int)(((nel + S - 1) / S + 255) / 256 * 256);
                const int Sy = (int)((nel + chunk - 1) / chunk);
                k_bn_bwd_part<<<dim3(a.M, Sy), 256, 0, stream>>>(a, part, chunk);
                k_bn_bwd_apply<<<dim3(a.M, Sy), 256, 0, stream>>>(a, part, chunk);
                return;
            }
        }
    }
    k_bn_act_bwd<<<a.M, 256, 0, stream>>>(a);
}

// library-owned ring for the partial statistics above (a few hundred floats per launch; a slice comes round again 4 MB later —
// a captured graph keeps the slices of its nodes, and graphs / eager steps of one engine never run at the same time)
#include <atomic>
static float* bn_part_scratch(size_t floats) {
    constexpr size_t RING = (size_t)1 << 20;
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

extern "C" int t2v_bn_act_fwd(const float* y, const float* stat_part, int nblk, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float* mean_out, float* rstd_out, float* out,
                              int B, int M, int T, int act, int training, float p_drop, float momentum, float eps,
                              uint64_t seed, uint32_t rng_stream, uint32_t rng_t, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!y || !gamma || !beta || !running_mean || !running_var || !out || B < 1 || M < 1 || T < 1) return T2V_ERR_ARG;
    if (training && (!mean_out || !rstd_out || (long)B * T < 2)) return T2V_ERR_ARG;
    BnFwdArgs a;
    a.y = y; a.stat_part = stat_part; a.nblk = nblk; a.gamma = gamma; a.beta = beta;
    a.running_mean = running_mean; a.running_var = running_var; a.mean_out = mean_out; a.rstd_out = rstd_out;
    a.out = out; a.B = B; a.M = M; a.T = T; a.act = act; a.training = training; a.p_drop = p_drop;
    a.momentum = momentum; a.eps = eps; a.seed = seed; a.rng_stream = rng_stream; a.rng_t = rng_t; a.step = t2v_step_for(stream);
    a.chunk = 0;
    // few wide channels without partials from a convolution's epilogue (BatchNorm2d of the reference encoder): cut every channel
    // over gridDim.y (T2V_BN_SPLIT=0: one workgroup per channel as before)
    static const int split_on = getenv("T2V_BN_SPLIT") ? atoi(getenv("T2V_BN_SPLIT")) : 1;
    const long nel = (long)B * T;
    if (split_on && training && !stat_part && M <= 128 && nel >= 16384) {
        long S = (nel + 4095) / 4096;                   // >= 4096 values per workgroup ...
        if (S * M > 1024) S = 1024 / M;                 // ... and at most ~4 workgroups per CU
        if (S > 64) S = 64;
        if (S >= 2) {
            float* part = bn_part_scratch((size_t)S * M * 2);
            if (part) {
                a.chunk = (int)(((nel + S - 1) / S + 255) / 256 * 256);
                const int Sy = (int)((nel + a.chunk - 1) / a.chunk);
                k_bn_stat_part<<<dim3(M, Sy), 256, 0, stream>>>(y, part, B, M, T, a.chunk);
                a.stat_part = part;
                a.nblk = Sy;
                k_bn_act_fwd<<<dim3(M, Sy), 256, 0, stream>>>(a);
                return t2v_check_launch();
            }
            a.chunk = 0;
        }
    }
    k_bn_act_fwd<<<M, 256, 0, stream>>>(a);
    return t2v_check_launch();
}

extern "C" int t2v_bn_act_bwd(const float* y, const float* dout, const float* mean, const float* rstd,
                              const float* gamma, const float* beta, float* dy, float* dgamma, float* dbeta,
                              float* dconv_bias, int B, int M, int T, int act, float p_drop, uint64_t seed,
                              uint32_t rng_stream, uint32_t rng_t, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!y || !dout || !mean || !rstd || !gamma || !beta || !dy || !dgamma || !dbeta) return T2V_ERR_ARG;
    BnBwdArgs a;
    a.y = y; a.dout = dout; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta; a.dy = dy;
    a.dgamma = dgamma; a.dbeta = dbeta; a.dconv_bias = dconv_bias; a.B = B; a.M = M; a.T = T; a.act = act; a.p_drop = p_drop;
    a.seed = seed; a.rng_stream = rng_stream; a.rng_t = rng_t; a.step = t2v_step_for(stream);
    a.eval_mode = 0;
    bn_bwd_launch(a, stream);
    return t2v_check_launch();
}

// eval-mode BatchNorm (model.eval(): running statistics): y = gamma (x - running_mean) rstd + beta with CONSTANT statistics,
// so dx = gamma rstd dz without the batch-mean terms and the convolution bias in front of it does get a gradient.
// mean = running_mean, rstd = 1 / sqrt(running_var + eps).
extern "C" int t2v_bn_act_bwd_eval(const float* y, const float* dout, const float* mean, const float* rstd,
                                   const float* gamma, const float* beta, float* dy, float* dgamma, float* dbeta,
                                   float* dconv_bias, int B, int M, int T, int act, float p_drop, uint64_t seed,
                                   uint32_t rng_stream, uint32_t rng_t, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!y || !dout || !mean || !rstd || !gamma || !beta || !dy || !dgamma || !dbeta) return T2V_ERR_ARG;
    BnBwdArgs a;
    a.y = y; a.dout = dout; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta; a.dy = dy;
    a.dgamma = dgamma; a.dbeta = dbeta; a.dconv_bias = dconv_bias; a.B = B; a.M = M; a.T = T; a.act = act; a.p_drop = p_drop;
    a.seed = seed; a.rng_stream = rng_stream; a.rng_t = rng_t; a.step = t2v_step_for(stream);
    a.eval_mode = 1;
    bn_bwd_launch(a, stream);
    return t2v_check_launch();
}
