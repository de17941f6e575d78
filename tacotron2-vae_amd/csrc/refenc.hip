// Reference-encoder building blocks (reference modules.py:34-85, CoordConv.py:37-74,142-161):
//   k_conv2d_s2_{fwd,dx,dw} : Conv2d 3x3, stride 2, pad 1 (direct form — the six layers total 0.28 GMAC),
//                             with the CoordConv coordinate channels generated on the fly for layer 1
//   k_gru_{fwd,bwd}         : nn.GRU(256 -> 256) recurrence over the <= 16 frames that survive 6 stride-2
//                             convs (input projections are a time-batched GEMM outside)
//   k_loss                  : Tacotron2Loss_VAE forward + gradient in one pass (loss_function.py:27-44)
// BatchNorm2d + ReLU reuse the per-channel kernels of bn_act.hip on the (B, C, H*W) view.
#include "t2v_common.h"
#include "t2v_kernels.h"

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

// one workgroup per (co, c): dw[co][c][kh][kw] = sum_{b,ho,wo} dy[b][co][ho][wo] * in[b][c][2ho-1+kh][2wo-1+kw]
__global__ __launch_bounds__(256) void k_conv2d_s2_dw(Conv2dArgs a) {
    __shared__ float red[4][9];
    const int Cin = a.Cx + (a.coord ? 3 : 0);
    const int co = blockIdx.x / Cin, c = blockIdx.x % Cin;
    const int tid = threadIdx.x;
    float acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.f;
    const int per = a.Ho * a.Wo;
    for (int i = tid; i < a.B * per; i += 256) {
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
    if (tid < 9) a.y[((size_t)co * Cin + c) * 9 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
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

extern "C" int t2v_conv2d_s2_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, int B, int Cx,
                                 int H, int W, int Cout, int coord, void* stream_) {
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
    k_conv2d_s2_dw<<<Cout * (Cx + (coord ? 3 : 0)), 256, 0, stream>>>(a);
    return t2v_check_launch();
}

// ------------------------------------------------------------------------------------------------ GRU
// nn.GRU semantics: r,z,n gates; n = tanh(gi_n + r * (W_hn h + b_hn)); h' = (1-z) n + z h.
// gi (B,T,768) = x·W_ih^T + b_ih is computed outside.  One workgroup of 768 threads: thread g owns gate row
// g of W_hh (256 floats streamed from L2 each step); h lives in LDS.  T <= 16 here (T_out / 64 frames).
struct GruArgs {
    const float* gi;      // (B,T,768)
    const float* whh;     // (768,256)
    const float* bhh;     // (768)
    float* hs;            // (B,T+1,256) hidden states, hs[:,0] = 0 written here
    float* gsave;         // (B,T,4,256): r, z, n, (W_hn h + b_hn) for the backward; or NULL
    const float* dh_last; // bwd: (B,256) gradient of the last hidden state
    float* dgi;           // bwd: (B,T,768) grad wrt gi
    float* dgh;           // bwd: (B,T,768) grad wrt (W_hh h + b_hh) rows
    int B, T;
};

__global__ __launch_bounds__(768) void k_gru_fwd(GruArgs a) {
    __shared__ float h[16][256];
    __shared__ float gh[16][768];
    const int g = threadIdx.x;
    for (int i = g; i < 16 * 256; i += 768) (&h[0][0])[i] = 0.f;
    for (int i = g; i < a.B * 256; i += 768) a.hs[((size_t)(i >> 8) * (a.T + 1)) * 256 + (i & 255)] = 0.f;
    __syncthreads();
    const float4* wrow = (const float4*)(a.whh + (size_t)g * 256);
    const float bias = a.bhh[g];
    for (int t = 0; t < a.T; ++t) {
        float acc[16];
#pragma unroll
        for (int b = 0; b < 16; ++b) acc[b] = bias;
#pragma unroll 8
        for (int k4 = 0; k4 < 64; ++k4) {
            const float4 w = wrow[k4];
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                if (b < a.B) {
                    const float4 hv = *(const float4*)&h[b][4 * k4];
                    acc[b] = fmaf(w.x, hv.x, acc[b]); acc[b] = fmaf(w.y, hv.y, acc[b]);
                    acc[b] = fmaf(w.z, hv.z, acc[b]); acc[b] = fmaf(w.w, hv.w, acc[b]);
                }
            }
        }
#pragma unroll
        for (int b = 0; b < 16; ++b) if (b < a.B) gh[b][g] = acc[b];
        __syncthreads();
        for (int i = g; i < a.B * 256; i += 768) {
            const int b = i >> 8, u = i & 255;
            const float* gi = a.gi + ((size_t)b * a.T + t) * 768;
            const float r = sigmoidf_(gi[u] + gh[b][u]);
            const float z = sigmoidf_(gi[256 + u] + gh[b][256 + u]);
            const float hn = gh[b][512 + u];
            const float n = tanhf_(gi[512 + u] + r * hn);
            const float hnew = (1.f - z) * n + z * h[b][u];
            if (a.gsave) {
                float* s = a.gsave + (((size_t)b * a.T + t) * 4) * 256 + u;
                s[0] = r; s[256] = z; s[512] = n; s[768] = hn;
            }
            a.hs[((size_t)b * (a.T + 1) + t + 1) * 256 + u] = hnew;
            h[b][u] = hnew;       // each (b,u) is touched by exactly one thread here
        }
        __syncthreads();
    }
}

// BPTT: thread u (256 threads... launched with 768) -> first the gate gradients of step t for (b,u),
// then dh_prev[k] = sum_g W_hh[g][k] * dgh[g]: thread k<256 owns column k (coalesced row reads).
__global__ __launch_bounds__(768) void k_gru_bwd(GruArgs a) {
    __shared__ float dh[16][256];
    __shared__ float dg[16][768];
    __shared__ float part[3][16][256];
    const int tid = threadIdx.x;
    for (int i = tid; i < a.B * 256; i += 768) dh[i >> 8][i & 255] = a.dh_last[i];
    __syncthreads();
    for (int t = a.T - 1; t >= 0; --t) {
        for (int i = tid; i < a.B * 256; i += 768) {
            const int b = i >> 8, u = i & 255;
            const float* s = a.gsave + (((size_t)b * a.T + t) * 4) * 256 + u;
            const float r = s[0], z = s[256], n = s[512], hn = s[768];
            const float hprev = a.hs[((size_t)b * (a.T + 1) + t) * 256 + u];
            const float d = dh[b][u];
            const float dn = d * (1.f - z) * (1.f - n * n);
            const float dz = d * (hprev - n) * z * (1.f - z);
            const float dr = dn * hn * r * (1.f - r);
            float* gi = a.dgi + ((size_t)b * a.T + t) * 768 + u;
            gi[0] = dr; gi[256] = dz; gi[512] = dn;
            float* gh = a.dgh + ((size_t)b * a.T + t) * 768 + u;
            gh[0] = dr; gh[256] = dz; gh[512] = dn * r;
            dg[b][u] = dr; dg[b][256 + u] = dz; dg[b][512 + u] = dn * r;
            dh[b][u] = d * z;                       // direct path h' = ... + z h
        }
        __syncthreads();
        {   // dh_prev[k] += sum_g W_hh[g][k] dg[g]: thread = (k = tid&255, third = tid>>8) sums 256 gate rows
            const int k = tid & 255, third = tid >> 8;
            float acc[16];
#pragma unroll
            for (int b = 0; b < 16; ++b) acc[b] = 0.f;
            const float* wp = a.whh + (size_t)(256 * third) * 256 + k;
#pragma unroll 16
            for (int g = 0; g < 256; ++g) {
                const float w = wp[(size_t)g * 256];
#pragma unroll
                for (int b = 0; b < 16; ++b) if (b < a.B) acc[b] = fmaf(w, dg[b][256 * third + g], acc[b]);
            }
#pragma unroll
            for (int b = 0; b < 16; ++b) if (b < a.B) part[third][b][k] = acc[b];
        }
        __syncthreads();
        for (int i = tid; i < a.B * 256; i += 768) {
            const int b = i >> 8, k = i & 255;
            dh[b][k] += (part[0][b][k] + part[1][b][k]) + part[2][b][k];
        }
        __syncthreads();
    }
}

extern "C" int t2v_gru_fwd(const float* gi, const float* whh, const float* bhh, float* hs, float* gsave, int B, int T,
                           void* stream_) {
    if (!gi || !whh || !bhh || !hs || B < 1 || B > 16 || T < 1) return T2V_ERR_ARG;
    GruArgs a;
    a.gi = gi; a.whh = whh; a.bhh = bhh; a.hs = hs; a.gsave = gsave; a.dh_last = nullptr; a.dgi = nullptr; a.dgh = nullptr;
    a.B = B; a.T = T;
    k_gru_fwd<<<1, 768, 0, (hipStream_t)stream_>>>(a);
    return t2v_check_launch();
}

extern "C" int t2v_gru_bwd(const float* whh, const float* hs, const float* gsave, const float* dh_last, float* dgi,
                           float* dgh, int B, int T, void* stream_) {
    if (!whh || !hs || !gsave || !dh_last || !dgi || !dgh || B < 1 || B > 16 || T < 1) return T2V_ERR_ARG;
    GruArgs a;
    a.gi = nullptr; a.whh = whh; a.bhh = nullptr; a.hs = (float*)hs; a.gsave = (float*)gsave; a.dh_last = dh_last;
    a.dgi = dgi; a.dgh = dgh; a.B = B; a.T = T;
    k_gru_bwd<<<1, 768, 0, (hipStream_t)stream_>>>(a);
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
    unsigned* ticket;
};

__global__ __launch_bounds__(256) void k_loss(LossArgs a) {
    __shared__ float scr[4];
    __shared__ int last;
    const int tid = threadIdx.x;
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
        a.dmu[i] = a.klw * m;
        a.dlogvar[i] = a.klw * -0.5f * (1.f - e);
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
        a.out[0] = recon + a.klw * (float)k;
        a.out[3] = a.klw;
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
    a.n_mel = n_mel; a.n_gate = n_gate; a.n_lat = n_lat; a.nblk = 64; a.klw = kl_weight; a.ticket = ticket;
    k_loss<<<64, 256, 0, (hipStream_t)stream_>>>(a);
    return t2v_check_launch();
}
