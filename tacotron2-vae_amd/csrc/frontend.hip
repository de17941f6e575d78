// STFT -> mel front end on device (reference layers.py:75-92 = STFT.transform stft.py:77-105 +
// mel_basis matmul + dynamic_range_compression audio_processing.py:77-83), for the default
// geometry n_fft = win = 1024, hop = 256, 80 mels.
//
// One wavefront per frame: reflect-padded, Hann-windowed 1024 real samples are packed as 512
// complex points, transformed by a Stockham radix-8 FFT (3 passes, 64 lanes x 8 points, LDS
// exchange inside the wave), untangled to the 513-bin real spectrum, |.|, sparse triangular mel
// filters (taps in LDS in lane order; CSR rows for any other filterbank), log(clamp(.,1e-5)).  A workgroup of
// 4 waves produces 16 consecutive frames so mel rows leave as 64-byte runs.  1 KiB of samples in, 320 B out per
// frame — and bound by its instruction count, not by HBM (round 6: VALU-issue counters in profiles/r06_pmc_frontend.txt;
// interior frames skip the 64-bit reflect arithmetic and load two samples at once: 257 -> 145 us for 51 328 frames).
#include "t2v_common.h"
#include "t2v_kernels.h"

#define FE_NFFT 1024
#define FE_HOP 256
#define FE_NMEL 80
#define FE_FRAMES_PER_WG 16
#define FE_WAVES 4
// mel stage (round 6): the filter taps of a lane's bins sit in LDS in lane order, [tap][lane], shared by the four waves — bins 0..63
// one per lane (<= FE_W1 taps), bins 64..79 on four lanes each (<= 4 FE_W2Q taps, quad sum by DPP).  The CSR loop it replaces walked
// <= 37 taps per bin with a global weight load in front of every FMA and 16 of 64 lanes active for the widest filters: a third of
// the kernel.  (The same taps in REGISTERS: 187 instead of 113 VGPRs = two waves per SIMD instead of four — faster for one batch
// of six utterances, slower on a full chip.)
#define FE_W1 20
#define FE_W2Q 10

struct c32 { float x, y; };
__device__ __forceinline__ c32 cmul(c32 a, c32 b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ c32 cadd(c32 a, c32 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ c32 csub(c32 a, c32 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ c32 mul_mi(c32 a) { return {a.y, -a.x}; }   // a * (-i)

// in-place 8-point DFT (forward, e^{-2 pi i/8}), outputs in natural order
__device__ __forceinline__ void dft8(c32* v) {
    const float h = 0.70710678118654752440f;
    c32 a0 = cadd(v[0], v[4]), a1 = csub(v[0], v[4]);
    c32 a2 = cadd(v[2], v[6]), a3 = mul_mi(csub(v[2], v[6]));
    c32 a4 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]);
    c32 a6 = cadd(v[3], v[7]), a7 = mul_mi(csub(v[3], v[7]));
    c32 b0 = cadd(a0, a2), b2 = csub(a0, a2), b1 = cadd(a1, a3), b3 = csub(a1, a3);
    c32 b4 = cadd(a4, a6), b6 = mul_mi(csub(a4, a6)), b5 = cadd(a5, a7), b7 = csub(a5, a7);
    // twiddles W8^1 = h(1 - i), W8^3 = -h(1 + i)
    c32 t5 = {h * (b5.x + b5.y), h * (b5.y - b5.x)};
    c32 t7 = {h * (-b7.x + b7.y), h * (-b7.y - b7.x)};
    v[0] = cadd(b0, b4); v[4] = csub(b0, b4);
    v[2] = cadd(b2, b6); v[6] = csub(b2, b6);
    v[1] = cadd(b1, t5); v[5] = csub(b1, t5);
    v[3] = cadd(b3, t7); v[7] = csub(b3, t7);
}

struct FrontendArgs {
    const float* wav_f32;        // (B, n_stride) or NULL
    const int16_t* wav_i16;      // (B, n_stride) or NULL
    const int64_t* n_samples;    // (B)
    int n_stride;
    float scale;                 // applied to the samples (1/32768 for int16 PCM)
    const float* window;         // (1024) periodic Hann
    const c32* tw512;            // (512)  exp(-2 pi i k/512)
    const c32* tw1024;           // (513)  exp(-2 pi i k/1024)
    const int* mel_start;        // (80)
    const int* mel_len;          // (80)
    const float* mel_w;          // (80, maxw)
    int maxw;
    float* mel_out;              // (B, 80, t_stride)
    int t_stride;
};

#ifndef FE_WAVES_PER_EU
#define FE_WAVES_PER_EU 3      // three waves per SIMD: 164 VGPRs, no spills (four: 128 VGPRs and 76 bytes of scratch per lane)
#endif
__global__ __launch_bounds__(256, FE_WAVES_PER_EU) void k_mel_frontend(FrontendArgs a) {
    __shared__ c32 zbuf[FE_WAVES][512];
    __shared__ float mag[FE_WAVES][516];
    __shared__ float tile[FE_NMEL][FE_FRAMES_PER_WG + 1];
    __shared__ float wl[(FE_W1 + FE_W2Q) * 64];
    const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t n = a.n_samples[b];
    const int T = (int)(n / FE_HOP) + 1;
    const int t0 = blockIdx.x * FE_FRAMES_PER_WG;
    c32* z = zbuf[wave];
    float* mg = mag[wave];
    // ---- mel filter taps of this lane (zero beyond a filter's length)
    const int ln1_ = a.mel_len[lane], ln2_ = a.mel_len[64 + (lane >> 2)];
    const bool mel_regs = __all(ln1_ <= FE_W1 && ln2_ <= 4 * FE_W2Q);      // (any other filterbank: the CSR loop below)
    const int st1 = a.mel_start[lane], st2 = a.mel_start[64 + (lane >> 2)];
    if (mel_regs) {
        for (int e = threadIdx.x; e < (FE_W1 + FE_W2Q) * 64; e += 256) {
            const int i = e >> 6, l = e & 63;
            float w;
            if (i < FE_W1) w = i < a.mel_len[l] ? a.mel_w[(size_t)l * a.maxw + i] : 0.f;
            else {
                const int m2 = 64 + (l >> 2), tap = (l & 3) + 4 * (i - FE_W1);
                w = tap < a.mel_len[m2] ? a.mel_w[(size_t)m2 * a.maxw + tap] : 0.f;
            }
            wl[e] = w;
        }
        __syncthreads();
    }

    for (int fi = wave; fi < FE_FRAMES_PER_WG; fi += FE_WAVES) {
        const int t = t0 + fi;
        if (t >= T) {            // beyond this utterance: the collate pad value is 0.0 (data_utils.py:126)
            for (int m = lane; m < FE_NMEL; m += 64) tile[m][fi] = 0.f;
            continue;
        }
        // ---- load + reflect pad (F.pad 'reflect' excludes the edge sample) + window, pack z = x[2n] + i x[2n+1]
        c32 v[8];
        const int64_t base = (int64_t)t * FE_HOP - FE_NFFT / 2;
        const size_t off0 = (size_t)b * a.n_stride + (size_t)(base > 0 ? base : 0);
        if (base >= 0 && base + FE_NFFT <= n && (off0 & 1) == 0) {
            // interior frame (all but the first two and the last two of an utterance): no reflection, the two samples of a packed
            // point come as ONE load (round 6: the general path below spends ~15 instructions per sample on 64-bit reflect arithmetic)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int nn = lane + 64 * k;
                const float2 w2 = ((const float2*)a.window)[nn];
                float s0, s1;
                if (a.wav_i16) {
                    const uint32_t pr = ((const uint32_t*)(a.wav_i16 + off0))[nn];
                    s0 = (float)(int16_t)(pr & 0xffffu);
                    s1 = (float)(int16_t)(pr >> 16);
                } else {
                    const float2 x2 = ((const float2*)(a.wav_f32 + off0))[nn];
                    s0 = x2.x; s1 = x2.y;
                }
                v[k] = {s0 * a.scale * w2.x, s1 * a.scale * w2.y};
            }
        } else
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int nn = lane + 64 * k;
            float s[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                int64_t p = base + 2 * nn + c;
                if (p < 0) p = -p;
                if (p >= n) p = 2 * (n - 1) - p;
                p = p < 0 ? 0 : p;      // n <= 512 is rejected by the host wrapper; never read out of bounds anyway
                const size_t off = (size_t)b * a.n_stride + (size_t)p;
                const float x = a.wav_i16 ? (float)a.wav_i16[off] : a.wav_f32[off];
                s[c] = x * a.scale * a.window[2 * nn + c];
            }
            v[k] = {s[0], s[1]};
        }
        // ---- 512-point complex FFT, Stockham radix-8: pass Ns = 1, 8, 64 (thread j = lane)
        dft8(v);
#pragma unroll
        for (int r = 0; r < 8; ++r) z[lane * 8 + r] = v[r];            // j0 = j*8, stride Ns = 1
        // pass 2: Ns = 8
        {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = z[lane + 64 * r];
            const int k = lane & 7;
#pragma unroll
            for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], a.tw512[8 * k * r]);     // W_64^{k r}
            dft8(v);
            const int j0 = (lane >> 3) * 64 + k;
#pragma unroll
            for (int r = 0; r < 8; ++r) z[j0 + 8 * r] = v[r];
        }
        // pass 3: Ns = 64
        {
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = z[lane + 64 * r];
#pragma unroll
            for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], a.tw512[lane * r]);      // W_512^{k r}
            dft8(v);
#pragma unroll
            for (int r = 0; r < 8; ++r) z[lane + 64 * r] = v[r];
        }
        // ---- untangle to the real spectrum X[k], k = 0..512, magnitude
        // (eight bins per lane in two unrolled groups of four: the LDS reads and twiddle loads of a group are in flight together —
        // all eight at once cost 28 more registers and a wave per SIMD; bin 512 by lane 0)
        {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                c32 zk[4], zc[4], tw[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = lane + 64 * (4 * h + r);
                    zk[r] = z[k];
                    zc[r] = z[(512 - k) & 511];
                    tw[r] = a.tw1024[k];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const c32 e = {0.5f * (zk[r].x + zc[r].x), 0.5f * (zk[r].y - zc[r].y)};      // (Z[k] + conj Z[N-k]) / 2
                    const c32 o = {0.5f * (zk[r].x - zc[r].x), 0.5f * (zk[r].y + zc[r].y)};      // (Z[k] - conj Z[N-k]) / 2
                    const c32 wo = cmul(tw[r], o);
                    const c32 X = cadd(e, mul_mi(wo));                                           // E - i W^k O
                    mg[lane + 64 * (4 * h + r)] = sqrtf(X.x * X.x + X.y * X.y);
                }
            }
            if (lane == 0) {
                const c32 z0 = z[0];
                const c32 e = {z0.x, 0.f}, o = {0.f, z0.y};
                const c32 wo = cmul(a.tw1024[512], o);
                const c32 X = cadd(e, mul_mi(wo));
                mg[512] = sqrtf(X.x * X.x + X.y * X.y);
            }
        }
        // ---- sparse mel filterbank + log compression
        if (mel_regs) {
            float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
            for (int i = 0; i < FE_W1; i += 2) {          // (taps past the filter: weight zero, index clamped to a written bin)
                acc0 = fmaf(wl[i * 64 + lane], mg[min(st1 + i, 512)], acc0);
                acc1 = fmaf(wl[(i + 1) * 64 + lane], mg[min(st1 + i + 1, 512)], acc1);
            }
            tile[lane][fi] = logf(fmaxf(acc0 + acc1, 1e-5f));
            float b0 = 0.f, b1 = 0.f;
            const int q = lane & 3;
#pragma unroll
            for (int i = 0; i < FE_W2Q; i += 2) {
                b0 = fmaf(wl[(FE_W1 + i) * 64 + lane], mg[min(st2 + q + 4 * i, 512)], b0);
                b1 = fmaf(wl[(FE_W1 + i + 1) * 64 + lane], mg[min(st2 + q + 4 * i + 4, 512)], b1);
            }
            float bs = b0 + b1;
            bs = T2V_DPP_ADD(bs, 0xB1);                   // quad_perm [1,0,3,2]
            bs = T2V_DPP_ADD(bs, 0x4E);                   // quad_perm [2,3,0,1]
            if (q == 0) tile[64 + (lane >> 2)][fi] = logf(fmaxf(bs, 1e-5f));
        } else
        for (int m = lane; m < FE_NMEL; m += 64) {
            const int st = a.mel_start[m], ln = a.mel_len[m];
            const float* w = a.mel_w + (size_t)m * a.maxw;
            float acc = 0.f;
            for (int i = 0; i < ln; ++i) acc = fmaf(w[i], mg[st + i], acc);
            tile[m][fi] = logf(fmaxf(acc, 1e-5f));
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < FE_NMEL * FE_FRAMES_PER_WG; i += 256) {
        const int m = i / FE_FRAMES_PER_WG, fi = i % FE_FRAMES_PER_WG;
        const int t = t0 + fi;
        if (t < a.t_stride) a.mel_out[((size_t)b * FE_NMEL + m) * a.t_stride + t] = tile[m][fi];
    }
}

extern "C" int t2v_mel_frontend(const float* wav_f32, const int16_t* wav_i16, const int64_t* n_samples, int B,
                                int n_stride, float scale, int n_fft, int hop, int n_mel, const float* window,
                                const float* tw512, const float* tw1024, const int32_t* mel_start,
                                const int32_t* mel_len, const float* mel_w, int maxw, float* mel_out,
                                int t_stride, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_fft != FE_NFFT || hop != FE_HOP || n_mel != FE_NMEL) return T2V_ERR_DIMS;
    if ((!wav_f32 && !wav_i16) || !n_samples || !window || !tw512 || !tw1024 || !mel_start || !mel_len || !mel_w ||
        !mel_out || B < 1 || t_stride < 1)
        return T2V_ERR_ARG;
    FrontendArgs a;
    a.wav_f32 = wav_f32; a.wav_i16 = wav_i16; a.n_samples = n_samples; a.n_stride = n_stride; a.scale = scale;
    a.window = window; a.tw512 = (const c32*)tw512; a.tw1024 = (const c32*)tw1024;
    a.mel_start = mel_start; a.mel_len = mel_len; a.mel_w = mel_w; a.maxw = maxw;
    a.mel_out = mel_out; a.t_stride = t_stride;
    dim3 grid((t_stride + FE_FRAMES_PER_WG - 1) / FE_FRAMES_PER_WG, B);
    k_mel_frontend<<<grid, 256, 0, stream>>>(a);
    return t2v_check_launch();
}
