// Free-running decode as ONE persistent launch (reference Decoder.inference model.py:428-464 / the loop at
// synthesizer.py:139-154): 256 workgroups x 512 threads stay resident for the whole utterance, every weight of the
// loop lives in registers / LDS, and the five dependent stages of a frame
//     attention_rnn -> attention -> decoder_rnn -> projection (+ folded Prenet layer 0) -> Prenet layer 1
// hand their small result vectors (1024 / 512 / 256 values per item) from CU to CU as plain 4-byte values in a FRESH,
// sentinel-filled exchange row per frame (write-through sc1 stores, sc1 loads; a word that is no longer 0xFFFFFFFF IS the
// data — no tags, no counters, no fences, no grid barrier).  Round 3: half the bytes of the former 8-byte {value, tag}
// granules — what a gather costs is its bytes (a CU pulls ~11 B/cycle from beyond its L2) — and an adaptive nap
// in front of every poll, so that the first round of a gather is not the one that is bound to fail.  Nothing is streamed from HBM per frame: the 71 MB of LSTM weights that the launch-per-stage
// loop (decoder_infer.hip) re-reads every frame are read once.
//
// Roles (every workgroup runs the frame loop, phases in the same order, so the waits cannot form a cycle):
//   all 256 workgroups : 4 hidden units of both LSTM cells (16 gate rows each): rows x K/8 per wave in VGPRs
//                        (136 per lane), VALU dot products against the LDS copy of the state vectors
//   wg 0 .. 8B-1       : attention slice (item b = wg/8, 16 attention dims + 64 context columns): W_q slice, memory /
//                        processed-memory slices and the alignment window stay in LDS for all frames
//   wg 64 .. 106       : 8 of the 337 projection rows each (80 mel + gate + 256 folded Prenet-0 rows), rows in LDS
//   wg 128 .. 159      : 8 Prenet layer-1 rows each
// The gate row decides the stop (model.py:453) and publishes it as a granule every workgroup reads at the top of the
// next frame: nothing is computed after the stop frame.
// Limits of this path: B <= 4, T_in <= 224 (LDS residency of the attention operands); the launch-per-stage loop serves
// everything else (and the stepwise decode() API).
#include "t2v_common.h"
#include "t2v_kernels.h"
#include <limits.h>

#define PD_THREADS 512
#define PD_MAXB 4
#define PD_MAXT 224
#ifndef PD_CTXW
#define PD_CTXW 4          // positions per round of the context sum (t2v_ctx_partial); the attention workgroups have few registers to spare
#endif
#define PD_SPIN 3000000u
#define PD_XW 2816                 // LDS state row: [h_att 1024 | ctx 512 | pre1 256 | h_dec 1024]
#define PD_X_HA 0
#define PD_X_CX 1024
#define PD_X_P1 1536
#define PD_X_HD 1792
#define PD_KATT 1792               // attention_rnn K: [h_att | ctx | pre1]   (contiguous in the LDS row)
#define PD_KDEC 2560               // decoder_rnn   K: [h_att | ctx | h_dec]  (h_dec sits 256 further)
#define PD_NROW 337
#define PD_WG_PROJ 64
#define PD_WG_PRE1 128

struct PersistArgs {
    const float* w_ih_att; const float* w_hh_att; const float* w_ih_dec; const float* w_hh_dec;   // nn.LSTMCell tensors
    const float* bias_att; const float* bias_dec;     // (4096) b_ih + b_hh
    const float* wq;          // (128,1024) query_layer weight
    const float* wcomb;       // fused location filter, forward copy F[d][g][st]
    const float* v;           // (128)
    const float* proj_w;      // (337,1536) [linear_projection; gate_layer; W0·linear_projection]
    const float* proj_b;      // (337)
    const float* w1;          // (256,256) Prenet layer 1
    const float* memory;      // (B,T_in,512)
    const float* pm;          // (B,T_in,128)
    const int32_t* lengths;   // (B) or NULL
    const float* pre_first;   // (B,256) Prenet(go frame)
    float* MEL; float* GATE; float* AL;      // (Tmax,B,80) (Tmax,B) (Tmax+1,B,T_in): AL[t+1] = weights of frame t
    int* stop_flag;
    float* xg;                // exchange rows, one per frame: t_end x pd_row(B) floats, sentinel-filled by the launcher
    unsigned* err;
    int B, T_in, t_end;
    float gate_logit_thr, p_prenet;
    uint64_t seed;
    unsigned long long* prof;   // optional: stamps of frame 100 (workgroup 0 slots 0..9, workgroup 64 slots 10..13, workgroup 128 slots 14..16)
};
// per-workgroup time line of frame 100 on the chip-wide 100 MHz counter: prof[64 + workgroup * 16 + slot] (tools/dbg_persist.py)
#define PD_RT(SLOT) do { if (a.prof && t == 100 && tid == 0) a.prof[64 + wg * 16 + (SLOT)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define PD_STAMP(WG, I) do { if (a.prof && t == 100 && wg == (WG) && tid == 0) a.prof[(I)] = __builtin_readcyclecounter(); } while (0)

// exchange row of one frame (floats)
__host__ __device__ static inline unsigned pd_hatt(int B) { (void)B; return 0u; }
__host__ __device__ static inline unsigned pd_hdec(int B) { return (unsigned)B * 1024u; }
__host__ __device__ static inline unsigned pd_ctx(int B) { return (unsigned)B * 2048u; }
__host__ __device__ static inline unsigned pd_pre0(int B) { return (unsigned)B * 2560u; }
__host__ __device__ static inline unsigned pd_pre1(int B) { return (unsigned)B * 2816u; }
__host__ __device__ static inline unsigned pd_ex(int B) { return (unsigned)B * 3072u; }          // [b][8][256]
__host__ __device__ static inline unsigned pd_stop(int B) { return (unsigned)B * 5120u; }
__host__ __device__ static inline unsigned pd_row(int B) { return (unsigned)B * 5120u + 32u; }

#define PD_SENT 0xFFFFFFFFu
#ifndef PD_POLL2
#define PD_POLL2 0
#endif
#define PD_SC1 16
typedef unsigned pd_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned pd_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pd_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
// offsets in floats
__device__ __forceinline__ void pd_put(__amdgpu_buffer_rsrc_t r, unsigned off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)(off * 4u), 0, PD_SC1);
}
__device__ __forceinline__ void pd_put4(__amdgpu_buffer_rsrc_t r, unsigned off, float v0, float v1, float v2, float v3) {
    const pd_u32x4 x = {__float_as_uint(v0), __float_as_uint(v1), __float_as_uint(v2), __float_as_uint(v3)};
    __builtin_amdgcn_raw_buffer_store_b128(x, r, (int)(off * 4u), 0, PD_SC1);
}
__device__ __forceinline__ unsigned pd_get(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, (int)(off * 4u), 0, PD_SC1);
}
__device__ __forceinline__ pd_u32x2 pd_get2(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b64(r, (int)(off * 4u), 0, PD_SC1);
}
__device__ __forceinline__ pd_u32x4 pd_get4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)(off * 4u), 0, PD_SC1);
}
__device__ __forceinline__ bool pd_give_up(unsigned& spins, unsigned* err, int* flag) {
    if (++spins > PD_SPIN || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = 0;
        return true;
    }
    return false;
}

// Poll the n (<= 2 * 512, even) values at `off` until none is the sentinel, then copy them to LDS: thread = two adjacent
// values (one 8-byte load).  Naps `nap` first (adapted by the caller from the returned number of failed rounds).  On a
// timeout the error word is set and *flag cleared (an LDS int, 1 on entry) — the caller syncs before reading dst.
__device__ __forceinline__ int pd_gather(float* dst, __amdgpu_buffer_rsrc_t r, unsigned off, int n, int nap, unsigned* err, int* flag) {
    const int tid = threadIdx.x;
    for (int i = 0; i < nap; i += 4) __builtin_amdgcn_s_sleep(4);
    const bool on = 2 * tid < n;
    pd_u32x2 x = {0u, 0u};
    unsigned spins = 0;
#if PD_POLL2
    // (measurement variant, -DPD_POLL2=1: two polls in flight, half a round trip apart.  Same-box A/B: 14.68 vs 14.39 us per
    // frame — twice the requests on lines that 256 workgroups already poll cost more than the quarter round trip they save)
    pd_u32x2 x0 = {PD_SENT, PD_SENT}, x1 = {PD_SENT, PD_SENT};
    if (on) x0 = pd_get2(r, off + 2u * (unsigned)tid);
    __builtin_amdgcn_s_sleep(4);
    if (on) x1 = pd_get2(r, off + 2u * (unsigned)tid);
    for (;;) {
        if (__all(!on || (x0[0] != PD_SENT && x0[1] != PD_SENT))) { x = x0; break; }
        if (on) x0 = pd_get2(r, off + 2u * (unsigned)tid);
        if (__all(!on || (x1[0] != PD_SENT && x1[1] != PD_SENT))) { x = x1; break; }
        if (on) x1 = pd_get2(r, off + 2u * (unsigned)tid);
        if (pd_give_up(spins, err, flag)) break;
    }
#else
    for (;;) {
        if (on) x = pd_get2(r, off + 2u * (unsigned)tid);
        if (__all(x[0] != PD_SENT && x[1] != PD_SENT)) break;
        __builtin_amdgcn_s_sleep(1);
        if (pd_give_up(spins, err, flag)) break;
    }
#endif
    if (on) *(float2*)(dst + 2 * tid) = make_float2(__uint_as_float(x[0]), __uint_as_float(x[1]));
    return (int)spins;
}
// The same row of ALL items of the batch (item b at dst + b * PD_XW, row offset off + b * stride) in ONE polling pass: the loads of
// the (up to four) items are in flight together.  Item by item — pd_gather in a loop — every item pays its own memory round
// trip on the frame's chain: 15.5 us per frame at B = 2 against 12.6 at B = 1 (round 6).  Items past B re-read item B - 1.
__device__ __forceinline__ int pd_gather_items(float* dst, __amdgpu_buffer_rsrc_t r, unsigned off, unsigned stride, int n, int B, int nap,
                                               unsigned* err, int* flag) {
    if (B == 1) return pd_gather(dst, r, off, n, nap, err, flag);
    const int tid = threadIdx.x;
    for (int i = 0; i < nap; i += 4) __builtin_amdgcn_s_sleep(4);
    const bool on = 2 * tid < n;
    pd_u32x2 x[PD_MAXB];
#pragma unroll
    for (int b = 0; b < PD_MAXB; ++b) x[b] = pd_u32x2{0u, 0u};
    unsigned spins = 0;
    for (;;) {
        bool ok = true;
        if (on) {
#pragma unroll
            for (int b = 0; b < PD_MAXB; ++b) x[b] = pd_get2(r, off + (unsigned)min(b, B - 1) * stride + 2u * (unsigned)tid);
#pragma unroll
            for (int b = 0; b < PD_MAXB; ++b) ok = ok && x[b][0] != PD_SENT && x[b][1] != PD_SENT;
        }
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(1);
        if (pd_give_up(spins, err, flag)) break;
    }
    if (on) {
#pragma unroll
        for (int b = 0; b < PD_MAXB; ++b)
            if (b < B) *(float2*)(dst + (size_t)b * PD_XW + 2 * tid) = make_float2(__uint_as_float(x[b][0]), __uint_as_float(x[b][1]));
    }
    return (int)spins;
}
// a copy of v the compiler cannot see through: what is derived from it is recomputed where it is used (two or three VALU
// instructions) instead of being hoisted out of the frame loop and kept — spilled — across it
__device__ __forceinline__ int pd_opaque(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ int pd_adapt(int nap, int rounds) {       // units of 64 cycles; first round should just succeed
    if (rounds > 1) return min(48, nap + 4 * min(rounds - 1, 3));
    if (rounds == 0) return (3 * nap) >> 2;
    return nap;
}

// LSTM gate rows of this workgroup for one cell: lane = (row r = lane>>2 (unit r>>2, gate r&3), kq = lane&3); weights in
// registers, VALU dot products against the LDS copy of the state vectors, partial dot products -> red[wave][r][b].
// Round 4: the two LSTM GEMVs of a frame are cut by WHEN their inputs exist, so that only the columns of the value that has
// just arrived are left on the frame's dependency chain:
//   attention_rnn(t+1) = [h_att(t) | ctx(t)] (1536 columns, known when ctx(t) has been gathered: evaluated in the shadow of
//                        the projection / Prenet stages of frame t)  +  Prenet output (256 columns: the chain)
//   decoder_rnn(t)     = h_dec(t-1) (1024 columns, known since the previous frame)  +  h_att(t) (1024, evaluated while the
//                        attention workgroups work)  +  ctx(t) (512 columns: the chain)
// Every wave owns an eighth of EACH part, so a late part is spread over all 8 waves of the workgroup.  Weight register j of a
// lane (wave w, k quarter kq) <-> logical column:
//   attention_rnn  j < 48: 192 w + 4 j + kq                      j >= 48: 1536 + 32 w + 4 (j - 48) + kq        [h_att|ctx] , [pre1]
//   decoder_rnn    j < 32: 128 w + 4 j + kq (h_att)   j < 64: 1536 + 128 w + 4 (j - 32) + kq (h_dec)   else: 1024 + 64 w + 4 (j - 64) + kq (ctx)
__device__ __forceinline__ int pd_ka(int wave, int j, int kq) { return j < 48 ? 192 * wave + 4 * j + kq : 1536 + 32 * wave + 4 * (j - 48) + kq; }
__device__ __forceinline__ int pd_kd(int wave, int j, int kq) {
    return j < 32 ? 128 * wave + 4 * j + kq : (j < 64 ? 1536 + 128 * wave + 4 * (j - 32) + kq : 1024 + 64 * wave + 4 * (j - 64) + kq);
}
// acc[b] += sum_{j in [J0, J1)} wreg[j] * x_b[column of j]   (DEC: decoder_rnn mapping; its h_dec columns sit 256 further in the LDS row)
template <int NJ, int J0, int J1, bool DEC>
__device__ __forceinline__ void pd_gemv_part(const float (&wreg)[NJ], const float* X, int B, float (&acc)[PD_MAXB]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kq = lane & 3;
#pragma unroll
    for (int b = 0; b < PD_MAXB; ++b) {
        if (b < B) {
            const float* xb = X + (size_t)b * PD_XW;
            float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
            for (int j = J0; j < J1; j += 2) {
                const int k0 = DEC ? pd_kd(wave, j, kq) : pd_ka(wave, j, kq), k1 = DEC ? pd_kd(wave, j + 1, kq) : pd_ka(wave, j + 1, kq);
                acc0 = fmaf(wreg[j], xb[k0 + ((DEC && j >= 32 && j < 64) ? 256 : 0)], acc0);
                acc1 = fmaf(wreg[j + 1], xb[k1 + ((DEC && j + 1 >= 32 && j + 1 < 64) ? 256 : 0)], acc1);
            }
            acc[b] += acc0 + acc1;
        }
    }
}
// quad sum of the lanes of a gate row -> red[wave][r][b]; the accumulators start the next frame at zero
__device__ __forceinline__ void pd_gemv_finish(float (&acc)[PD_MAXB], int B, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kq = lane & 3, r = lane >> 2;
#pragma unroll
    for (int b = 0; b < PD_MAXB; ++b) {
        if (b < B) {
            float v = acc[b];
            v = T2V_DPP_ADD(v, 0xB1);
            v = T2V_DPP_ADD(v, 0x4E);
            if (kq == 0) red[(wave * 16 + r) * PD_MAXB + b] = v;
        }
        acc[b] = 0.f;
    }
}

__global__ __launch_bounds__(PD_THREADS) void k_decode_persist(PersistArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wg = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int B = a.B, Tp = a.T_in;
    // ---- LDS carve (floats; sizes follow the runtime B and T_in: pd_lds_floats())
    const int Tcap = (Tp + 15) & ~15, TW = Tcap + 32;
    float* X = lds;                                      // [B][2816]
    float* red = X + B * PD_XW;                          // [8][16][MAXB]
    float* gst = red + 8 * 16 * PD_MAXB;                 // [MAXB][16] gate pre-activations
    float* cst = gst + PD_MAXB * 16;                     // [2][MAXB][4] cell states (attention_rnn, decoder_rnn)
    int* flag = (int*)(cst + 2 * PD_MAXB * 4);           // [4]
    float* role = (float*)(flag + 4);                    // role area
    // attention role
    float* wq_s = role;                                  // [16][1028]
    float* mem_s = wq_s + 16 * 1028;                     // [Tcap][64]
    float* pm_s = mem_s + Tcap * 64;                     // [Tcap][16]
    float* win = pm_s + Tcap * 16;                       // [2][TW]: alignment window, index x <-> position x - 15
    float* eall = win + 2 * TW;                          // [Tcap + T2V_CTX_PAD]: attention weights, zero from Tp on (t2v_ctx_partial)
    float* qv = eall + Tcap + T2V_CTX_PAD;               // [16]
    float* cred = qv + 16;                               // [8][64]   (round 6: a 2 KB reduction buffer nobody used any more sat here — three
                                                         //            utterances of 160 symbols now fit the 160 KB)
    float* rsm = cred + 8 * 64;                          // [32] row maxima
    float* rss = rsm + 32;                               // [32] row sums
    // projection / Prenet-1 roles (their own workgroups: alias the same area)
    float* prow_s = role;                                // [8][1536]
    float* w1_s = role;                                  // [8][256]

    const bool is_attn = wg < 8 * B;
    const int ab = wg >> 3, as = wg & 7;                 // attention item / slice
    const int gw = wg * 8 + wave;                        // global wave index
    const int prow = (wg >= PD_WG_PROJ && wg < PD_WG_PROJ + 43) ? (wg - PD_WG_PROJ) * 8 + wave : -1;      // projection row of this wave
    const bool is_proj = prow >= 0 && prow < PD_NROW;
    const bool wg_proj = prow >= 0;                       // whole workgroup (the last one has idle waves)
    const int p1row = (wg >= PD_WG_PRE1 && wg < PD_WG_PRE1 + 32) ? (wg - PD_WG_PRE1) * 8 + wave : -1;    // Prenet-1 row of this wave
    (void)gw;

    // ---- one-time loads: LSTM weights of this workgroup's 16 gate rows per cell into registers
    const int kq = lane & 3, r16 = lane >> 2;
    const int grow = (r16 & 3) * T2V_H + 4 * wg + (r16 >> 2);           // gate-major row of (unit 4wg + r16>>2, gate r16&3)
    float wa[PD_KATT / 32], wd[PD_KDEC / 32];
#pragma unroll
    for (int j = 0; j < PD_KATT / 32; ++j) {
        const int k = pd_ka(wave, j, kq);                                // [h_att | ctx | pre1]
        wa[j] = k < T2V_H ? a.w_hh_att[(size_t)grow * T2V_H + k]
                          : (k < T2V_KATT ? a.w_ih_att[(size_t)grow * 768 + T2V_PRE + (k - T2V_H)] : a.w_ih_att[(size_t)grow * 768 + (k - T2V_KATT)]);
    }
#pragma unroll
    for (int j = 0; j < PD_KDEC / 32; ++j) {
        const int k = pd_kd(wave, j, kq);                                // [h_att | ctx | h_dec]
        wd[j] = k < T2V_KATT ? a.w_ih_dec[(size_t)grow * T2V_KATT + k] : a.w_hh_dec[(size_t)grow * T2V_H + (k - T2V_KATT)];
    }
    float bias_a = 0.f, bias_d = 0.f;                                   // wave 0: thread (row r = tid & 15, item) holds the bias of row r
    if (tid < 64) {
        const int row = (tid & 3) * T2V_H + 4 * wg + ((tid & 15) >> 2);
        bias_a = a.bias_att[row];
        bias_d = a.bias_dec[row];
    }
    for (int i = tid; i < B * PD_XW; i += PD_THREADS) X[i] = 0.f;
    if (tid < 2 * PD_MAXB * 4) cst[tid] = 0.f;
    if (tid == 0) flag[0] = 1;
    // role operands
    // (round 4: the fused location filter's MFMA operand lives in LDS — 16 registers per lane that only the attention
    // workgroups used, next to 136 weight registers and the carried partial gate sums, made the kernel spill)
    float* areg_s = rss + 32;                              // [16 steps][64 lanes]
    float4 vr = make_float4(0.f, 0.f, 0.f, 0.f);
    if (is_attn) {
        for (int i = tid; i < 16 * 1024; i += PD_THREADS) wq_s[(i >> 10) * 1028 + (i & 1023)] = a.wq[(size_t)(16 * as) * 1024 + i];
        for (int i = tid; i < Tp * 64; i += PD_THREADS) mem_s[i] = a.memory[((size_t)ab * Tp + (i >> 6)) * T2V_E + 64 * as + (i & 63)];
        for (int i = tid; i < Tp * 16; i += PD_THREADS) pm_s[i] = a.pm[((size_t)ab * Tp + (i >> 4)) * T2V_A + 16 * as + (i & 15)];
        for (int i = tid; i < 2 * TW; i += PD_THREADS) win[i] = 0.f;
        for (int i = Tp + tid; i < Tcap + T2V_CTX_PAD; i += PD_THREADS) eall[i] = 0.f;
        const int g = lane >> 4, c16 = lane & 15;
        const float4* wp = (const float4*)(a.wcomb + (16 * as + c16) * 64 + 16 * g);
        if (wave == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 w4 = wp[u];
                areg_s[(4 * u) * 64 + lane] = w4.x; areg_s[(4 * u + 1) * 64 + lane] = w4.y;
                areg_s[(4 * u + 2) * 64 + lane] = w4.z; areg_s[(4 * u + 3) * 64 + lane] = w4.w;
            }
        }
        vr = *(const float4*)(a.v + 16 * as + 4 * g);
    } else if (prow >= 0) {
        if (is_proj)
            for (int i = lane; i < 1536; i += 64) prow_s[wave * 1536 + i] = a.proj_w[(size_t)prow * 1536 + i];
    } else if (p1row >= 0) {
        for (int i = lane; i < 256; i += 64) w1_s[wave * 256 + i] = a.w1[(size_t)p1row * 256 + i];
    }
    const float pbias = is_proj ? a.proj_b[prow] : 0.f;
    // Prenet of the go frame (frame 0 input)
    for (int i = tid; i < B * T2V_PRE; i += PD_THREADS) X[(i >> 8) * PD_XW + PD_X_P1 + (i & 255)] = a.pre_first[i];
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rx = pd_rsrc(a.xg);
    int nap_e = 0, nap_h = 0, nap_x = 0, nap_c = 0, nap_p = 0, nap_q = 0;      // adaptive naps in front of the polls (64-cycle units)
    // partial gate sums carried from where their inputs appear to where the gates are needed (see pd_gemv_part); frame 0
    // starts from h_att = ctx = h_dec = 0, i.e. from zero partial sums
    float ea[PD_MAXB], ed[PD_MAXB];
#pragma unroll
    for (int b = 0; b < PD_MAXB; ++b) ea[b] = ed[b] = 0.f;
    for (int t = 0; t < a.t_end; ++t) {
        const unsigned xcur = (unsigned)t * pd_row(B), xprev = xcur - pd_row(B);      // float offsets of this / the previous frame's row
        // ---- frame entry (t > 0): stop decision of the previous frame, Prenet output of the new frame's input
        if (t > 0) {
            // Prenet output of the new frame's input + stop decision of the previous frame.  Every workgroup of the chip wants
            // the same 1 KB (+ one word) at the same moment: with all 8 waves of all 256 workgroups polling it, the eight cache
            // lines behind it were a hot spot that took ~4 us to hand the values over (time line of tools/dbg_persist.py).  So
            // ONE wave per item polls (one 16-byte load per lane) and thread 0 alone watches the stop word.
            if (tid < 64 * B) {
                pd_u32x4 pv = {0u, 0u, 0u, 0u};
                unsigned spins = 0, sx = 0u;
                for (int i = 0; i < nap_e; i += 4) __builtin_amdgcn_s_sleep(4);
                for (;;) {
                    pv = pd_get4(rx, xprev + pd_pre1(B) + 4u * (unsigned)tid);          // [b][256] is contiguous
                    if (tid == 0) sx = pd_get(rx, xprev + pd_stop(B));
                    if (tid == 0 && sx == 2u) flag[0] = 2;              // the gate fired on the previous frame: nothing runs after it
                    if (__all(pv[0] != PD_SENT && pv[1] != PD_SENT && pv[2] != PD_SENT && pv[3] != PD_SENT && sx != PD_SENT)) break;
                    if (__any(tid == 0 && sx == 2u)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (pd_give_up(spins, a.err, flag)) break;
                }
                nap_e = pd_adapt(nap_e, (int)spins);
                const int i = 4 * tid;
                *(float4*)(X + (size_t)(i >> 8) * PD_XW + PD_X_P1 + (i & 255)) =
                    make_float4(__uint_as_float(pv[0]), __uint_as_float(pv[1]), __uint_as_float(pv[2]), __uint_as_float(pv[3]));
            }
            __syncthreads();
            if (flag[0] != 1) return;                          // stopped on the gate (2) or timed out (0)
        }
        PD_STAMP(0, 0); PD_STAMP(64, 10); PD_STAMP(128, 14);
        PD_RT(0);
        if (a.prof && wg == 0 && tid == 0 && (t == 100 || t == 600)) {       // steady-state frame period: 500 frames between two stamps
            a.prof[t == 100 ? 20 : 22] = __builtin_readcyclecounter();
            a.prof[t == 100 ? 21 : 23] = __builtin_amdgcn_s_memrealtime();
        }
        // ---- 1. attention_rnn(t): gates of this workgroup's 4 units, cell update, publish h_att
        pd_gemv_part<PD_KATT / 32, 48, 56, false>(wa, X, B, ea);      // the Prenet columns; [h_att | ctx] were added during frame t-1
        pd_gemv_finish(ea, B, red);
        __syncthreads();
        if (tid < 16 * B) {                                    // thread = (row r = tid & 15, item b = tid >> 4)
            const int r = tid & 15, b = tid >> 4;
            float s = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) s += red[(w8 * 16 + r) * PD_MAXB + b];
            gst[b * 16 + r] = s + bias_a;                      // pre-activation where the unit's thread finds its four gates
        }
        __syncthreads();
        if (tid < 4 * B) {                                     // thread = (unit u = tid & 3, item b = tid >> 2)
            const int u = tid & 3, b = tid >> 2;
            const float* gp = gst + b * 16 + 4 * u;
            const float gi = sigmoidf_(gp[0]), gf = sigmoidf_(gp[1]), gg = tanhf_(gp[2]), go = sigmoidf_(gp[3]);
            const float c = gf * cst[b * 4 + u] + gi * gg;
            cst[b * 4 + u] = c;
            // the four units of an item leave as ONE 16-byte store (round 4: four 4-byte stores from four lanes were four
            // write-through transactions into the same 32-byte sector)
            const float h = go * tanhf_(c);
            const float h1 = T2V_DPP_QUAD_F(h, 1), h2 = T2V_DPP_QUAD_F(h, 2), h3 = T2V_DPP_QUAD_F(h, 3);      // (used by lane u == 0 only)
            // (offset rebuilt from an opaque copy of the thread index: hoisted out of the frame loop it was spilled, and its reload from
            // scratch — a vector-memory load + s_waitcnt vmcnt(0) — sat in front of this store on every frame's chain; round 6)
            if (u == 0) pd_put4(rx, xcur + pd_hatt(B) + (unsigned)((pd_opaque(threadIdx.x) >> 2) * 1024 + 4 * wg), h, h1, h2, h3);
        }
        PD_STAMP(0, 1);
        PD_RT(1);
        // ---- 2. h_att(t) for everyone (attention slices need it now, the others for decoder_rnn)
        // location features of this frame's tiles (fused filter, K = 64): they depend on the PREVIOUS frame's weights only,
        // so the attention workgroups evaluate them while h_att(t) is still on its way (round 3, from the training kernel)
        f32x4 lacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (is_attn) {
            const int g = lane >> 4, c16 = lane & 15;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int jt = wave + 8 * i;
                if (16 * jt < Tp) {
                    float bop[16];
#pragma unroll
                    for (int st = 0; st < 16; ++st) {
                        const int kk = 4 * st + g;
                        bop[st] = win[(kk >> 5) * TW + 16 * jt + c16 + (kk & 31)];
                    }
                    f32x4 l0 = {0.f, 0.f, 0.f, 0.f}, l1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int st = 0; st < 16; st += 2) {
                        l0 = mfma16x4(areg_s[st * 64 + lane], bop[st], l0);
                        l1 = mfma16x4(areg_s[(st + 1) * 64 + lane], bop[st + 1], l1);
                    }
                    lacc[i] = l0 + l1;
                }
            }
        }
        {
            // (only the attention workgroups need h_att(t) NOW; everybody else uses it for decoder_rnn microseconds later and
            // comes for it late, with a fixed nap — 256 workgroups polling the same 32 lines the moment they land made this the
            // longest hand-off of the frame)
            const int rounds = pd_gather_items(X + PD_X_HA, rx, xcur + pd_hatt(B), 1024u, 1024, B, is_attn ? nap_h : 80, a.err, flag);
            if (is_attn) nap_h = pd_adapt(nap_h, rounds);
        }
        __syncthreads();
        if (flag[0] != 1) return;
        PD_STAMP(0, 2);
        PD_RT(2);
        // decoder_rnn(t), part 2: the h_att(t) columns — now, while the attention workgroups are busy (they do theirs behind
        // their context hand-off, in the shadow of the context gather)
        if (!is_attn) {
            pd_gemv_part<PD_KDEC / 32, 0, 32, true>(wd, X, B, ed);
            // ... and the h_dec(t-1) columns: the other workgroups fetch that row only now, a frame after it was published and
            // long after the projection workgroups (who needed it at once) are done with it
            if (t > 0 && !wg_proj) {
                (void)pd_gather_items(X + PD_X_HD, rx, xprev + pd_hdec(B), 1024u, 1024, B, 0, a.err, flag);
                __syncthreads();
                if (flag[0] != 1) return;
                pd_gemv_part<PD_KDEC / 32, 32, 64, true>(wd, X, B, ed);
            }
        }
        if (is_attn) {
            const int g = lane >> 4, c16 = lane & 15;
            const int len = a.lengths ? a.lengths[ab] : Tp;
            // query slice: thread = (dim d = tid >> 5, k part kq = tid & 31): k = 4 kq + 128 i, 16-byte LDS operands; 32-lane
            // sum = 16-lane DPP row sum + one cross-row exchange (one barrier instead of two and a 32-way tree)
            {
                const int d = tid >> 5, kq = tid & 31;
                const float* wrow = wq_s + d * 1028 + 4 * kq;
                const float* hx = X + (size_t)ab * PD_XW + PD_X_HA + 4 * kq;
                float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float4 w4 = *(const float4*)(wrow + 128 * i);
                    const float4 h4 = *(const float4*)(hx + 128 * i);
                    acc0 = fmaf(w4.x, h4.x, acc0); acc1 = fmaf(w4.y, h4.y, acc1);
                    acc0 = fmaf(w4.z, h4.z, acc0); acc1 = fmaf(w4.w, h4.w, acc1);
                }
                float q = row16_sum(acc0 + acc1);
                q = rows2_sum(q);
                if (kq == 0) qv[d] = q;
            }
            __syncthreads();
            const float4 q4 = make_float4(qv[4 * g], qv[4 * g + 1], qv[4 * g + 2], qv[4 * g + 3]);
            // partial energies of this slice: wave -> position tiles wave, wave + 8
            const unsigned exw = xcur + pd_ex(B) + (unsigned)((ab * 8 + as) * 256);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int jt = wave + 8 * i;
                if (16 * jt < Tp) {
                    const f32x4 acc = lacc[i];
                    const int j = 16 * jt + c16;
                    const float4 pm4 = *(const float4*)(pm_s + min(j, Tp - 1) * 16 + 4 * g);
                    const float s0 = tanhf_(q4.x + acc[0] + pm4.x), s1 = tanhf_(q4.y + acc[1] + pm4.y);
                    const float s2 = tanhf_(q4.z + acc[2] + pm4.z), s3 = tanhf_(q4.w + acc[3] + pm4.w);
                    float esum = vr.x * s0 + vr.y * s1 + vr.z * s2 + vr.w * s3;
                    esum += __shfl_xor(esum, 16, 64);
                    esum += __shfl_xor(esum, 32, 64);
                    if (g == 0 && j < Tp) pd_put(rx, exw + (unsigned)j, esum);
                }
            }
            PD_STAMP(0, 3);
            PD_RT(3);
            // gather the 8 partials of every position, masked softmax
            float ev0 = -INFINITY;
            if (tid < Tp) {
                const unsigned e0 = xcur + pd_ex(B) + (unsigned)(ab * 8 * 256 + tid);
                float p[8];
                unsigned spins = 0;
                for (int i = 0; i < nap_x; i += 4) __builtin_amdgcn_s_sleep(4);
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const unsigned x = pd_get(rx, e0 + (unsigned)(i * 256));
                        p[i] = __uint_as_float(x);
                        ok = ok && x != PD_SENT;
                    }
                    if (__all(ok)) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (pd_give_up(spins, a.err, flag)) break;
                }
                nap_x = pd_adapt(nap_x, (int)spins);
                const float ev = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
                ev0 = tid < len ? ev : -INFINITY;
            }
            {
                float mloc = ev0;
                mloc = T2V_DPP_MAX(mloc, 0xB1); mloc = T2V_DPP_MAX(mloc, 0x4E);
                mloc = T2V_DPP_MAX(mloc, 0x141); mloc = T2V_DPP_MAX(mloc, 0x140);
                mloc = rows4_max(mloc);
                if (lane == 0) rsm[wave] = mloc;
            }
            __syncthreads();
            if (flag[0] != 1) return;
            float m;
            {
                const float4 a0 = *(const float4*)rsm, a1 = *(const float4*)(rsm + 4);
                m = fmaxf(fmaxf(fmaxf(a0.x, a0.y), fmaxf(a0.z, a0.w)), fmaxf(fmaxf(a1.x, a1.y), fmaxf(a1.z, a1.w)));
            }
            const float e0v = tid < Tp ? expf(ev0 - m) : 0.f;
            {
                const float sloc = rows4_sum(row16_sum(e0v));
                if (lane == 0) rss[wave] = sloc;
            }
            __syncthreads();
            float ssum;
            {
                const float4 a0 = *(const float4*)rss, a1 = *(const float4*)(rss + 4);
                ssum = ((a0.x + a0.y) + (a0.z + a0.w)) + ((a1.x + a1.y) + (a1.z + a1.w));
            }
            const float al = e0v * (1.0f / ssum);
            if (tid < Tp) {
                eall[tid] = al;
                win[15 + tid] = al;                                         // previous weights of the next frame
                win[TW + 15 + tid] += al;                                   // cumulative weights
            }
            __syncthreads();
            PD_STAMP(0, 4);
            PD_RT(4);
            // context columns 64 as .. 64 as + 63: thread = (column c = tid & 63, part = tid >> 6)
            {
                const int c = tid & 63, part = tid >> 6;
                cred[part * 64 + c] = t2v_ctx_partial_v<PD_CTXW>(eall, mem_s, part, c, Tp);
            }
            __syncthreads();
            if (tid < 64) {
                float acc = 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += cred[u * 64 + tid];
                pd_put(rx, xcur + pd_ctx(B) + (unsigned)(ab * 512 + 64 * as + tid), acc);
            }
            // (the saved alignment row leaves AFTER the context hand-off: nothing sits in this CU's memory pipe in front of it)
            if (as == 0 && tid < Tp) a.AL[((size_t)(t + 1) * B + ab) * Tp + tid] = al;
        }
        PD_STAMP(0, 5); PD_STAMP(64, 11);
        PD_RT(5);
        if (is_attn) pd_gemv_part<PD_KDEC / 32, 0, 32, true>(wd, X, B, ed);      // (the context of the other items is in flight meanwhile)
        // ---- 3. ctx(t) for everyone, the context columns of decoder_rnn(t)
        {
            const int rounds = pd_gather_items(X + PD_X_CX, rx, xcur + pd_ctx(B), 512u, 512, B, nap_c, a.err, flag);
            nap_c = pd_adapt(nap_c, rounds);
        }
        __syncthreads();
        if (flag[0] != 1) return;
        PD_STAMP(0, 6);
        PD_RT(6);
        pd_gemv_part<PD_KDEC / 32, 64, 80, true>(wd, X, B, ed);
        pd_gemv_finish(ed, B, red);
        __syncthreads();
        if (tid < 16 * B) {
            const int r = tid & 15, b = tid >> 4;
            float s = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) s += red[(w8 * 16 + r) * PD_MAXB + b];
            gst[b * 16 + r] = s + bias_d;
        }
        __syncthreads();
        if (tid < 4 * B) {
            const int u = tid & 3, b = tid >> 2;
            const float* gp = gst + b * 16 + 4 * u;
            const float gi = sigmoidf_(gp[0]), gf = sigmoidf_(gp[1]), gg = tanhf_(gp[2]), go = sigmoidf_(gp[3]);
            const float c = gf * cst[PD_MAXB * 4 + b * 4 + u] + gi * gg;
            cst[PD_MAXB * 4 + b * 4 + u] = c;
            const float h = go * tanhf_(c);
            const float h1 = T2V_DPP_QUAD_F(h, 1), h2 = T2V_DPP_QUAD_F(h, 2), h3 = T2V_DPP_QUAD_F(h, 3);      // (used by lane u == 0 only)
            if (u == 0) pd_put4(rx, xcur + pd_hdec(B) + (unsigned)((pd_opaque(threadIdx.x) >> 2) * 1024 + 4 * wg), h, h1, h2, h3);
        }
        PD_STAMP(0, 7); PD_STAMP(64, 12);
        PD_RT(7);
        // attention_rnn(t+1), the [h_att(t) | ctx(t)] columns: in the shadow of the projection / Prenet stages (the projection
        // workgroups do theirs once their rows are out)
        if (!wg_proj) pd_gemv_part<PD_KATT / 32, 0, 48, false>(wa, X, B, ea);
        // ---- 4. projection rows (mel, gate, folded Prenet layer 0)
        if (prow >= 0) {            // whole workgroup takes the branch: barriers inside are uniform
            // the context columns of the rows first: h_dec(t) is still on its way
            float pacc[PD_MAXB];
#pragma unroll
            for (int b = 0; b < PD_MAXB; ++b) pacc[b] = 0.f;
            if (is_proj) {
                const float* wr = prow_s + wave * 1536;
#pragma unroll
                for (int b = 0; b < PD_MAXB; ++b) {
                    if (b < B) {
                        const float* xb = X + (size_t)b * PD_XW;
                        float acc = 0.f;
#pragma unroll
                        for (int i = 16; i < 24; ++i) acc = fmaf(wr[lane + 64 * i], xb[PD_X_CX + (lane + 64 * i - T2V_H)], acc);
                        pacc[b] = acc;
                    }
                }
            }
            // (round 6: the Prenet-0 dropout factors of this wave's row — a 64-bit counter hash per item, a function of (seed, frame, row)
            // alone — are evaluated HERE, in the shadow of the h_dec hand-off, not behind the dot product on the frame's chain)
            float drop0[PD_MAXB];
#pragma unroll
            for (int b = 0; b < PD_MAXB; ++b)
                drop0[b] = (is_proj && prow > T2V_NMEL && b < B)
                               ? t2v_drop_scale(a.seed, T2V_RNG_PRENET0, t + 1, (uint32_t)(b * T2V_PRE + (prow - (T2V_NMEL + 1))), a.p_prenet) : 0.f;
            {
                const int rounds = pd_gather_items(X + PD_X_HD, rx, xcur + pd_hdec(B), 1024u, 1024, B, nap_p, a.err, flag);
                nap_p = pd_adapt(nap_p, rounds);
            }
            __syncthreads();
            if (flag[0] != 1) return;
            if (is_proj) {
                const float* wr = prow_s + wave * 1536;
                bool all_fired = true;
                // (an opaque copy of the row number: the output addresses derived from it are rebuilt per frame instead of being
                // hoisted out of the frame loop, spilled, and reloaded from scratch in front of every store)
                int prow_o = prow;
                asm volatile("" : "+v"(prow_o));
                const int prow = prow_o;
#pragma unroll
                for (int b = 0; b < PD_MAXB; ++b) {
                    if (b >= B) continue;
                    const float* xb = X + (size_t)b * PD_XW;
                    float acc = pacc[b];
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc = fmaf(wr[lane + 64 * i], xb[PD_X_HD + lane + 64 * i], acc);      // the h_dec columns
                    acc = wave_sum_rl(acc) + pbias;
                    if (lane == 0) {
                        if (prow < T2V_NMEL) a.MEL[((size_t)t * B + b) * T2V_NMEL + prow] = acc;
                        else if (prow == T2V_NMEL) a.GATE[(size_t)t * B + b] = acc;
                        else {
                            const float pv = fmaxf(acc, 0.f) * drop0[b];
                            gst[b * 8 + wave] = pv;         // published below, eight rows with ONE store instruction
                        }
                    }
                    all_fired = all_fired && acc > a.gate_logit_thr;
                }
                if (prow == T2V_NMEL && lane == 0) {
                    // stop rule sigmoid(gate) > threshold for every item (model.py:453; B == 1 in the reference)
                    if (all_fired) atomicMin(a.stop_flag, t);
                    pd_put(rx, xcur + pd_stop(B), __uint_as_float(all_fired ? 2u : 1u));
                }
            }
            __syncthreads();        // X[HD] now holds h_dec(t): the next frame's decoder_rnn input for this workgroup
            // Prenet layer-0 rows of this workgroup: one coalesced 32-byte write per item instead of eight 4-byte writes from
            // eight waves — 32 single-word writes per cache line from all over the chip took microseconds to land
            if (tid < 8 * B) {
                const int b = tid >> 3, row = (wg - PD_WG_PROJ) * 8 + (tid & 7);
                if (row > T2V_NMEL && row < PD_NROW) {
                    const int to = pd_opaque(threadIdx.x);
                    pd_put(rx, xcur + pd_pre0(B) + (unsigned)((to >> 3) * 256 + (wg - PD_WG_PROJ) * 8 + (to & 7) - (T2V_NMEL + 1)), gst[b * 8 + (tid & 7)]);
                }
            }
            pd_gemv_part<PD_KATT / 32, 0, 48, false>(wa, X, B, ea);
        } else if (is_attn && t + 1 < a.t_end) {
            // decoder_rnn(t+1), part 1 (the h_dec(t) columns) of the ATTENTION workgroups: in the tail of this frame — at the top
            // of the next one they go straight from "h_att published" to their gather of it and the attention (the chain).  The
            // projection workgroups are gathering the same row right now and ARE the chain: let them go first.
            __builtin_amdgcn_s_sleep(48);
            (void)pd_gather_items(X + PD_X_HD, rx, xcur + pd_hdec(B), 1024u, 1024, B, 0, a.err, flag);
            __syncthreads();
            if (flag[0] != 1) return;
        }
        if ((is_attn || wg_proj) && t + 1 < a.t_end) pd_gemv_part<PD_KDEC / 32, 32, 64, true>(wd, X, B, ed);
        PD_STAMP(64, 13); PD_STAMP(128, 15);
        PD_RT(8);
        // ---- 5. Prenet layer 1 rows: wave 0 fetches pre0 for the whole workgroup (256 waves polling the same 1 KB were the
        //         other hot spot of the frame)
        if (p1row >= 0) {
            float* p0_s = w1_s + 8 * 256;                                  // [B][256]
            // (what does not depend on pre0 goes in front of the wait for it: the row's weights and its dropout factors)
            const float4 w4 = *(const float4*)(w1_s + wave * 256 + 4 * lane);
            float drop1[PD_MAXB];
#pragma unroll
            for (int b = 0; b < PD_MAXB; ++b)
                drop1[b] = b < B ? t2v_drop_scale(a.seed, T2V_RNG_PRENET1, t + 1, (uint32_t)(b * T2V_PRE + p1row), a.p_prenet) : 0.f;
            if (wave == 0) {
                unsigned spins = 0;
                for (int i = 0; i < nap_q; i += 4) __builtin_amdgcn_s_sleep(4);
                if (B == 1) {
                    const unsigned gq = xcur + pd_pre0(B) + (unsigned)(4 * lane);
                    pd_u32x4 x;
                    for (;;) {
                        x = pd_get4(rx, gq);
                        if (__all(x[0] != PD_SENT && x[1] != PD_SENT && x[2] != PD_SENT && x[3] != PD_SENT)) break;
                        __builtin_amdgcn_s_sleep(1);
                        if (pd_give_up(spins, a.err, flag)) break;
                    }
                    *(float4*)(p0_s + 4 * lane) = make_float4(__uint_as_float(x[0]), __uint_as_float(x[1]), __uint_as_float(x[2]), __uint_as_float(x[3]));
                } else {
                    // (all items in one polling pass, their loads in flight together: item by item every item paid its own round trip)
                    pd_u32x4 x[PD_MAXB];
                    for (;;) {
                        bool ok = true;
    #pragma unroll
                        for (int b = 0; b < PD_MAXB; ++b) {
                            if (b < B || b == 0) x[b] = pd_get4(rx, xcur + pd_pre0(B) + (unsigned)(min(b, B - 1) * 256 + 4 * lane));
                            else x[b] = x[0];
                            ok = ok && x[b][0] != PD_SENT && x[b][1] != PD_SENT && x[b][2] != PD_SENT && x[b][3] != PD_SENT;
                        }
                        if (__all(ok)) break;
                        __builtin_amdgcn_s_sleep(1);
                        if (pd_give_up(spins, a.err, flag)) break;
                    }
    #pragma unroll
                    for (int b = 0; b < PD_MAXB; ++b)
                        if (b < B)
                            *(float4*)(p0_s + b * 256 + 4 * lane) =
                                make_float4(__uint_as_float(x[b][0]), __uint_as_float(x[b][1]), __uint_as_float(x[b][2]), __uint_as_float(x[b][3]));
                }
                nap_q = pd_adapt(nap_q, (int)spins);
            }
            __syncthreads();
            if (flag[0] != 1) return;
#pragma unroll
            for (int b = 0; b < PD_MAXB; ++b) {
                if (b >= B) continue;
                const float4 xv = *(const float4*)(p0_s + b * 256 + 4 * lane);
                float acc = w4.x * xv.x;
                acc = fmaf(w4.y, xv.y, acc); acc = fmaf(w4.z, xv.z, acc); acc = fmaf(w4.w, xv.w, acc);
                acc = wave_sum_rl(acc);
                if (lane == 0) {
                    acc = fmaxf(acc, 0.f) * drop1[b];
                    gst[b * 8 + wave] = acc;
                }
            }
            __syncthreads();
            if (tid < 8 * B) {
                const int to = pd_opaque(threadIdx.x);
                pd_put(rx, xcur + pd_pre1(B) + (unsigned)((to >> 3) * 256 + (wg - PD_WG_PRE1) * 8 + (to & 7)), gst[tid]);
            }
        }
        PD_STAMP(128, 16); PD_STAMP(0, 8);
        PD_RT(9);
    }
}

static size_t pd_lds_bytes(int B, int T_in) {
    const size_t Tcap = (size_t)((T_in + 15) / 16) * 16;
    size_t f = (size_t)B * PD_XW + 8 * 16 * PD_MAXB + PD_MAXB * 16 + 2 * PD_MAXB * 4 + 4;
    const size_t attn = 16 * 1028 + Tcap * 64 + Tcap * 16 + 2 * (Tcap + 32) + Tcap + T2V_CTX_PAD + 16 + 8 * 64 + 64 + 16 * 64;
    const size_t proj = 8 * 1536;
    f += attn > proj ? attn : proj;
    return f * sizeof(float);
}
#define PD_LDS_MAX (160 * 1024)

// exchange scratch: one sentinel-filled row per frame
extern "C" long t2v_decoder_persist_scratch_floats(int B, int t_end) {
    return (B < 1 || B > PD_MAXB || t_end < 1) ? 0 : (long)((size_t)t_end * pd_row(B));
}
__global__ void k_pd_fill(uint4* p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_uint4(PD_SENT, PD_SENT, PD_SENT, PD_SENT);
}
// The workgroups of this kernel spin on each other's granules, so ALL 256 must be resident at once: one per CU on a
// device with >= 256 CUs whose occupancy query admits this launch configuration (512 threads, the run-time LDS carve).
// Answered once per device and (B, T_in) class; a device that is shared with other work can still fail to co-schedule
// them — the bounded spins then set the error word and Decoder.inference re-runs the utterance on the launch-per-stage
// loop.
static int pd_device_ok(size_t lds) {
    static int cus = -1;
    if (cus < 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
        cus = prop.multiProcessorCount;
    }
    if (cus < T2V_NWG) return 0;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_decode_persist, hipFuncAttributeMaxDynamicSharedMemorySize, PD_LDS_MAX) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
        attr_set = true;
    }
    int nblk = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, (const void*)k_decode_persist, PD_THREADS, lds) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return nblk >= 1;
}

extern "C" int t2v_decoder_persist_supported(int B, int T_in) {
    if (!(B >= 1 && B <= PD_MAXB && T_in >= 1 && T_in <= PD_MAXT && pd_lds_bytes(B, T_in) <= PD_LDS_MAX)) return 0;
    return pd_device_ok(pd_lds_bytes(B, T_in));
}

extern "C" int t2v_decoder_infer_persistent(const t2v_dec_persist_weights* w, const t2v_dec_persist_bufs* s, int B, int T_in,
                                            int t_end, float gate_threshold, float p_prenet, uint64_t seed, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!w || !s || !t2v_decoder_persist_supported(B, T_in) || t_end < 1) return T2V_ERR_ARG;
    if (!w->w_ih_att || !w->w_hh_att || !w->w_ih_dec || !w->w_hh_dec || !w->bias_att || !w->bias_dec || !w->wq || !w->wcomb ||
        !w->v || !w->proj_w || !w->proj_b || !w->prenet_w1 || !s->memory || !s->pm || !s->pre_first || !s->MEL || !s->GATE ||
        !s->AL || !s->stop_flag || !s->exchange || !s->err_word)
        return T2V_ERR_ARG;
    if (((uintptr_t)s->exchange & 15) || (size_t)t_end * pd_row(B) * 4 >= 0x7fffffffull) return T2V_ERR_ARG;      // 31-bit buffer offsets
    const size_t lds = pd_lds_bytes(B, T_in);     // (t2v_decoder_persist_supported raised the dynamic-LDS limit)
    k_pd_fill<<<1024, 256, 0, stream>>>((uint4*)s->exchange, (size_t)t_end * pd_row(B) / 4);
    (void)hipMemsetAsync(s->err_word, 0, sizeof(unsigned), stream);
    PersistArgs a;
    a.w_ih_att = w->w_ih_att; a.w_hh_att = w->w_hh_att; a.w_ih_dec = w->w_ih_dec; a.w_hh_dec = w->w_hh_dec;
    a.bias_att = w->bias_att; a.bias_dec = w->bias_dec; a.wq = w->wq; a.wcomb = w->wcomb; a.v = w->v;
    a.proj_w = w->proj_w; a.proj_b = w->proj_b; a.w1 = w->prenet_w1;
    a.memory = s->memory; a.pm = s->pm; a.lengths = s->lengths; a.pre_first = s->pre_first;
    a.MEL = s->MEL; a.GATE = s->GATE; a.AL = s->AL; a.stop_flag = s->stop_flag;
    a.xg = (float*)s->exchange; a.err = s->err_word;
    a.B = B; a.T_in = T_in; a.t_end = t_end;
    a.gate_logit_thr = gate_threshold <= 0.f ? -INFINITY : (gate_threshold >= 1.f ? INFINITY : logf(gate_threshold / (1.f - gate_threshold)));
    a.p_prenet = p_prenet; a.seed = seed;
    a.prof = g_t2v_prof;
    k_decode_persist<<<T2V_NWG, PD_THREADS, lds, stream>>>(a);
    return t2v_check_launch();
}
