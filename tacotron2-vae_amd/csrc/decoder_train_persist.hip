// Teacher-forced decoder recurrence, forward, as ONE persistent launch (reference Decoder.forward's time loop
// model.py:415-421 -> Decoder.decode 346-389 -> Attention.forward 67-88): 256 workgroups x 512 threads stay resident for
// all T_out steps, the 67 MB of LSTM weights live in registers, nothing is streamed per step.  The launch-per-step path
// (decoder_fwd.hip + attn_fwd.hip: 2 launches and a 67 MB weight stream per step) stays the general path; this kernel
// serves B <= 6 and T_in <= 224 — the headline training shape (B = 6 per GPU, T_in = 84).
//
// Roles (one workgroup per CU, all co-resident):
//   T  : workgroups [0, 8B)   — location-sensitive attention of item b = wg / 8, attention dims [16s, 16s + 16) and context
//        columns [64s, 64s + 64), s = wg % 8.  W_q slice, memory / processed-memory slices and the alignment window stay
//        in LDS for all steps (as in decoder_persist.hip).
//   L  : workgroups [8B, 256) — LSTM rows of BOTH cells: workgroup j owns hidden units [j*1024/NL, (j+1)*1024/NL) (4 or 5
//        units = 16 or 20 gate rows per cell), weights in VGPRs (thread = (gate, 1/128 of K): 60 + 100 registers).
// Teacher forcing takes decoder_rnn off the critical path: attention_rnn(t+1) needs only h_att(t) and ctx(t) (the Prenet
// term gpre is precomputed for all steps), so the per-step chain is
//     A-gemv(t) -> h_att hop -> attention(t) -> ctx hop -> A-gemv(t+1)
// while every L workgroup runs D-gemv(t-1) and its own gathers in the shadow of attention(t).
//
// Hand-off = the saved activations themselves.  Every recurrent value is produced exactly once per pass, so it needs no
// tag and no flag: the exchange buffer G (one row per step, laid out [plane][k][4 items] so that a consumer's 16-byte
// load is an LDS-ready operand) is pre-filled with a NaN sentinel (0xFFFFFFFF, never produced by arithmetic); producers
// store with sc1 (write-through), consumers poll their own words with sc1 loads until none is the sentinel.  4 bytes per
// value on the wire instead of an 8-byte {value, tag} granule, 16-byte loads, no memory ordering needed (MI355X guide
// G16 form R2 with an implicit tag).  The partial energies of the 8 attention slices of an item travel the same way (EX).
// The backward pass reads the usual arena (XS, CA, CD, GA, GD, AL, ACUM, S), written here with plain stores.
#include <stdlib.h>
#include "t2v_common.h"
#include "t2v_kernels.h"

#define PT_THREADS 512
#define PT_MAXB 6
#define PT_MAXT 224                  // LDS-resident W_q / processed-memory slices up to here
#define PT_MAXT_LONG 560             // register-resident ones beyond (k_dec_train_persist<.., true>)
#define PT_NTI_LONG 5                // 16-position tiles per wave of the long form: 8 waves x 5 x 16 = 640 >= 560
#define PT_MAXU 5
#define PT_SPIN 1500000u
#define PT_SENT 0xFFFFFFFFu
#define PT_JA (T2V_KATT / 128)       // 12 k per thread, attention_rnn  [h_att | ctx]
#define PT_JD (T2V_XW / 128)         // 20 k per thread, decoder_rnn    [h_att | ctx | h_dec]

struct PTArgs {
    const float* w_ih_att; const float* w_hh_att; const float* w_ih_dec; const float* w_hh_dec;
    const float* bias_dec; const float* wq; const float* wcomb; const float* v;
    const float* gpre; const float* memory; const float* pm; const int32_t* lengths;
    float* XS; float* CA; float* CD; float* GA; float* GD; float* AL; float* ACUM; float* S;
    float* G;                 // (T+2) rows x NP planes x 2560 k x 4 items, sentinel-filled
    float* EX;                // T x B x 8 x Tcap partial energies, sentinel-filled
    unsigned* err;
    int B, T_in, T_out;
    float p_att, p_dec;
    uint64_t seed;
    const t2v_step_params* step;
    unsigned long long* prof;   // optional: stamps of step T_out/2 (L workgroup 8B: slots 0..7, T workgroup 0: slots 8..15)
};
// wall clock (100 MHz, the same counter on every CU): hop latencies between workgroups
#define PT_WALL(COND, I) do { if (a.prof && (COND) && tid == 0) a.prof[(I)] = wall_clock64(); } while (0)
#define PT_STAMP(COND, I) do { if (a.prof && (COND) && tid == 0) a.prof[(I)] = __builtin_readcyclecounter(); } while (0)

// 16-byte / 4-byte accesses at agent scope (sc1: bypass the CU's vector L1, stores are write-through) as raw buffer
// operations: the compiler tracks their vmcnt itself (an inline-asm load is invisible to its scoreboard — the result
// registers can be read or copied before the data has landed).  Offsets are BYTES from the buffer base (< 2 GiB).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define PT_SC1 16
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pt_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ f32x4 pt_ld16(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, PT_SC1);
    return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ void pt_st16(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)off, 0, PT_SC1);
}
__device__ __forceinline__ void pt_st4(__amdgpu_buffer_rsrc_t r, unsigned off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)off, 0, PT_SC1);
}
__device__ __forceinline__ unsigned pt_ld4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, PT_SC1);
}
__device__ __forceinline__ bool pt_valid(f32x4 v, int nw) {      // the first nw words are not the sentinel
    bool ok = __float_as_uint(v[0]) != PT_SENT;
    ok = ok && (nw < 2 || __float_as_uint(v[1]) != PT_SENT);
    ok = ok && (nw < 3 || __float_as_uint(v[2]) != PT_SENT);
    ok = ok && (nw < 4 || __float_as_uint(v[3]) != PT_SENT);
    return ok;
}

// Gather nk columns [k0, k0 + nk) of all planes of one G row (byte offset row_off) into the LDS state planes.  PER
// chunks per thread (NP * nk <= PER * 512).  A chunk is valid when the words of the items it carries are all written.
// Returns through *flag (LDS, stays 1 unless a spin timed out); the caller syncs before reading X.
template <int PER, int NP>
__device__ __forceinline__ int pt_gather(f32x4* X, __amdgpu_buffer_rsrc_t rG, unsigned row_off, int k0, int nk, int B, int nap,
                                         unsigned* err, int* flag, int gap_at = 1 << 30, int gap = 0) {
    const int tid = threadIdx.x;
    unsigned src[PER];
    int dsti[PER], nw[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        int c = tid + PT_THREADS * u;
        const bool on = c < NP * nk;
        c = on ? c : 0;
        const int pl = c / nk;
        int kk = k0 + (c - pl * nk);
        kk += kk >= gap_at ? gap : 0;            // two column segments in one pass: [k0, gap_at) and [gap_at + gap, ..)
        src[u] = row_off + (unsigned)(pl * T2V_XW + kk) * 16u;
        dsti[u] = on ? pl * T2V_XW + kk : -1;
        nw[u] = min(4, B - 4 * pl);
    }
    // nap first (s_sleep units of 64 clocks): the caller passes what the previous steps' waits suggested, so that a
    // known-long wait issues no load at all; then the payload itself is polled — the first round that finds every
    // word written IS the gather (no separate flag round trip).  Rounds are counted per wave (wave-uniform loop).
    for (int i = 0; i < nap; i += 8) __builtin_amdgcn_s_sleep(8);
    f32x4 v[PER];
    int rounds = 0;
    for (;;) {
#pragma unroll
        for (int u = 0; u < PER; ++u) v[u] = pt_ld16(rG, src[u]);
        bool ok = true;
#pragma unroll
        for (int u = 0; u < PER; ++u) ok = ok && pt_valid(v[u], nw[u]);
        if (__all(ok)) break;
        __builtin_amdgcn_s_sleep(2);
        if (++rounds > (int)(PT_SPIN / 4) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = 0;
            break;
        }
    }
#pragma unroll
    for (int u = 0; u < PER; ++u)
        if (dsti[u] >= 0) X[dsti[u]] = v[u];
    return rounds;
}

// Sparse wait ahead of a bulk gather: threads [0, npoll) each watch ONE word (one per producing workgroup); everybody
// else parks at the block barrier.  Hundreds of threads per CU polling the payload itself cost the producers' stores and
// the other consumers' loads (MI355X guide: polling-cost row) — this way a CU issues npoll 4-byte loads per round.
// nap: s_sleep units (64 clocks each) to spend before the first look — the caller passes what the previous step's wait
// suggested, so that most of a known-long wait issues no load at all.  Returns the rounds it took.
__device__ __forceinline__ int pt_wait_words(__amdgpu_buffer_rsrc_t r, unsigned off, int npoll, int nap, unsigned* err, int* flag) {
    const int tid = threadIdx.x;
    if (tid < 64)
        for (int i = 0; i < nap; i += 16) __builtin_amdgcn_s_sleep(16);
    int rounds = 0;
    unsigned spins = 0;
    for (;;) {
        bool ok = true;
        if (tid < npoll) ok = pt_ld4(r, off) != PT_SENT;
        if (__syncthreads_and(ok)) break;
        ++rounds;
        if (tid == 0 && (++spins > PT_SPIN / 8 || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = 0;
        }
        __syncthreads();
        if (*flag != 1) break;
    }
    return rounds;
}

// acc[u][pair] += w[u] * x[pair] for the 5 unit rows of a thread and the item PAIRS (0,1), (2,3)[, (4,5)] as ONE volatile
// asm block of packed FMAs.  (a) v_pk_fma_f32 does two fp32 FMAs per lane in the 4 cycles a plain v_fma_f32 needs for one
// — measured here: the scalar-FMA version of this loop ran at 4.8 cycles per FMA and SIMD — with the weight broadcast to
// both halves through op_sel (weights of two consecutive k share a register pair: even k = low word, odd k = high word);
// (b) volatile asm, because left to itself the compiler sinks the (pure) FMAs of the fully unrolled k loop to the end of
// the function, first loads all operand vectors (80 registers) and then walks one accumulator at a time as a dependent
// chain — with 160 weight registers live that spills, and the chains stall.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NB>
struct PTAcc {
    static constexpr int NPAIR = NB > 4 ? 3 : 2;
    f32x2 p[PT_MAXU][NPAIR];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int u = 0; u < PT_MAXU; ++u)
#pragma unroll
            for (int i = 0; i < NPAIR; ++i) p[u][i] = f32x2{0.f, 0.f};
    }
};
template <bool ODD>
__device__ __forceinline__ void pt_pkfma(PTAcc<6>& acc, f32x2 w0, f32x2 w1, f32x2 w2, f32x2 w3, f32x2 w4, f32x2 x01, f32x2 x23, f32x2 x45) {
    if (ODD)
        asm volatile("v_pk_fma_f32 %0, %15, %20, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %1, %15, %21, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %2, %15, %22, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %3, %16, %20, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %4, %16, %21, %4 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %5, %16, %22, %5 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %6, %17, %20, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %7, %17, %21, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %8, %17, %22, %8 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %9, %18, %20, %9 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %10, %18, %21, %10 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %11, %18, %22, %11 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %12, %19, %20, %12 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %13, %19, %21, %13 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %14, %19, %22, %14 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                 : "+v"(acc.p[0][0]), "+v"(acc.p[0][1]), "+v"(acc.p[0][2]), "+v"(acc.p[1][0]), "+v"(acc.p[1][1]), "+v"(acc.p[1][2]), "+v"(acc.p[2][0]), "+v"(acc.p[2][1]), "+v"(acc.p[2][2]), "+v"(acc.p[3][0]), "+v"(acc.p[3][1]), "+v"(acc.p[3][2]), "+v"(acc.p[4][0]), "+v"(acc.p[4][1]), "+v"(acc.p[4][2])
                 : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4), "v"(x01), "v"(x23), "v"(x45));
    else
        asm volatile("v_pk_fma_f32 %0, %15, %20, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                 "v_pk_fma_f32 %1, %15, %21, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                 "v_pk_fma_f32 %2, %15, %22, %2 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                 "v_pk_fma_f32 %3, %16, %20, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                 "v_pk_fma_f32 %4, %16, %21, %4 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                 "v_pk_fma_f32 %5, %16, %22, %5 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                 "v_pk_fma_f32 %6, %17, %20, %6 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                 "v_pk_fma_f32 %7, %17, %21, %7 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                 "v_pk_fma_f32 %8, %17, %22, %8 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                 "v_pk_fma_f32 %9, %18, %20, %9 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                 "v_pk_fma_f32 %10, %18, %21, %10 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                 "v_pk_fma_f32 %11, %18, %22, %11 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                 "v_pk_fma_f32 %12, %19, %20, %12 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                 "v_pk_fma_f32 %13, %19, %21, %13 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                 "v_pk_fma_f32 %14, %19, %22, %14 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                 : "+v"(acc.p[0][0]), "+v"(acc.p[0][1]), "+v"(acc.p[0][2]), "+v"(acc.p[1][0]), "+v"(acc.p[1][1]), "+v"(acc.p[1][2]), "+v"(acc.p[2][0]), "+v"(acc.p[2][1]), "+v"(acc.p[2][2]), "+v"(acc.p[3][0]), "+v"(acc.p[3][1]), "+v"(acc.p[3][2]), "+v"(acc.p[4][0]), "+v"(acc.p[4][1]), "+v"(acc.p[4][2])
                 : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4), "v"(x01), "v"(x23), "v"(x45));
}
template <bool ODD>
__device__ __forceinline__ void pt_pkfma(PTAcc<4>& acc, f32x2 w0, f32x2 w1, f32x2 w2, f32x2 w3, f32x2 w4, f32x2 x01, f32x2 x23, f32x2) {
    if (ODD)
        asm volatile("v_pk_fma_f32 %0, %10, %15, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %1, %10, %16, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %2, %11, %15, %2 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %3, %11, %16, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %4, %12, %15, %4 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %5, %12, %16, %5 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %6, %13, %15, %6 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %7, %13, %16, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %8, %14, %15, %8 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                 "v_pk_fma_f32 %9, %14, %16, %9 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                 : "+v"(acc.p[0][0]), "+v"(acc.p[0][1]), "+v"(acc.p[1][0]), "+v"(acc.p[1][1]), "+v"(acc.p[2][0]), "+v"(acc.p[2][1]), "+v"(acc.p[3][0]), "+v"(acc.p[3][1]), "+v"(acc.p[4][0]), "+v"(acc.p[4][1])
                 : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4), "v"(x01), "v"(x23));
    else
        asm volatile("v_pk_fma_f32 %0, %10, %15, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                 "v_pk_fma_f32 %1, %10, %16, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                 "v_pk_fma_f32 %2, %11, %15, %2 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                 "v_pk_fma_f32 %3, %11, %16, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                 "v_pk_fma_f32 %4, %12, %15, %4 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                 "v_pk_fma_f32 %5, %12, %16, %5 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                 "v_pk_fma_f32 %6, %13, %15, %6 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                 "v_pk_fma_f32 %7, %13, %16, %7 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                 "v_pk_fma_f32 %8, %14, %15, %8 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                 "v_pk_fma_f32 %9, %14, %16, %9 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                 : "+v"(acc.p[0][0]), "+v"(acc.p[0][1]), "+v"(acc.p[1][0]), "+v"(acc.p[1][1]), "+v"(acc.p[2][0]), "+v"(acc.p[2][1]), "+v"(acc.p[3][0]), "+v"(acc.p[3][1]), "+v"(acc.p[4][0]), "+v"(acc.p[4][1])
                 : "v"(w0), "v"(w1), "v"(w2), "v"(w3), "v"(w4), "v"(x01), "v"(x23));
}

// acc += sum_{j in [J0, J1)} w[.][j] * x[kp + 128 j][.]: per k one 16-byte (+ one 8-byte) LDS operand, PT_PF of them in
// flight; w2[u][j / 2] holds the weights of k-blocks j (even, low word) and j + 1 (high word)
#define PT_PF 3
template <int NJ2, int J0, int J1, int NB>
__device__ __forceinline__ void pt_fma_range(const f32x2 (&w2)[PT_MAXU][NJ2], const f32x4* X, int kp, PTAcc<NB>& acc) {
    f32x4 xq[PT_PF];
    f32x2 yq[PT_PF];
#pragma unroll
    for (int d = 0; d < PT_PF; ++d) {
        yq[d] = f32x2{0.f, 0.f};
        if (J0 + d < J1) {
            xq[d] = X[kp + 128 * (J0 + d)];
            if constexpr (NB > 4) yq[d] = *(const f32x2*)&X[T2V_XW + kp + 128 * (J0 + d)];
        }
    }
#pragma unroll
    for (int j = J0; j < J1; ++j) {
        const f32x4 xc = xq[(j - J0) % PT_PF];
        const f32x2 yc = yq[(j - J0) % PT_PF];
        const f32x2 x01 = {xc[0], xc[1]}, x23 = {xc[2], xc[3]};
        if (j & 1) pt_pkfma<true>(acc, w2[0][j / 2], w2[1][j / 2], w2[2][j / 2], w2[3][j / 2], w2[4][j / 2], x01, x23, yc);
        else pt_pkfma<false>(acc, w2[0][j / 2], w2[1][j / 2], w2[2][j / 2], w2[3][j / 2], w2[4][j / 2], x01, x23, yc);
        if (j + PT_PF < J1) {
            xq[(j - J0) % PT_PF] = X[kp + 128 * (j + PT_PF)];
            if constexpr (NB > 4) yq[(j - J0) % PT_PF] = *(const f32x2*)&X[T2V_XW + kp + 128 * (j + PT_PF)];
        }
    }
}

// 16-lane row sums of the 5 x NB accumulators as a TRANSPOSING butterfly: 32 values (index u * NB + b, zero padded) are
// halved four times — at every stage a lane keeps the half selected by one bit of its lane id and adds its partner's
// copy of that half (row_mirror, row_half_mirror, quad xor 2, quad xor 1: the partner always differs in the selector
// bit) — 30 DPP adds + 60 selects instead of 120 dependent DPP adds + 30 masked stores.  Lane c of a row ends with the
// row sums of values 2c and 2c + 1 and stores them as one float2: red[gate][partial = 2 waves x 4 rows][32].
#define PT_DPP_F(v, CTRL) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (CTRL), 0xF, 0xF, true))
template <int NB>
__device__ __forceinline__ void pt_reduce_store(const PTAcc<NB>& acc, float* red) {
    const int tid = threadIdx.x, lane = tid & 63, g = tid >> 7;
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = 0.f;
#pragma unroll
    for (int u = 0; u < PT_MAXU; ++u)
#pragma unroll
        for (int i = 0; i < PTAcc<NB>::NPAIR; ++i) {
            v[u * NB + 2 * i] = acc.p[u][i][0];
            v[u * NB + 2 * i + 1] = acc.p[u][i][1];
        }
    const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
    float w16[16], w8[8], w4[4], w2[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float keep = b3 ? v[16 + i] : v[i], send = b3 ? v[i] : v[16 + i];
        w16[i] = keep + PT_DPP_F(send, 0x140);            // row_mirror
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float keep = b2 ? w16[8 + i] : w16[i], send = b2 ? w16[i] : w16[8 + i];
        w8[i] = keep + PT_DPP_F(send, 0x141);             // row_half_mirror
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float keep = b1 ? w8[4 + i] : w8[i], send = b1 ? w8[i] : w8[4 + i];
        w4[i] = keep + PT_DPP_F(send, 0x4E);              // quad_perm [2,3,0,1]
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float keep = b0 ? w4[2 + i] : w4[i], send = b0 ? w4[i] : w4[2 + i];
        w2[i] = keep + PT_DPP_F(send, 0xB1);              // quad_perm [1,0,3,2]
    }
    const int part = ((tid >> 6) & 1) * 4 + (lane >> 4);
    *(float2*)(red + (g * 8 + part) * 32 + 2 * (lane & 15)) = make_float2(w2[0], w2[1]);
}

// NB — 4: B <= 4 (one item plane), 6: B = 5, 6 (two planes).  LONG — the attention role for 224 < T_in <= 560 (koemo's longest
// sentence has 555 symbols): the W_q slice and the processed-memory slice move from LDS into registers (32 + 20 per thread: an
// attention workgroup holds no LSTM weights), which leaves the 160 KB to the memory slice (560 x 64 floats = 140 KB); five
// position tiles per wave instead of two, two positions per thread in the softmax.
template <int NB, bool LONG>
__global__ __launch_bounds__(PT_THREADS) void k_dec_train_persist(PTArgs a) {
    constexpr int NP = NB > 4 ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const uint64_t seed = t2v_step_seed(a.seed, a.step);
    const int wg = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int B = a.B, Tp = a.T_in, T = a.T_out;
    const int NT = 8 * B, NL = T2V_NWG - NT;
    const unsigned grow_b = (unsigned)NP * T2V_XW * 16u;      // bytes per G row
    const __amdgpu_buffer_rsrc_t rG = pt_rsrc(a.G), rE = pt_rsrc(a.EX);
    const int Tcap = (Tp + 15) & ~15;

    if (wg >= NT) {
        // =========================================================================== L role: LSTM rows of both cells
        f32x4* X = (f32x4*)lds;                              // [NP][2560] state planes: X[pl*2560 + k] = items 4pl..4pl+3 of column k
        float* red = lds + (size_t)NP * T2V_XW * 4;          // [4 gates][8 partials][32]
        float* cst = red + 4 * 8 * 32;                       // [2 cells][32] cell states, [4][32] decoder_rnn biases
        float* fdrop = cst + 6 * 32;                         // [2][32] dropout factors (cell state, hidden state) of attention_rnn's coming step
        int* flag = (int*)(cst + 8 * 32);
        const int j = wg - NT;
        const int u0 = (j * T2V_H) / NL, nu = ((j + 1) * T2V_H) / NL - u0;      // 4 or 5 units
        const int g = tid >> 7, kp = tid & 127;
        f32x2 wa[PT_MAXU][PT_JA / 2], wd[PT_MAXU][PT_JD / 2];     // [u][j / 2][j & 1]: weight of k = kp + 128 j
#pragma unroll
        for (int u = 0; u < PT_MAXU; ++u) {
            const bool on = u < nu;
            const size_t row = (size_t)g * T2V_H + u0 + (on ? u : 0);
#pragma unroll
            for (int jj = 0; jj < PT_JA; ++jj) {          // [h_att | ctx]: weight_hh, then weight_ih columns 256..767
                const int k = kp + 128 * jj;
                const float w = jj < 8 ? a.w_hh_att[row * T2V_H + k] : a.w_ih_att[row * (T2V_PRE + T2V_E) + T2V_PRE + (k - T2V_H)];
                wa[u][jj / 2][jj & 1] = on ? w : 0.f;
            }
#pragma unroll
            for (int jj = 0; jj < PT_JD; ++jj) {          // [h_att | ctx | h_dec]: weight_ih, then weight_hh
                const int k = kp + 128 * jj;
                const float w = jj < 12 ? a.w_ih_dec[row * T2V_KATT + k] : a.w_hh_dec[row * T2V_H + (k - T2V_KATT)];
                wd[u][jj / 2][jj & 1] = on ? w : 0.f;
            }
        }
        for (int i = tid; i < NP * T2V_XW; i += PT_THREADS) X[i] = f32x4{0.f, 0.f, 0.f, 0.f};      // row 0: zero initial states
        if (tid == 0) flag[0] = 1;
        // cell threads: tid = u * NB + b (wave 0); each keeps its two cell states in registers for the whole pass
        const int cu = tid / NB, cb = tid - cu * NB;
        const bool cell_on = tid < PT_MAXU * NB && cu < nu && cb < B;
        const int U = u0 + (cu < nu ? cu : 0);
        if (tid < 64) { cst[tid] = 0.f; }
        if (cell_on) {
#pragma unroll
            for (int r = 0; r < 4; ++r) cst[64 + 32 * r + tid] = a.bias_dec[r * T2V_H + U];
        }
        __syncthreads();
        int ctx_nap = 0;
        PTAcc<NB> accA;
        accA.clear();                  // row 0 is zero: the h_att half of attention_rnn(0) is zero
        // State-dropout factors of attention_rnn's cell of the COMING step (round 6): two 64-bit counter hashes (~300 cycles each) that
        // depend on (seed, step, unit) alone.  They used to sit behind the gate sums, between "context arrived" and "h_att published";
        // now the cell threads evaluate them BEFORE they poll for the context and park them in LDS (as two more live registers next
        // to the 160 weight registers they spill into the FMA phases and the step gets slower: measured).  Same-box A/B of the whole
        // training step: 10.86-10.90 -> 10.75-10.79 ms.  (All four factors of a step on the idle wave 1 during the cell phases —
        // built, parity-green: 10.93-10.98 ms; the decoder_rnn cell is not on the chain and the extra scratch costs more.)
        if (cell_on) { fdrop[tid] = 1.0f; fdrop[32 + tid] = t2v_drop_scale(seed, T2V_RNG_ATT_H, 0, (uint32_t)cb * T2V_H + U, a.p_att); }

        for (int t = 0; t <= T; ++t) {
            const bool do_att = t < T, do_dec = t >= 1;
            const unsigned grow = (unsigned)(t + 1) * grow_b;      // row t+1 = [h_att(t) | ctx(t) | h_dec(t-1)] (byte offset)
            PT_STAMP(wg == NT && t == T / 2, 0);
            // ---- attention_rnn(t): X holds row t = [h_att(t-1) | ctx(t-1) | h_dec(t-2)]; the h_att half of the sum was
            // accumulated into accA before ctx(t-1) arrived (end of the previous iteration), the ctx half is left
            if (do_att) {
                float gp[4] = {0.f, 0.f, 0.f, 0.f};
                if (cell_on) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) gp[r] = a.gpre[((size_t)t * B + cb) * T2V_G + r * T2V_H + U];
                }
                pt_fma_range<PT_JA / 2, 8, PT_JA, NB>(wa, X, kp, accA);
                pt_reduce_store<NB>(accA, red);
                PT_STAMP(wg == NT && t == T / 2, 16);
                __syncthreads();
                PT_STAMP(wg == NT && t == T / 2, 17);
                if (wave == 0) {
                    float hd = 0.f;
                    if (cell_on) {
                        float s[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float* rp = red + (r * 8) * 32 + cu * NB + cb;
                            s[r] = ((rp[0] + rp[32]) + (rp[64] + rp[96])) + ((rp[128] + rp[160]) + (rp[192] + rp[224])) + gp[r];
                        }
                        const float gi = sigmoidf_(s[0]), gf = sigmoidf_(s[1]), gg = tanhf_(s[2]), go = sigmoidf_(s[3]);
                        float cprev = cst[tid];
                        if (t > 0) cprev *= fdrop[tid];
                        const float c = gf * cprev + gi * gg;
                        cst[tid] = c;
                        a.CA[((size_t)(t + 1) * B + cb) * T2V_H + U] = c;
                        if (a.GA) {
                            float* gs = a.GA + ((size_t)t * B + cb) * T2V_G + U;
                            gs[0] = gi; gs[T2V_H] = gf; gs[2 * T2V_H] = gg; gs[3 * T2V_H] = go;
                        }
                        hd = go * tanhf_(c) * fdrop[32 + tid];
                        a.XS[((size_t)(t + 1) * B + cb) * T2V_XW + U] = hd;
                    }
                    // publish FIRST (the write-through store is what attention(t) waits for): lane (u, plane) sends the 4
                    // items of its plane as one 16-byte store; the saved activations follow
                    const int pu = lane / NP, pp = lane - pu * NP;
                    f32x4 v4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int bb = 4 * pp + i;
                        const float x = __shfl(hd, min(pu, PT_MAXU - 1) * NB + min(bb, NB - 1), 64);
                        v4[i] = bb < B ? x : 0.f;
                    }
                    if (lane < PT_MAXU * NP && pu < nu) pt_st16(rG, grow + (unsigned)(pp * T2V_XW + u0 + pu) * 16u, v4);
                    PT_WALL(wg == NT && t == T / 2, 20);
                }
            }
            PT_STAMP(wg == NT && t == T / 2, 1);
            // ---- decoder_rnn(t-1): same row (its h_dec(t-2) columns included)
            if (do_dec) {
                PTAcc<NB> accD;
                accD.clear();
                pt_fma_range<PT_JD / 2, 0, PT_JD, NB>(wd, X, kp, accD);
                __syncthreads();               // attention_rnn's cell threads are done with red
                pt_reduce_store<NB>(accD, red);
                __syncthreads();
                if (wave == 0) {
                    float hd = 0.f;
                    if (cell_on) {
                        float s[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float* rp = red + (r * 8) * 32 + cu * NB + cb;
                            s[r] = ((rp[0] + rp[32]) + (rp[64] + rp[96])) + ((rp[128] + rp[160]) + (rp[192] + rp[224])) + cst[64 + 32 * r + tid];
                        }
                        const float gi = sigmoidf_(s[0]), gf = sigmoidf_(s[1]), gg = tanhf_(s[2]), go = sigmoidf_(s[3]);
                        const uint32_t idx = (uint32_t)cb * T2V_H + U;
                        const int tt = t - 1;
                        float cprev = cst[32 + tid];
                        if (tt > 0) cprev *= t2v_drop_scale(seed, T2V_RNG_DEC_C, tt - 1, idx, a.p_dec);
                        const float c = gf * cprev + gi * gg;
                        cst[32 + tid] = c;
                        a.CD[((size_t)t * B + cb) * T2V_H + U] = c;
                        if (a.GD) {
                            float* gs = a.GD + ((size_t)tt * B + cb) * T2V_G + U;
                            gs[0] = gi; gs[T2V_H] = gf; gs[2 * T2V_H] = gg; gs[3 * T2V_H] = go;
                        }
                        hd = go * tanhf_(c) * t2v_drop_scale(seed, T2V_RNG_DEC_H, tt, idx, a.p_dec);
                        a.XS[((size_t)(t + 1) * B + cb) * T2V_XW + T2V_KATT + U] = hd;
                    }
                    const int pu = lane / NP, pp = lane - pu * NP;
                    f32x4 v4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int bb = 4 * pp + i;
                        const float x = __shfl(hd, min(pu, PT_MAXU - 1) * NB + min(bb, NB - 1), 64);
                        v4[i] = bb < B ? x : 0.f;
                    }
                    if (t < T && lane < PT_MAXU * NP && pu < nu) pt_st16(rG, grow + (unsigned)(pp * T2V_XW + T2V_KATT + u0 + pu) * 16u, v4);
                }
            } else if (wave == 0) {
                // t = 0: h_dec(-1) = 0 (XS row 1 was cleared by the reset launch)
                const int pu = lane / NP, pp = lane - pu * NP;
                if (lane < PT_MAXU * NP && pu < nu)
                    pt_st16(rG, grow + (unsigned)(pp * T2V_XW + T2V_KATT + u0 + pu) * 16u, f32x4{0.f, 0.f, 0.f, 0.f});
            }
            if (t == T) break;
            PT_STAMP(wg == NT && t == T / 2, 2);
            __syncthreads();                   // every wave is done reading X (row t)
            // ---- row t+1 into X: h_att(t) is there (or about to be), h_dec(t-1) follows, ctx(t) comes last.  Each bulk
            // gather is preceded by a sparse wait on one word per producer (first unit of every L workgroup / last context
            // column of every T workgroup); the bulk loads then find their data on the first try almost always
            {
                // (both segments in ONE pass — 8 loads of 16 bytes in flight per thread — measured slower: the 32 extra
                // registers spill next to the 160 weight registers and the cell phases pay for it: 9.2 vs 8.7 us per step)
                pt_gather<(NP * T2V_H + PT_THREADS - 1) / PT_THREADS, NP>(X, rG, grow, 0, T2V_H, B, 0, a.err, flag);
                PT_STAMP(wg == NT && t == T / 2, 3);
                pt_gather<(NP * T2V_H + PT_THREADS - 1) / PT_THREADS, NP>(X, rG, grow, T2V_KATT, T2V_H, B, 0, a.err, flag);
                PT_STAMP(wg == NT && t == T / 2, 4);
                __syncthreads();
                if (flag[0] != 1) return;
                // the h_att(t) half of attention_rnn(t+1), in the shadow of attention(t)
                accA.clear();
                if (t + 1 < T) pt_fma_range<PT_JA / 2, 0, 8, NB>(wa, X, kp, accA);
                if (cell_on && t + 1 < T) {
                    const uint32_t idx = (uint32_t)cb * T2V_H + U;
                    fdrop[tid] = t2v_drop_scale(seed, T2V_RNG_ATT_C, t, idx, a.p_att);
                    fdrop[32 + tid] = t2v_drop_scale(seed, T2V_RNG_ATT_H, t + 1, idx, a.p_att);
                }
                PT_STAMP(wg == NT && t == T / 2, 6);
                const int rounds = pt_gather<(NP * T2V_E + PT_THREADS - 1) / PT_THREADS, NP>(X, rG, grow, T2V_H, T2V_E, B, ctx_nap, a.err, flag);
                // adaptive nap: wake up just before the context lands (a poll round is about 16 nap units long)
                ctx_nap = t2v_adapt_nap(ctx_nap, rounds);
                PT_WALL(wg == NT && t == T / 2, 23);
                if (a.prof && wg == NT && t == T / 2 && tid == 0) { a.prof[24] = (unsigned long long)rounds; a.prof[25] = (unsigned long long)ctx_nap; }
            }
            __syncthreads();
            if (flag[0] != 1) return;
            PT_STAMP(wg == NT && t == T / 2, 5);
        }
        return;
    }

    // =============================================================================== T role: attention slice (b, s)
    const int ab = wg >> 3, as = wg & 7;
    constexpr int NTI = LONG ? PT_NTI_LONG : 2;          // position tiles per wave: tile jt = wave + 8 i
    constexpr int NPP = LONG ? 2 : 1;                    // positions per thread in the softmax: tid + 512 u
    const int TW = Tcap + 32;
    float* wq_s = lds;                                   // [16][1028]   (LONG: in registers)
    float* mem_s = wq_s + (LONG ? 0 : 16 * 1028);        // [Tcap][64]
    float* pm_s = mem_s + Tcap * 64;                     // [Tcap][16]   (LONG: in registers)
    float* win = pm_s + (LONG ? 0 : Tcap * 16);          // [2][TW]: alignment window, index x <-> position x - 15
    float* eall = win + 2 * TW;                          // [Tcap] (LONG: + T2V_CTX_PAD, zero from Tp on: t2v_ctx_partial)
    float* hx = eall + Tcap + (LONG ? T2V_CTX_PAD : 0);  // [1024] h_att(t) of this item
    float* qv = hx + T2V_H;                              // [16]
    float* qred = qv + 16;                               // [32][16]
    float* cred = qred + 32 * 16;                        // [8][64]
    float* rsm = cred + 8 * 64;                          // [32]
    float* rss = rsm + 32;                               // [32]
    int* flag = (int*)(rss + 32);
    const int g = lane >> 4, c16 = lane & 15;
    float4 wqr[LONG ? 8 : 1], pmr[LONG ? NTI : 1];
    if constexpr (LONG) {
        // the operands of a thread's own part of the query product (dim tid >> 5, k = 4 (tid & 31) + 128 i) and of its tiles'
        // energies (position 16 jt + c16, dims 16 as + 4 g ..) never change over the pass
        const float* wrow = a.wq + (size_t)(16 * as + (tid >> 5)) * 1024 + 4 * (tid & 31);
#pragma unroll
        for (int i = 0; i < 8; ++i) wqr[i] = *(const float4*)(wrow + 128 * i);
#pragma unroll
        for (int i = 0; i < NTI; ++i) {
            const int jp = 16 * (wave + 8 * i) + c16;
            pmr[i] = *(const float4*)(a.pm + ((size_t)ab * Tp + min(jp, Tp - 1)) * T2V_A + 16 * as + 4 * g);
        }
    } else {
        for (int i = tid; i < 16 * 1024; i += PT_THREADS) wq_s[(i >> 10) * 1028 + (i & 1023)] = a.wq[(size_t)(16 * as) * 1024 + i];
    }
    for (int i = tid; i < Tp * 64; i += PT_THREADS) mem_s[i] = a.memory[((size_t)ab * Tp + (i >> 6)) * T2V_E + 64 * as + (i & 63)];
    if constexpr (!LONG)
        for (int i = tid; i < Tp * 16; i += PT_THREADS) pm_s[i] = a.pm[((size_t)ab * Tp + (i >> 4)) * T2V_A + 16 * as + (i & 15)];
    for (int i = tid; i < 2 * TW; i += PT_THREADS) win[i] = 0.f;
    if constexpr (LONG)
        for (int i = Tp + tid; i < Tcap + T2V_CTX_PAD; i += PT_THREADS) eall[i] = 0.f;
    if (tid == 0) flag[0] = 1;
    float areg[16];
    {
        const float4* wp = (const float4*)(a.wcomb + (16 * as + c16) * 64 + 16 * g);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 w4 = wp[u];
            areg[4 * u] = w4.x; areg[4 * u + 1] = w4.y; areg[4 * u + 2] = w4.z; areg[4 * u + 3] = w4.w;
        }
    }
    const float4 vr = *(const float4*)(a.v + 16 * as + 4 * g);
    const int len = a.lengths ? a.lengths[ab] : Tp;
    const int mypl = ab >> 2, myw = ab & 3;
    __syncthreads();
    int h_nap = 0;

    for (int t = 0; t < T; ++t) {
        const unsigned grow = (unsigned)(t + 1) * grow_b;
        PT_STAMP(wg == 0 && t == T / 2, 8);
        // ---- location features of this step's tiles (fused filter, K = 64): they depend on alpha(t-1) only, so they
        // are evaluated BEFORE h_att(t) arrives (wave -> tiles wave, wave + 8)
        f32x4 lacc[NTI];
#pragma unroll
        for (int i = 0; i < NTI; ++i) {
            const int jt = wave + 8 * i;
            lacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
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
                    l0 = mfma16x4(areg[st], bop[st], l0);
                    l1 = mfma16x4(areg[st + 1], bop[st + 1], l1);
                }
                lacc[i] = l0 + l1;
            }
        }
        // ---- h_att(t) of this item: word myw of the 1024 chunks of plane mypl (two per thread); nap, then poll the payload
        {
            const unsigned s0 = grow + (unsigned)(mypl * T2V_XW + tid) * 16u + 4u * (unsigned)myw;
            float v0, v1;
            for (int i = 0; i < h_nap; i += 8) __builtin_amdgcn_s_sleep(8);
            int rounds = 0;
            for (;;) {
                v0 = __uint_as_float(pt_ld4(rG, s0));
                v1 = __uint_as_float(pt_ld4(rG, s0 + PT_THREADS * 16u));
                if (__all(__float_as_uint(v0) != PT_SENT && __float_as_uint(v1) != PT_SENT)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++rounds > (int)(PT_SPIN / 4) || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    flag[0] = 0;
                    break;
                }
            }
            h_nap = t2v_adapt_nap(h_nap, rounds);
            PT_WALL(wg == 0 && t == T / 2, 21);
            if (a.prof && wg == 0 && t == T / 2 && tid == 0) { a.prof[26] = (unsigned long long)rounds; a.prof[27] = (unsigned long long)h_nap; }
            hx[tid] = v0;
            hx[tid + PT_THREADS] = v1;
        }
        __syncthreads();
        if (flag[0] != 1) return;
        PT_STAMP(wg == 0 && t == T / 2, 9);
        // ---- query slice: thread = (dim d = tid >> 5, k part kq = tid & 31): k = 4 kq + 128 i, 16-byte LDS operands;
        // 32-lane sum = 16-lane DPP row sum + one cross-row exchange
        {
            const int d = tid >> 5, kq = tid & 31;
            const float* wrow = wq_s + d * 1028 + 4 * kq;
            const float* hp = hx + 4 * kq;
            float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 w4 = LONG ? wqr[LONG ? i : 0] : *(const float4*)(wrow + 128 * i);
                const float4 h4 = *(const float4*)(hp + 128 * i);
                acc0 = fmaf(w4.x, h4.x, acc0); acc1 = fmaf(w4.y, h4.y, acc1);
                acc0 = fmaf(w4.z, h4.z, acc0); acc1 = fmaf(w4.w, h4.w, acc1);
            }
            float q = row16_sum(acc0 + acc1);
            q += __shfl_xor(q, 16, 64);
            if (kq == 0) qv[d] = q;
        }
        __syncthreads();
        const float4 q4 = *(const float4*)(qv + 4 * g);
        PT_STAMP(wg == 0 && t == T / 2, 13);
        // ---- partial energies of this slice
        const unsigned exw = (unsigned)(((t * B + ab) * 8 + as) * Tcap) * 4u;
#pragma unroll
        for (int i = 0; i < NTI; ++i) {
            const int jt = wave + 8 * i;
            if (16 * jt < Tp) {
                const f32x4 acc = lacc[i];
                const int jp = 16 * jt + c16;
                const float4 pm4 = LONG ? pmr[LONG ? i : 0] : *(const float4*)(pm_s + min(jp, Tp - 1) * 16 + 4 * g);
                float4 sv;
                sv.x = tanhf_(q4.x + acc[0] + pm4.x); sv.y = tanhf_(q4.y + acc[1] + pm4.y);
                sv.z = tanhf_(q4.z + acc[2] + pm4.z); sv.w = tanhf_(q4.w + acc[3] + pm4.w);
                float esum = vr.x * sv.x + vr.y * sv.y + vr.z * sv.z + vr.w * sv.w;
                esum += __shfl_xor(esum, 16, 64);
                esum += __shfl_xor(esum, 32, 64);
                if (g == 0 && jp < Tp) pt_st4(rE, exw + 4u * (unsigned)jp, esum);
                if (a.S && jp < Tp) *(float4*)(a.S + (((size_t)t * B + ab) * Tp + jp) * T2V_A + 16 * as + 4 * g) = sv;
            }
        }
        PT_STAMP(wg == 0 && t == T / 2, 10);
        // (two copies of the softmax: the one-position form is kept word for word so that the short kernels keep their instruction
        // stream — round 6 checked the ISA of <.., false> against the previous build, identical)
        if constexpr (!LONG) {
            // ---- the 8 partials of every position (fixed order), masked softmax
            float ev0 = -INFINITY;
            if (tid < Tp) {
                const unsigned e0 = (unsigned)((t * B + ab) * 8 * Tcap + tid) * 4u;
                unsigned p[8];
                unsigned spins = 0;
                for (;;) {
                    bool ok = true;
    #pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        p[i] = pt_ld4(rE, e0 + (unsigned)(i * Tcap) * 4u);
                        ok = ok && p[i] != PT_SENT;
                    }
                    if (ok) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > PT_SPIN || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        flag[0] = 0;
                        break;
                    }
                }
                const float ev = ((__uint_as_float(p[0]) + __uint_as_float(p[1])) + (__uint_as_float(p[2]) + __uint_as_float(p[3]))) +
                                 ((__uint_as_float(p[4]) + __uint_as_float(p[5])) + (__uint_as_float(p[6]) + __uint_as_float(p[7])));
                ev0 = tid < len ? ev : -INFINITY;
            }
            PT_STAMP(wg == 0 && t == T / 2, 14);
            {
                float mloc = ev0;
                mloc = T2V_DPP_MAX(mloc, 0xB1); mloc = T2V_DPP_MAX(mloc, 0x4E);
                mloc = T2V_DPP_MAX(mloc, 0x141); mloc = T2V_DPP_MAX(mloc, 0x140);
                mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
                mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
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
                float sloc = row16_sum(e0v);
                sloc += __shfl_xor(sloc, 16, 64);
                sloc += __shfl_xor(sloc, 32, 64);
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
                win[15 + tid] = al;                                        // previous weights of the next step
                const float cum = win[TW + 15 + tid] + al;                 // cumulative weights
                win[TW + 15 + tid] = cum;
                if (as == 0) {
                    a.AL[((size_t)(t + 1) * B + ab) * Tp + tid] = al;
                    a.ACUM[((size_t)(t + 1) * B + ab) * Tp + tid] = cum;
                }
            }
        } else {
            // ---- the 8 partials of every position (fixed order), masked softmax; thread -> positions tid + 512 u
            float ev0[NPP];
    #pragma unroll
            for (int u = 0; u < NPP; ++u) ev0[u] = -INFINITY;
            if (tid < Tp) {
                const unsigned e0 = (unsigned)((t * B + ab) * 8 * Tcap + tid) * 4u;
                unsigned p[NPP][8];
                unsigned spins = 0;
                for (;;) {
                    bool ok = true;
    #pragma unroll
                    for (int u = 0; u < NPP; ++u) {
                        // (a second position past the end re-reads the first one's words: no branch around the loads)
                        const unsigned eu = e0 + ((u > 0 && tid + PT_THREADS * u < Tp) ? (unsigned)(PT_THREADS * u) * 4u : 0u);
    #pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            p[u][i] = pt_ld4(rE, eu + (unsigned)(i * Tcap) * 4u);
                            ok = ok && p[u][i] != PT_SENT;
                        }
                    }
                    if (ok) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > PT_SPIN || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                        __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        flag[0] = 0;
                        break;
                    }
                }
    #pragma unroll
                for (int u = 0; u < NPP; ++u) {
                    const float ev = ((__uint_as_float(p[u][0]) + __uint_as_float(p[u][1])) + (__uint_as_float(p[u][2]) + __uint_as_float(p[u][3]))) +
                                     ((__uint_as_float(p[u][4]) + __uint_as_float(p[u][5])) + (__uint_as_float(p[u][6]) + __uint_as_float(p[u][7])));
                    ev0[u] = tid + PT_THREADS * u < len ? ev : -INFINITY;
                }
            }
            PT_STAMP(wg == 0 && t == T / 2, 14);
            {
                float mloc = ev0[0];
    #pragma unroll
                for (int u = 1; u < NPP; ++u) mloc = fmaxf(mloc, ev0[u]);
                mloc = T2V_DPP_MAX(mloc, 0xB1); mloc = T2V_DPP_MAX(mloc, 0x4E);
                mloc = T2V_DPP_MAX(mloc, 0x141); mloc = T2V_DPP_MAX(mloc, 0x140);
                mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
                mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
                if (lane == 0) rsm[wave] = mloc;
            }
            __syncthreads();
            if (flag[0] != 1) return;
            float m;
            {
                const float4 a0 = *(const float4*)rsm, a1 = *(const float4*)(rsm + 4);
                m = fmaxf(fmaxf(fmaxf(a0.x, a0.y), fmaxf(a0.z, a0.w)), fmaxf(fmaxf(a1.x, a1.y), fmaxf(a1.z, a1.w)));
            }
            float e0v[NPP];
    #pragma unroll
            for (int u = 0; u < NPP; ++u) e0v[u] = tid + PT_THREADS * u < Tp ? expf(ev0[u] - m) : 0.f;
            {
                float sloc = e0v[0];
    #pragma unroll
                for (int u = 1; u < NPP; ++u) sloc += e0v[u];
                sloc = row16_sum(sloc);
                sloc += __shfl_xor(sloc, 16, 64);
                sloc += __shfl_xor(sloc, 32, 64);
                if (lane == 0) rss[wave] = sloc;
            }
            __syncthreads();
            float ssum;
            {
                const float4 a0 = *(const float4*)rss, a1 = *(const float4*)(rss + 4);
                ssum = ((a0.x + a0.y) + (a0.z + a0.w)) + ((a1.x + a1.y) + (a1.z + a1.w));
            }
            const float rinv = 1.0f / ssum;
    #pragma unroll
            for (int u = 0; u < NPP; ++u) {
                const int pos = tid + PT_THREADS * u;
                const float al = e0v[u] * rinv;
                if (pos < Tp) {
                    eall[pos] = al;
                    win[15 + pos] = al;                                        // previous weights of the next step
                    const float cum = win[TW + 15 + pos] + al;                 // cumulative weights
                    win[TW + 15 + pos] = cum;
                    if (as == 0) {
                        a.AL[((size_t)(t + 1) * B + ab) * Tp + pos] = al;
                        a.ACUM[((size_t)(t + 1) * B + ab) * Tp + pos] = cum;
                    }
                }
            }
        }
        __syncthreads();
        PT_STAMP(wg == 0 && t == T / 2, 11);
        // ---- context columns [64 as, 64 as + 64): thread = (column c = tid & 63, part = tid >> 6)
        {
            const int c = tid & 63, part = tid >> 6;
            if constexpr (LONG) {
                // (up to 70 positions per thread: eight per round, reads first — 8 000 -> 6 500 cycles at 555 symbols; at <= 224 symbols
                // the plain loop is as fast and the short kernels keep their instruction stream)
                cred[part * 64 + c] = t2v_ctx_partial<8>(eall, mem_s, part, c, Tp);
            } else {
                float acc = 0.f;
                for (int jj = part; jj < Tp; jj += 8) acc = fmaf(eall[jj], mem_s[jj * 64 + c], acc);
                cred[part * 64 + c] = acc;
            }
        }
        __syncthreads();
        if (tid < 64) {
            float acc = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += cred[u * 64 + tid];
            pt_st4(rG, grow + (unsigned)(mypl * T2V_XW + T2V_H + 64 * as + tid) * 16u + 4u * (unsigned)myw, acc);
            a.XS[((size_t)(t + 1) * B + ab) * T2V_XW + T2V_H + 64 * as + tid] = acc;       // (after the publish)
        }
        PT_WALL(wg == 0 && t == T / 2, 22);
        PT_STAMP(wg == 0 && t == T / 2, 12);
    }
}

// sentinel fill of the exchange buffers (16 bytes per thread and iteration)
__global__ __launch_bounds__(256) void k_pt_fill(uint4* p, size_t n16) {
    const uint4 s = {PT_SENT, PT_SENT, PT_SENT, PT_SENT};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = s;
}

// the long form of the attention role: beyond PT_MAXT symbols (T2V_PT_LONG=1: for every length — a measurement switch)
static bool pt_long(int T_in) {
    static const int forced = [] { const char* e = getenv("T2V_PT_LONG"); return e && e[0] == '1' ? 1 : 0; }();
    return T_in > PT_MAXT || forced;
}
static size_t pt_lds_bytes(int B, int T_in) {
    const size_t Tcap = (size_t)((T_in + 15) / 16) * 16;
    const size_t np = B > 4 ? 2 : 1;
    const size_t lrole = np * T2V_XW * 4 + 4 * 8 * 32 + 8 * 32 + 4;
    const size_t resident = pt_long(T_in) ? Tcap * 64 : 16 * 1028 + Tcap * 64 + Tcap * 16;      // LONG: W_q / processed memory in registers
    const size_t trole = resident + 2 * (Tcap + 32) + Tcap + (pt_long(T_in) ? T2V_CTX_PAD : 0) + T2V_H + 16 + 32 * 16 + 8 * 64 + 64 + 4;
    return sizeof(float) * (lrole > trole ? lrole : trole);
}
#define PT_LDS_MAX (160 * 1024)
static size_t pt_g_floats(int B, int T_out) { return (size_t)(T_out + 2) * (B > 4 ? 2 : 1) * T2V_XW * 4; }
static size_t pt_ex_floats(int B, int T_in, int T_out) { return (size_t)T_out * B * 8 * t2v_tcap(T_in); }

static const void* pt_kernel(int B, int T_in) {
    if (pt_long(T_in)) return B > 4 ? (const void*)k_dec_train_persist<6, true> : (const void*)k_dec_train_persist<4, true>;
    return B > 4 ? (const void*)k_dec_train_persist<6, false> : (const void*)k_dec_train_persist<4, false>;
}
static int pt_device_ok(int B, int T_in, size_t lds) {
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
        for (int b = 4; b <= 6; b += 2)
            for (int tin = PT_MAXT; tin <= PT_MAXT + 1; ++tin)
                if (hipFuncSetAttribute(pt_kernel(b, tin), hipFuncAttributeMaxDynamicSharedMemorySize, PT_LDS_MAX) != hipSuccess) {
                    (void)hipGetLastError();
                    return 0;
                }
        attr_set = true;
    }
    int nblk = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nblk, pt_kernel(B, T_in), PT_THREADS, lds) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return nblk >= 1;
}

extern "C" int t2v_decoder_train_persist_supported(int B, int T_in) {
    if (!(B >= 1 && B <= PT_MAXB && T_in >= 1 && T_in <= PT_MAXT_LONG && pt_lds_bytes(B, T_in) <= PT_LDS_MAX)) return 0;
    return pt_device_ok(B, T_in, pt_lds_bytes(B, T_in));
}
extern "C" long t2v_decoder_train_persist_scratch_floats(int B, int T_in, int T_out) {
    if (B < 1 || B > PT_MAXB || T_in < 1 || T_out < 1) return 0;
    return (long)(pt_g_floats(B, T_out) + pt_ex_floats(B, T_in, T_out));
}

extern "C" int t2v_decoder_train_fwd_persistent(const t2v_dec_train_persist_weights* w, const t2v_dec_train_bufs* s, float* scratch,
                                                int B, int T_in, int T_out, float p_att, float p_dec, uint64_t seed, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!w || !s || !scratch || T_out < 1 || !t2v_decoder_train_persist_supported(B, T_in)) return T2V_ERR_ARG;
    if (!w->w_ih_att || !w->w_hh_att || !w->w_ih_dec || !w->w_hh_dec || !w->bias_dec || !w->wq || !w->wcomb || !w->v || !s->gpre ||
        !s->memory || !s->pm || !s->XS || !s->CA || !s->CD || !s->QP || !s->AL || !s->ACUM)
        return T2V_ERR_ARG;
    if ((uintptr_t)scratch & 15) return T2V_ERR_ARG;
    if (pt_g_floats(B, T_out) * 4 >= 0x7fffffffull || pt_ex_floats(B, T_in, T_out) * 4 >= 0x7fffffffull) return T2V_ERR_ARG;   // 31-bit buffer offsets
    // per-pass resets: the sync / error words, the zero initial states of the arena (as t2v_decoder_train_fwd)
    unsigned* sync = (unsigned*)(s->QP + t2v_qp_sync_off(B));
    T2VZeroRegions z;
    z.add(sync, 64 * sizeof(uint32_t));
    z.add(s->XS, sizeof(float) * 2 * B * T2V_XW);
    z.add(s->CA, sizeof(float) * B * T2V_H);
    z.add(s->CD, sizeof(float) * B * T2V_H);
    z.add(s->AL, sizeof(float) * B * T_in);
    z.add(s->ACUM, sizeof(float) * B * T_in);
    t2v_zero_regions(z, stream);
    const size_t nfl = pt_g_floats(B, T_out) + pt_ex_floats(B, T_in, T_out);
    k_pt_fill<<<1024, 256, 0, stream>>>((uint4*)scratch, nfl / 4);
    PTArgs a;
    a.w_ih_att = w->w_ih_att; a.w_hh_att = w->w_hh_att; a.w_ih_dec = w->w_ih_dec; a.w_hh_dec = w->w_hh_dec;
    a.bias_dec = w->bias_dec; a.wq = w->wq; a.wcomb = w->wcomb; a.v = w->v;
    a.gpre = s->gpre; a.memory = s->memory; a.pm = s->pm; a.lengths = s->lengths;
    a.XS = s->XS; a.CA = s->CA; a.CD = s->CD; a.GA = s->GA; a.GD = s->GD; a.AL = s->AL; a.ACUM = s->ACUM; a.S = s->S;
    a.G = scratch;
    a.EX = scratch + pt_g_floats(B, T_out);
    a.err = sync + 31;
    a.B = B; a.T_in = T_in; a.T_out = T_out; a.p_att = p_att; a.p_dec = p_dec; a.seed = seed;
    a.step = t2v_step_for(stream);
    a.prof = g_t2v_prof;
    const size_t lds = pt_lds_bytes(B, T_in);
    if (pt_long(T_in)) {
        if (B > 4) k_dec_train_persist<6, true><<<T2V_NWG, PT_THREADS, lds, stream>>>(a);
        else k_dec_train_persist<4, true><<<T2V_NWG, PT_THREADS, lds, stream>>>(a);
    } else {
        if (B > 4) k_dec_train_persist<6, false><<<T2V_NWG, PT_THREADS, lds, stream>>>(a);
        else k_dec_train_persist<4, false><<<T2V_NWG, PT_THREADS, lds, stream>>>(a);
    }
    return t2v_check_launch();
}
