// Encoder BiLSTM(512 -> 256 x 2) recurrence (reference model.py:171-173,183-190: nn.LSTM on a packed
// sequence), forward and BPTT, as PERSISTENT cooperative kernels: 16 workgroups per direction, each owning
// 16 hidden units whose recurrent weights (64 gate rows x 256, fp32) stay in VGPRs as
// v_mfma_f32_16x16x4_f32 A-fragments for all T steps (1 MB per direction = 16 x 4 waves x 64 VGPRs; with 8
// workgroups the per-step MFMA chain was twice as long: 576 -> 478 us forward, -160 us forward+backward).
// Per step the workgroups of a direction exchange the new hidden state (forward) / gate gradients
// (backward) through global memory as 8-byte {value, step tag} granules (one sc1 store each, the data is the
// flag: MI355X guide, Guideline 16 form R2): the consumers poll the granules themselves — no arrival counter,
// no drain, no barrier (the counter barrier cost ~3 us of the 5 / 9.5 us a step took).  Every spin is bounded:
// on a timeout the kernel sets an error word and every workgroup leaves (no hang).
// The input projection X·W_ih^T + b (time-batched GEMM) is done outside; packing semantics = each
// sequence runs over its OWN length (reverse direction starts at len_b-1), padded outputs are zero.
#include "t2v_common.h"
#include "t2v_kernels.h"
#include "t2v_coop.h"

#define BL_H 256
#define BL_G (4 * BL_H)
#define BL_NW 16                // workgroups per direction
#define BL_UNITS (BL_H / BL_NW) // 16 units per workgroup
#define BL_NT (BL_UNITS / 16)   // forward: 16-row MFMA tiles (4 units x 4 gates) per wave; backward: 16-column tiles
                                // of W_hh^T per workgroup

struct BiLstmFwdArgs {
    const float* gx;        // (2, B, T, 1024) input projections + both biases, gate-major i,f,g,o
    const float* whh;       // (2, 1024, 256)
    const int32_t* lengths; // (B)
    float* y;               // (B, T, 512) outputs [fwd | rev], zero at padded positions (pre-zeroed by caller)
    float* gates;           // (2, B, T, 1024) saved gate activations (training) or NULL
    float* cells;           // (2, B, T, 256)  saved cell states (training) or NULL
    t2v_u64* hx;            // (2, 2, 16, 256) granule exchange buffer (double buffered by step parity), zeroed by the launcher
    unsigned* sync;         // [2] error word; zeroed by the launcher
    int B, T;
};

#define BL_SPIN 4000000u
__device__ __forceinline__ void bl_put(t2v_u64* p, float v, unsigned tag) {
    __hip_atomic_store(p, ((t2v_u64)tag << 32) | (t2v_u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int BQ>       // batch rounded up to a multiple of 4: items polled per thread
// Round 4: 8 waves — waves 4..7 take the upper half of K (h_att columns 128..255) of the same tiles as waves 0..3, so the
// chain of dependent MFMAs of a step is 32 long instead of 64 (0.85 -> 0.43 us); their partial tiles cross through LDS.
__global__ __launch_bounds__(512) void k_bilstm_fwd(BiLstmFwdArgs a) {
    const int dir = blockIdx.x / BL_NW, j = blockIdx.x % BL_NW;
    const int tid = threadIdx.x, wave = (tid >> 6) & 3, kh = tid >> 8, lane = tid & 63;
    const int b = lane & 15, g = lane >> 4;
    const bool bvalid = b < a.B;
    __shared__ float hbuf[16][BL_H + 4];
    __shared__ f32x4 kpart[4][BL_NT][64];
    __shared__ int ok_flag;
    if (tid == 0) ok_flag = 1;
    // recurrent weights of this wave's two 16-row tiles (4 units x 4 gates each) as MFMA A fragments
    float wreg[BL_NT][32];
    {
        const float* W = a.whh + (size_t)dir * BL_G * BL_H;
#pragma unroll
        for (int tt = 0; tt < BL_NT; ++tt) {
            const int unit = j * BL_UNITS + (BL_NT * wave + tt) * 4 + ((lane & 15) >> 2);
            const int row = (lane & 3) * BL_H + unit;
#pragma unroll
            for (int s = 0; s < 32; ++s) wreg[tt][s] = W[(size_t)row * BL_H + 4 * (32 * kh + s) + g];
        }
    }
    for (int i = tid; i < 16 * (BL_H + 4); i += 512) (&hbuf[0][0])[i] = 0.f;
    const int len = bvalid ? a.lengths[b] : 0;
    float cst[BL_NT];
#pragma unroll
    for (int tt = 0; tt < BL_NT; ++tt) cst[tt] = 0.f;
    __syncthreads();

    for (int step = 0; step < a.T; ++step) {
        const bool active = step < len;
        const int t = dir == 0 ? step : len - 1 - step;        // own-length reverse (packed sequence)
        // input projections for this lane's (unit, item): 4 gates per tile
        float gxv[BL_NT][4];
#pragma unroll
        for (int tt = 0; tt < BL_NT; ++tt) {
            const int U = j * BL_UNITS + (BL_NT * wave + tt) * 4 + g;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                gxv[tt][r] = (active && kh == 0) ? a.gx[(((size_t)dir * a.B + b) * a.T + t) * BL_G + r * BL_H + U] : 0.f;
        }
        f32x4 accv[BL_NT];
#pragma unroll
        for (int tt = 0; tt < BL_NT; ++tt) accv[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* hrow = &hbuf[b][g + 128 * kh];
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const float hv = hrow[4 * s];
#pragma unroll
            for (int tt = 0; tt < BL_NT; ++tt) accv[tt] = mfma16x4(wreg[tt][s], hv, accv[tt]);
        }
        if (kh == 1) {
#pragma unroll
            for (int tt = 0; tt < BL_NT; ++tt) kpart[wave][tt][lane] = accv[tt];
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int tt = 0; tt < BL_NT; ++tt) accv[tt] += kpart[wave][tt][lane];
        }
        t2v_u64* hx_w = a.hx + ((size_t)(dir * 2 + (step & 1)) * 16) * BL_H;
#pragma unroll
        for (int tt = 0; tt < BL_NT; ++tt) {
            const f32x4 acc = accv[tt];
            const int U = j * BL_UNITS + (BL_NT * wave + tt) * 4 + g;
            if (bvalid && kh == 0) {
                float hnew = hbuf[b][U];                       // frozen once the sequence has ended
                float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f, c = 0.f;
                if (active) {
                    gi = sigmoidf_(acc[0] + gxv[tt][0]); gf = sigmoidf_(acc[1] + gxv[tt][1]);
                    gg = tanhf_(acc[2] + gxv[tt][2]); go = sigmoidf_(acc[3] + gxv[tt][3]);
                    c = gf * cst[tt] + gi * gg;
                    cst[tt] = c;
                    hnew = go * tanhf_(c);
                }
                // the hand-off first (round 4: the granule every workgroup of this direction polls must not sit behind six
                // scattered stores in this CU's memory pipe), the saved activations after it
                bl_put(hx_w + (size_t)b * BL_H + U, hnew, (unsigned)step + 1u);
                if (active) {
                    a.y[((size_t)b * a.T + t) * (2 * BL_H) + dir * BL_H + U] = hnew;
                    if (a.gates) {
                        float* gs = a.gates + (((size_t)dir * a.B + b) * a.T + t) * BL_G + U;
                        gs[0] = gi; gs[BL_H] = gf; gs[2 * BL_H] = gg; gs[3 * BL_H] = go;
                        a.cells[(((size_t)dir * a.B + b) * a.T + t) * BL_H + U] = c;
                    }
                }
            }
        }
        if (step + 1 == a.T) break;
        {   // gather the new hidden state of every item (thread = unit tid < 256, items 0..BQ-1): poll until every tag matches
            float hv[BQ];
            unsigned spins = 0;
            for (; tid < BL_H;) {
                bool ok = true;
#pragma unroll
                for (int u = 0; u < BQ; ++u) {
                    const t2v_u64 x = __hip_atomic_load(hx_w + (size_t)min(u, a.B - 1) * BL_H + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    hv[u] = __uint_as_float((unsigned)x);
                    ok = ok && (unsigned)(x >> 32) == (unsigned)step + 1u;
                }
                if (ok) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > BL_SPIN || __hip_atomic_load(a.sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(a.sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok_flag = 0;
                    break;
                }
            }
            __syncthreads();           // every wave is done reading hbuf of this step
#pragma unroll
            for (int u = 0; u < BQ; ++u)
                if (u < a.B && tid < BL_H) hbuf[u][tid] = hv[u];
        }
        __syncthreads();
        if (!ok_flag) return;
    }
}

struct BiLstmBwdArgs {
    const float* whh;       // (2, 1024, 256)
    const int32_t* lengths;
    const float* dy;        // (B, T, 512)
    const float* gates;     // (2, B, T, 1024)
    const float* cells;     // (2, B, T, 256)
    float* dg;              // (2, B, T, 1024) out: grad wrt gate pre-activations (zeroed by the launcher)
    t2v_u64* dgx;           // (2 dirs, 2 parities, 16 producers, 16 items, 256 units) granules, zeroed by the launcher
    unsigned* sync;
    int B, T;
};

// BPTT as a reduce-scatter: workgroup j keeps the SAME 64 gate rows of W_hh as in the forward (its 16 units x 4 gates) and
// turns its own gate gradients of step t into the partial recurrent gradient of ALL 256 units,
//     P_j[b][u] = sum_{r in rows of j} W_hh[r][u] * dgate[b][r]          (K = 64, on the MFMA, operands in registers)
// and publishes the 256 x B partials as {value, step tag} granules; the consumer thread (item b, unit u) sums the 16
// partials of its unit (fixed order) — 16 granule loads — instead of the whole workgroup pulling all 1024 x B gate
// gradients (an all-gather) before it could start its matrix product.  Per workgroup and step: 256 B stores + 256 B
// loads, the loads being the only round trip on the critical path.
#define BLB_GR ((size_t)16 * 16 * BL_H)     // granules per (direction, parity)
// (round 4: 8 waves x 2 column tiles instead of 4 x 4 — the 64 MFMAs of a wave per step become 32; the cell backward stays on
// threads 0..255)
__global__ __launch_bounds__(512) void k_bilstm_bwd(BiLstmBwdArgs a) {
    const int dir = blockIdx.x / BL_NW, j = blockIdx.x % BL_NW;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = lane & 15, g = lane >> 4;
    __shared__ __attribute__((aligned(16))) float dgl[16][64 + 4];      // own gate gradients [item][k = gate*16 + unit]
    __shared__ int ok_flag;
    if (tid == 0) ok_flag = 1;
    // A fragments of W_hh^T: tile tt of this wave = units 32 wave + 16 tt .. +16, k-step s covers own rows k = 4s + g
    float wreg[2][16];
    {
        const float* W = a.whh + (size_t)dir * BL_G * BL_H;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int k = 4 * s + g;
            const int row = (k >> 4) * BL_H + j * BL_UNITS + (k & 15);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) wreg[tt][s] = W[(size_t)row * BL_H + 32 * wave + 16 * tt + (lane & 15)];
        }
    }
    for (int i = tid; i < 16 * 68; i += 512) (&dgl[0][0])[i] = 0.f;
    // cell-backward ownership: thread -> (item bb = tid / 16, unit uu = tid % 16)
    const int uu = tid & 15, bb = tid >> 4, U = j * BL_UNITS + uu;
    const bool live = bb < a.B;
    const int len = live ? a.lengths[bb] : 0;
    float dcrec = 0.f;
    __syncthreads();

    for (int step = a.T - 1; step >= 0; --step) {
        const unsigned tag = (unsigned)(a.T - step);           // tag of the partials THIS iteration publishes
        // operands of this step's cell backward: independent of the recurrence, requested before the wait below
        const bool on = live && step < len;
        const int t = dir == 0 ? step : len - 1 - step;
        const size_t idx = ((size_t)dir * a.B + (live ? bb : 0)) * a.T + (on ? t : 0);
        float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f, c = 0.f, cprev = 0.f, dyv = 0.f;
        if (on) {
            const float* gs = a.gates + idx * BL_G + U;
            gi = gs[0]; gf = gs[BL_H]; gg = gs[2 * BL_H]; go = gs[3 * BL_H];
            c = a.cells[idx * BL_H + U];
            if (step > 0) cprev = a.cells[(((size_t)dir * a.B + bb) * a.T + (dir == 0 ? t - 1 : t + 1)) * BL_H + U];
            dyv = a.dy[((size_t)bb * a.T + t) * (2 * BL_H) + dir * BL_H + U];
        }
        // recurrent gradient of (bb, U): the 16 partials published by iteration step + 1 (tag - 1)
        float dhr = 0.f;
        if (step < a.T - 1 && live) {
            const t2v_u64* src = a.dgx + (size_t)(dir * 2 + ((step + 1) & 1)) * BLB_GR + (size_t)bb * BL_H + U;
            unsigned spins = 0;
            for (;;) {
                t2v_u64 x[16];
#pragma unroll
                for (int p = 0; p < 16; ++p) x[p] = __hip_atomic_load(src + (size_t)p * 16 * BL_H, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bool ok = true;
#pragma unroll
                for (int p = 0; p < 16; ++p) ok = ok && (unsigned)(x[p] >> 32) == tag - 1u;
                if (ok) {
                    float v[16];
#pragma unroll
                    for (int p = 0; p < 16; ++p) v[p] = __uint_as_float((unsigned)x[p]);
#pragma unroll
                    for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
                        for (int p = 0; p < w; ++p) v[p] += v[p + w];
                    dhr = v[0];
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
                if (++spins > BL_SPIN || __hip_atomic_load(a.sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(a.sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok_flag = 0;
                    break;
                }
            }
        }
        float di = 0.f, df = 0.f, dgg = 0.f, dob = 0.f;
        if (on) {
            const float dh = dyv + dhr;
            const float tc = tanhf_(c);
            const float dct = dcrec + dh * go * (1.f - tc * tc);
            dob = dh * tc * go * (1.f - go);
            di = dct * gg * gi * (1.f - gi);
            df = dct * cprev * gf * (1.f - gf);
            dgg = dct * gi * (1.f - gg * gg);
            dcrec = dct * gf;
            if (step == 0) {
                float* o = a.dg + idx * BL_G + U;
                o[0] = di; o[BL_H] = df; o[2 * BL_H] = dgg; o[3 * BL_H] = dob;
            }
        }
        if (live) { dgl[bb][uu] = di; dgl[bb][16 + uu] = df; dgl[bb][32 + uu] = dgg; dgl[bb][48 + uu] = dob; }
        __syncthreads();
        if (!ok_flag) return;
        if (step == 0) break;
        // partial recurrent gradient of all 256 units from the own 64 gate rows
        {
            float bop[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) bop[s] = dgl[b][4 * s + g];
            __builtin_amdgcn_sched_barrier(0);
            f32x4 acc[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) acc[tt] = mfma16x4(wreg[tt][s], bop[s], acc[tt]);
            if (b < a.B) {
                t2v_u64* dst = a.dgx + (size_t)(dir * 2 + (step & 1)) * BLB_GR + ((size_t)j * 16 + b) * BL_H + 32 * wave + 4 * g;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) bl_put(dst + 16 * tt + r, acc[tt][r], tag);
            }
        }
        if (on) {       // the saved gate gradients leave AFTER the hand-off (round 4)
            float* o = a.dg + idx * BL_G + U;
            o[0] = di; o[BL_H] = df; o[2 * BL_H] = dgg; o[3 * BL_H] = dob;
        }
        __syncthreads();        // dgl is rewritten by the next iteration
    }
}

extern "C" int t2v_bilstm_fwd(const float* gx, const float* whh, const int32_t* lengths, float* y, float* gates,
                              float* cells, float* hx_scratch, uint32_t* sync3, int B, int T, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!gx || !whh || !lengths || !y || !hx_scratch || !sync3 || B < 1 || B > 16 || T < 1) return T2V_ERR_ARG;
    if ((gates == nullptr) != (cells == nullptr)) return T2V_ERR_ARG;
    T2VZeroRegions z;
    z.add(sync3, 3 * sizeof(uint32_t));
    z.add(hx_scratch, sizeof(t2v_u64) * 2 * 2 * 16 * BL_H);       // granule tags
    z.add(y, sizeof(float) * (size_t)B * T * 2 * BL_H);           // padded positions stay zero (pad_packed_sequence)
    t2v_zero_regions(z, stream);
    BiLstmFwdArgs a;
    a.gx = gx; a.whh = whh; a.lengths = lengths; a.y = y; a.gates = gates; a.cells = cells; a.hx = (t2v_u64*)hx_scratch;
    a.sync = sync3; a.B = B; a.T = T;
    if (B <= 4) k_bilstm_fwd<4><<<2 * BL_NW, 512, 0, stream>>>(a);
    else if (B <= 8) k_bilstm_fwd<8><<<2 * BL_NW, 512, 0, stream>>>(a);
    else if (B <= 12) k_bilstm_fwd<12><<<2 * BL_NW, 512, 0, stream>>>(a);
    else k_bilstm_fwd<16><<<2 * BL_NW, 512, 0, stream>>>(a);
    return t2v_check_launch();
}

extern "C" int t2v_bilstm_bwd(const float* whh, const int32_t* lengths, const float* dy, const float* gates,
                              const float* cells, float* dg, float* dgx_scratch, uint32_t* sync3, int B, int T,
                              void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!whh || !lengths || !dy || !gates || !cells || !dg || !dgx_scratch || !sync3 || B < 1 || B > 16 || T < 1)
        return T2V_ERR_ARG;
    T2VZeroRegions z;
    z.add(sync3, 3 * sizeof(uint32_t));
    z.add(dgx_scratch, sizeof(t2v_u64) * 2 * 2 * BLB_GR);         // granule tags
    z.add(dg, sizeof(float) * (size_t)2 * B * T * BL_G);          // padded positions carry no gradient
    t2v_zero_regions(z, stream);
    BiLstmBwdArgs a;
    a.whh = whh; a.lengths = lengths; a.dy = dy; a.gates = gates; a.cells = cells; a.dg = dg; a.dgx = (t2v_u64*)dgx_scratch;
    a.sync = sync3; a.B = B; a.T = T;
    k_bilstm_bwd<<<2 * BL_NW, 512, 0, stream>>>(a);
    return t2v_check_launch();
}
