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
__global__ __launch_bounds__(256) void k_bilstm_fwd(BiLstmFwdArgs a) {
    const int dir = blockIdx.x / BL_NW, j = blockIdx.x % BL_NW;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = lane & 15, g = lane >> 4;
    const bool bvalid = b < a.B;
    __shared__ float hbuf[16][BL_H + 4];
    __shared__ int ok_flag;
    if (tid == 0) ok_flag = 1;
    // recurrent weights of this wave's two 16-row tiles (4 units x 4 gates each) as MFMA A fragments
    float wreg[BL_NT][64];
    {
        const float* W = a.whh + (size_t)dir * BL_G * BL_H;
#pragma unroll
        for (int tt = 0; tt < BL_NT; ++tt) {
            const int unit = j * BL_UNITS + (BL_NT * wave + tt) * 4 + ((lane & 15) >> 2);
            const int row = (lane & 3) * BL_H + unit;
#pragma unroll
            for (int s = 0; s < 64; ++s) wreg[tt][s] = W[(size_t)row * BL_H + 4 * s + g];
        }
    }
    for (int i = tid; i < 16 * (BL_H + 4); i += 256) (&hbuf[0][0])[i] = 0.f;
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
                gxv[tt][r] = active ? a.gx[(((size_t)dir * a.B + b) * a.T + t) * BL_G + r * BL_H + U] : 0.f;
        }
        f32x4 accv[BL_NT];
#pragma unroll
        for (int tt = 0; tt < BL_NT; ++tt) accv[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* hrow = &hbuf[b][g];
#pragma unroll
        for (int s = 0; s < 64; ++s) {
            const float hv = hrow[4 * s];
#pragma unroll
            for (int tt = 0; tt < BL_NT; ++tt) accv[tt] = mfma16x4(wreg[tt][s], hv, accv[tt]);
        }
        t2v_u64* hx_w = a.hx + ((size_t)(dir * 2 + (step & 1)) * 16) * BL_H;
#pragma unroll
        for (int tt = 0; tt < BL_NT; ++tt) {
            const f32x4 acc = accv[tt];
            const int U = j * BL_UNITS + (BL_NT * wave + tt) * 4 + g;
            if (bvalid) {
                float hnew = hbuf[b][U];                       // frozen once the sequence has ended
                if (active) {
                    const float gi = sigmoidf_(acc[0] + gxv[tt][0]), gf = sigmoidf_(acc[1] + gxv[tt][1]);
                    const float gg = tanhf_(acc[2] + gxv[tt][2]), go = sigmoidf_(acc[3] + gxv[tt][3]);
                    const float c = gf * cst[tt] + gi * gg;
                    cst[tt] = c;
                    hnew = go * tanhf_(c);
                    a.y[((size_t)b * a.T + t) * (2 * BL_H) + dir * BL_H + U] = hnew;
                    if (a.gates) {
                        float* gs = a.gates + (((size_t)dir * a.B + b) * a.T + t) * BL_G + U;
                        gs[0] = gi; gs[BL_H] = gf; gs[2 * BL_H] = gg; gs[3 * BL_H] = go;
                        a.cells[(((size_t)dir * a.B + b) * a.T + t) * BL_H + U] = c;
                    }
                }
                bl_put(hx_w + (size_t)b * BL_H + U, hnew, (unsigned)step + 1u);
            }
        }
        if (step + 1 == a.T) break;
        {   // gather the new hidden state of every item (thread = unit tid, items 0..BQ-1): poll until every tag matches
            float hv[BQ];
            unsigned spins = 0;
            for (;;) {
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
                if (u < a.B) hbuf[u][tid] = hv[u];
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
    float* dg;              // (2, B, T, 1024) out: grad wrt gate pre-activations (pre-zeroed by caller)
    t2v_u64* dgx;           // (2, 2, 16, 1024) granule exchange buffer, zeroed by the launcher
    unsigned* sync;
    int B, T;
};

template <int BQ>
__global__ __launch_bounds__(256) void k_bilstm_bwd(BiLstmBwdArgs a) {
    const int dir = blockIdx.x / BL_NW, j = blockIdx.x % BL_NW;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int b = lane & 15, g = lane >> 4;
    __shared__ float dgbuf[16][BL_G + 4];           // all gate gradients of the current step
    __shared__ f32x4 red[4][BL_NT][64];
    __shared__ float dhrec[16][BL_UNITS + 1];       // dL/dh_{prev} for this workgroup's 32 units
    __shared__ int ok_flag;
    if (tid == 0) ok_flag = 1;
    // W_hh^T rows of this workgroup's 32 units (2 tiles of 16), K = 1024 split over the 4 waves
    float wreg[BL_NT][64];
    {
        const float* W = a.whh + (size_t)dir * BL_G * BL_H;
#pragma unroll
        for (int tt = 0; tt < BL_NT; ++tt) {
            const int col = j * BL_UNITS + tt * 16 + (lane & 15);       // h unit = column of W_hh
#pragma unroll
            for (int s = 0; s < 64; ++s) wreg[tt][s] = W[(size_t)(256 * wave + 4 * s + g) * BL_H + col];
        }
    }
    for (int i = tid; i < 16 * (BL_UNITS + 1); i += 256) (&dhrec[0][0])[i] = 0.f;
    // cell-backward ownership: thread -> (item bb, unit uu) pairs, uu = tid % BL_UNITS, bb = tid / BL_UNITS (+ BL_IPP)
    constexpr int BL_IPP = 256 / BL_UNITS, BL_REPS = 16 / BL_IPP;      // items per pass, passes
    const int uu = tid & (BL_UNITS - 1), U = j * BL_UNITS + uu;
    float dcrec[BL_REPS];
#pragma unroll
    for (int rep = 0; rep < BL_REPS; ++rep) dcrec[rep] = 0.f;
    __syncthreads();

    for (int step = a.T - 1; step >= 0; --step) {
        t2v_u64* dgx_w = a.dgx + ((size_t)(dir * 2 + (step & 1)) * 16) * BL_G;
        const unsigned tag = (unsigned)(a.T - step);
#pragma unroll
        for (int rep = 0; rep < BL_REPS; ++rep) {
            const int bb = tid / BL_UNITS + BL_IPP * rep;
            if (bb < a.B) {
                const int len = a.lengths[bb];
                float di = 0.f, df = 0.f, dgg = 0.f, dob = 0.f;
                if (step < len) {
                    const int t = dir == 0 ? step : len - 1 - step;
                    const size_t idx = ((size_t)dir * a.B + bb) * a.T + t;
                    const float* gs = a.gates + idx * BL_G + U;
                    const float gi = gs[0], gf = gs[BL_H], gg = gs[2 * BL_H], go = gs[3 * BL_H];
                    const float c = a.cells[idx * BL_H + U];
                    float cprev = 0.f;
                    if (step > 0) {
                        const int tp = dir == 0 ? t - 1 : t + 1;
                        cprev = a.cells[(((size_t)dir * a.B + bb) * a.T + tp) * BL_H + U];
                    }
                    const float dh = a.dy[((size_t)bb * a.T + t) * (2 * BL_H) + dir * BL_H + U] + dhrec[bb][uu];
                    const float tc = tanhf_(c);
                    const float dct = dcrec[rep] + dh * go * (1.f - tc * tc);
                    dob = dh * tc * go * (1.f - go);
                    di = dct * gg * gi * (1.f - gi);
                    df = dct * cprev * gf * (1.f - gf);
                    dgg = dct * gi * (1.f - gg * gg);
                    dcrec[rep] = dct * gf;
                    float* o = a.dg + idx * BL_G + U;
                    o[0] = di; o[BL_H] = df; o[2 * BL_H] = dgg; o[3 * BL_H] = dob;
                }
                t2v_u64* x = dgx_w + (size_t)bb * BL_G + U;
                bl_put(x, di, tag); bl_put(x + BL_H, df, tag); bl_put(x + 2 * BL_H, dgg, tag); bl_put(x + 3 * BL_H, dob, tag);
            }
        }
        if (step == 0) break;
        // one wave polls a sentinel granule per (producer workgroup, item) — the last one each producing thread writes —
        // with naps in between; the other waves stay off the memory system until the rows have landed (256 threads x 32
        // polling loads per workgroup slowed every producer down)
        if (wave == 0) {
            const int nsent = BL_NW * a.B;
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
                for (int i = lane; i < nsent; i += 64) {
                    const int pj = i % BL_NW, pb = i / BL_NW;
                    const t2v_u64 x = __hip_atomic_load(dgx_w + (size_t)pb * BL_G + 3 * BL_H + pj * BL_UNITS + (BL_UNITS - 1), __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT);
                    ok = ok && (unsigned)(x >> 32) == tag;
                }
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > BL_SPIN || __hip_atomic_load(a.sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(a.sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok_flag = 0;
                    break;
                }
            }
        }
        __syncthreads();
        if (!ok_flag) return;
        {   // gather all gate gradients of this step (thread = gate rows tid, tid+256, .. of items 0..BQ-1)
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int u = 0; u < BQ; ++u) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const t2v_u64 x = __hip_atomic_load(dgx_w + (size_t)min(u, a.B - 1) * BL_G + 256 * q + tid, __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT);
                        ok = ok && (unsigned)(x >> 32) == tag;
                        if (u < a.B) dgbuf[u][256 * q + tid] = __uint_as_float((unsigned)x);
                    }
                }
                if (ok) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > BL_SPIN || __hip_atomic_load(a.sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(a.sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok_flag = 0;
                    break;
                }
            }
        }
        __syncthreads();
        if (!ok_flag) return;
        // dh_rec[b][unit] = sum_k W_hh[k][unit] * dgates[b][k]
        f32x4 accv[BL_NT];
#pragma unroll
        for (int tt = 0; tt < BL_NT; ++tt) accv[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* drow = &dgbuf[b][256 * wave + g];
#pragma unroll
        for (int s = 0; s < 64; ++s) {
            const float dv = drow[4 * s];
#pragma unroll
            for (int tt = 0; tt < BL_NT; ++tt) accv[tt] = mfma16x4(wreg[tt][s], dv, accv[tt]);
        }
#pragma unroll
        for (int tt = 0; tt < BL_NT; ++tt) red[wave][tt][lane] = accv[tt];
        __syncthreads();
        if (wave < BL_NT) {     // wave tt finalises tile tt: lane (col = item b, rows 4g+r = units 16tt+4g+r)
            const f32x4 s4 = red[0][wave][lane] + red[1][wave][lane] + red[2][wave][lane] + red[3][wave][lane];
            if (b < a.B) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dhrec[b][16 * wave + 4 * g + r] = s4[r];
            }
        }
        __syncthreads();
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
    if (B <= 4) k_bilstm_fwd<4><<<2 * BL_NW, 256, 0, stream>>>(a);
    else if (B <= 8) k_bilstm_fwd<8><<<2 * BL_NW, 256, 0, stream>>>(a);
    else if (B <= 12) k_bilstm_fwd<12><<<2 * BL_NW, 256, 0, stream>>>(a);
    else k_bilstm_fwd<16><<<2 * BL_NW, 256, 0, stream>>>(a);
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
    z.add(dgx_scratch, sizeof(t2v_u64) * 2 * 2 * 16 * BL_G);      // granule tags
    z.add(dg, sizeof(float) * (size_t)2 * B * T * BL_G);          // padded positions carry no gradient
    t2v_zero_regions(z, stream);
    BiLstmBwdArgs a;
    a.whh = whh; a.lengths = lengths; a.dy = dy; a.gates = gates; a.cells = cells; a.dg = dg; a.dgx = (t2v_u64*)dgx_scratch;
    a.sync = sync3; a.B = B; a.T = T;
    if (B <= 4) k_bilstm_bwd<4><<<2 * BL_NW, 256, 0, stream>>>(a);
    else if (B <= 8) k_bilstm_bwd<8><<<2 * BL_NW, 256, 0, stream>>>(a);
    else if (B <= 12) k_bilstm_bwd<12><<<2 * BL_NW, 256, 0, stream>>>(a);
    else k_bilstm_bwd<16><<<2 * BL_NW, 256, 0, stream>>>(a);
    return t2v_check_launch();
}
