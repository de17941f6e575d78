/* libt2vae_hip — C ABI of the MI355X-native Tacotron2-VAE hot path.
 *
 * The reference (jinhan/tacotron2-vae) has no FFI of its own: its hot path sits behind Python
 * module names.  This header is the boundary our host-side mirror
 * (the Python modules under tacotron2-vae_amd/) binds with ctypes; each entry point cites the reference code it
 * replaces.  All pointers are DEVICE pointers owned by the caller (PyTorch's allocator in our
 * host code); the library allocates nothing, launches asynchronously on `stream`
 * (a hipStream_t passed as void*), and returns 0 or a negative t2v error code.
 * Layouts are row-major fp32 unless stated; LSTM gates are stacked i,f,g,o (PyTorch order).
 */
#ifndef T2VAE_H
#define T2VAE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define T2V_OK 0
#define T2V_ERR_DIMS -1     /* geometry differs from the compiled default hparams */
#define T2V_ERR_ARG -2      /* null pointer / bad size */
#define T2V_ERR_LAUNCH -3   /* hipGetLastError() != hipSuccess after a launch */

const char* t2v_version(void);
/* last HIP error string recorded by a failing call (thread-local) */
const char* t2v_last_error(void);

/* Tracing aid: device buffer of 32 uint64 that the attention kernels fill with s_memtime stamps at
 * their phase boundaries (forward slots 0..7, backward 16..23); NULL (default) disables it. */
void t2v_set_phase_profile(unsigned long long* dev_buf32);
/* Measurement aid: a one-thread launch on `stream` that writes the chip-wide 100 MHz wall clock into dev_buf[slot]
 * (phase time line of an undisturbed graph replay: tools/stamps.py). */
int t2v_stamp(unsigned long long* dev_buf, int slot, void* stream);
/* Test aid: `wgs` workgroups (256 threads each) that occupy CUs for `microseconds` doing nothing — a stand-in for a
 * neighbour (another process, a communication kernel) while the persistent kernels start. */
int t2v_debug_spin(int wgs, int microseconds, void* stream);

/* Per-step parameters in DEVICE memory (optional).  Kernel arguments are frozen when a training step is captured
 * into a HIP graph; this 32-byte record is read by the kernels at run time instead, so every replay gets fresh
 * dropout masks (epoch is mixed into every dropout seed: seed + epoch * 0x9E3779B97F4A7C15), the right Adam bias
 * corrections / learning rate and KL weight.  NULL (default): the by-value arguments of each call are used. */
typedef struct t2v_step_params {
    uint64_t epoch;      /* training iteration */
    float lr;            /* Adam step size (train.py:209-210 sets it every iteration) */
    float bc1;           /* 1 - beta1^t */
    float bc2s;          /* sqrt(1 - beta2^t) */
    float kl_weight;     /* loss_function.py:15-24 */
    float pad[2];
} t2v_step_params;
void t2v_set_step_params(const t2v_step_params* dev);
/* Per-stream binding (one training engine = one rank = one stream, SURVEY.md 8(b) "thread-safe per ctx"): launches on
 * `stream` read THIS record; launches on a stream without a binding read the process default above.  dev = NULL removes
 * the binding.  Two engines in one process therefore never share a step record as long as each runs on its own stream. */
int t2v_set_step_params_stream(void* stream, const t2v_step_params* dev);

/* ------------------------------------------------------------------ weight packing
 * Re-lays the two decoder LSTM cells' weights into MFMA-fragment order for the per-step
 * weight-streaming kernels.  Replaces nothing in the reference (cuDNN/cuBLAS choose their own
 * layouts); inputs are the nn.LSTMCell tensors themselves (model.py:224-235), read in place:
 *   w_ih_att (4096,768) [prenet | ctx], w_hh_att (4096,1024), w_ih_dec (4096,1536) [h_att | ctx], w_hh_dec (4096,1024)
 *   logical Wcat_att (4096,k_att) = [w_hh_att | w_ih_att[:,256:768] (| w_ih_att[:,0:256])]
 *              k_att = 1536 (training; prenet term hoisted) or 1792 (inference)
 *   logical Wcat_dec (4096,2560)  = [w_ih_dec | w_hh_dec]
 *   packF_*  : forward tiles,  4096*k_att and 4096*2560 floats
 *   packB_*  : transposed tiles for the backward data-gradient GEMV, 4096*1536 and 4096*2560 floats (may be NULL)
 *   packF_att and packF_dec may BOTH be NULL when only the transposed tiles are wanted (forward on the persistent kernel) */
int t2v_pack_lstm_weights(const float* w_ih_att, const float* w_hh_att, const float* w_ih_dec,
                          const float* w_hh_dec, int k_att, float* packF_att, float* packF_dec,
                          float* packB_att, float* packB_dec, void* stream);
/* bf16_run: the same tiles with every weight rounded to bf16 (RNE), laid out in 32-column blocks (16 bytes = 8 values per
 * lane and block: lane l of a 16-row tile holds columns 32 kb + 8 (l >> 4) .. + 8): the two per-step LSTM kernels then
 * stream 33.5 MB instead of 67 MB with HALF the weight-load instructions and multiply bf16-rounded states on
 * v_mfma_f32_16x16x32_bf16 (fp32 accumulation, fp32 cell state, fp32 saved activations).  Same buffer sizes or half. */
int t2v_pack_lstm_weights_bf16(const float* w_ih_att, const float* w_hh_att, const float* w_ih_dec,
                               const float* w_hh_dec, int k_att, void* packF_att, void* packF_dec,
                               void* packB_att, void* packB_dec, void* stream);

/* LocationLayer (model.py:12-28) is a bias-free conv followed by a bias-free linear layer, i.e. ONE linear map of the
 * 2 x 31 alignment window: W_comb[d][32c + k] = sum_f loc_dense[d][f] * loc_conv[f][c][k] (128 x 64; k < 31, columns
 * 31 and 63 are zero).  The attention kernels evaluate the layer (and its backward) through this fused filter bank.
 * wcomb: 2*128*64 floats, W_comb stored twice in the order the consuming lanes read it as float4 runs:
 *   [0, 8192)      F[d][g][st]  = W_comb[d][4st + g]     (g < 4, st < 16)   forward
 *   [8192, 16384)  R[kk][g][st] = W_comb[4st + g][kk]    (g < 4, st < 32)   backward
 * Call once per pass after the weights changed. */
int t2v_fuse_location_weights(const float* loc_conv, const float* loc_dense, float* wcomb, void* stream);

typedef struct t2v_dec_weights {
    const float* packF_att;   /* from t2v_pack_lstm_weights */
    const float* packF_dec;
    const float* packB_att;   /* NULL for inference */
    const float* packB_dec;
    const float* bias_att;    /* (4096) bias_ih+bias_hh; used only when gpre == NULL */
    const float* bias_dec;    /* (4096) bias_ih+bias_hh */
    const float* wqT;         /* (1024,128) query_layer weight, transposed (model.py:35) */
    const float* wcomb;       /* (2,128,64) fused location filter bank (t2v_fuse_location_weights; model.py:17-22) */
    const float* v;           /* (128)      attention v (model.py:39) */
    int32_t packs_bf16;       /* != 0: the four packs came from t2v_pack_lstm_weights_bf16 (hparams bf16_run) */
} t2v_dec_weights;

/* Saved-activation arena of one teacher-forced decoder pass (caller allocates; t2v_decoder_train_fwd clears the rows
 * marked "row 0 = 0" — the zero initial states of model.py:280-296 — together with its sync words in one launch). */
typedef struct t2v_dec_train_bufs {
    const float* gpre;      /* (T,B,4096)  prenet(x_t)·W_ih[:, :256]^T + b_ih + b_hh */
    const float* memory;    /* (B,T_in,512) encoder outputs + style */
    const float* pm;        /* (B,T_in,128) memory_layer(memory) (model.py:290) */
    const int32_t* lengths; /* (B) valid encoder positions; NULL = no mask */
    float* XS;    /* (T+2,B,2560) XS[t+1] = [h_att_t | ctx_t | h_dec_{t-1}]; row 0 = 0 (rows 0..1 are cleared) */
    float* CA;    /* (T+1,B,1024) pre-dropout cell of attention_rnn; row 0 = 0 */
    float* CD;    /* (T+1,B,1024) pre-dropout cell of decoder_rnn;   row 0 = 0 */
    float* GA;    /* (T,B,4096) gate activations i,f,g,o of attention_rnn; NULL (with GD, S) = forward only */
    float* GD;    /* (T,B,4096) gate activations of decoder_rnn */
    float* QP;    /* (t2v_decoder_qp_floats(B, T_in)) scratch: per-workgroup partial queries, sync / error words, the
                     decode loop's exchange area, then the attention kernel's energy-exchange granules */
    float* AL;    /* (T+1,B,T_in) AL[t+1] = attention weights of step t; row 0 = 0 */
    float* ACUM;  /* (T+1,B,T_in) cumulative weights; row 0 = 0 */
    float* S;     /* (T,B,T_in,128) tanh(...) of the energies; overwritten with dpre by bwd; NULL = forward only */
} t2v_dec_train_bufs;

/* floats the QP scratch needs for (B, T_in): B*256*128 + 4160 + 2*B*8*ceil16(T_in) */
long t2v_decoder_qp_floats(int B, int T_in);

/* Decoder.forward's time loop (model.py:415-421 → Decoder.decode 346-389 → Attention.forward
 * 67-88), teacher forced, both LSTM cells + location-sensitive attention.  The 80-mel/gate
 * projection (model.py:385-388) is hoisted out of the loop: it reads [h_dec_t | ctx_t] from XS.
 * Dropout on the LSTM states (model.py:361-364,378-381) uses a counter-based RNG keyed by
 * (seed, stream, t, b, unit); p = 0 disables it.  B <= 16 per call (callers split larger batches: items are
 * independent inside the loop); 1 <= T_in <= 4096 (the reference is unbounded; koemo reaches 555 symbols). */
int t2v_decoder_train_fwd(const t2v_dec_weights* w, const t2v_dec_train_bufs* s,
                          int B, int T_in, int T_out, float p_att, float p_dec,
                          uint64_t seed, void* stream);
/* The same forward loop as ONE persistent launch (csrc/decoder_train_persist.hip): 256 workgroups stay resident for all
 * T_out steps, the LSTM weights of both cells live in registers (read once per pass from the nn.LSTMCell tensors — no
 * packs), attention_rnn -> attention -> attention_rnn is the only per-step dependency chain (teacher forcing: decoder_rnn
 * trails one step behind on the same workgroups), and the recurrent state travels between CUs THROUGH a sentinel-filled
 * copy of the saved activations (`scratch`): a word that is no longer 0xFFFFFFFF has been produced — no tags, no flags,
 * no grid barrier.  Writes the same arena as t2v_decoder_train_fwd (XS, CA, CD, GA, GD, AL, ACUM, S; QP only supplies
 * the error word), so t2v_decoder_train_bwd runs on it unchanged; dropout masks are the same counter-based ones, i.e.
 * the two paths agree to fp32 summation order.  Supported when t2v_decoder_train_persist_supported(B, T_in) != 0:
 * B <= 6, T_in <= 560 (round 6; up to 224 symbols the attention workgroups keep their W_q / processed-memory slices in LDS,
 * beyond in registers), a device with >= 256 CUs that can hold one 512-thread workgroup with this LDS carve per CU.
 * scratch: t2v_decoder_train_persist_scratch_floats(B, T_in, T_out) floats, 16-byte aligned (filled by the call). */
typedef struct t2v_dec_train_persist_weights {
    const float* w_ih_att; const float* w_hh_att;   /* attention_rnn (4096,768) [prenet | ctx], (4096,1024) */
    const float* w_ih_dec; const float* w_hh_dec;   /* decoder_rnn   (4096,1536) [h_att | ctx], (4096,1024) */
    const float* bias_dec;  /* (4096) bias_ih + bias_hh of decoder_rnn (attention_rnn's biases are inside gpre) */
    const float* wq;        /* (128,1024) query_layer weight (model.py:35), NOT transposed */
    const float* wcomb;     /* t2v_fuse_location_weights output */
    const float* v;         /* (128) */
} t2v_dec_train_persist_weights;
int t2v_decoder_train_persist_supported(int B, int T_in);
long t2v_decoder_train_persist_scratch_floats(int B, int T_in, int T_out);
int t2v_decoder_train_fwd_persistent(const t2v_dec_train_persist_weights* w, const t2v_dec_train_bufs* s, float* scratch,
                                     int B, int T_in, int T_out, float p_att, float p_dec, uint64_t seed, void* stream);

/* The same pass for hparams.bf16_run (BASELINE configs[4]: B = 16 per GPU; replaces reference fp16_optimizer.py:51-382 /
 * train.py:22-28 on the decoder loop model.py:415-421, 346-389): B <= 16, T_in <= 560.  The LSTM weights are rounded to bf16
 * (RNE, as t2v_pack_lstm_weights_bf16) into MFMA tiles that stay in registers for the whole pass — a workgroup owns 8 hidden
 * units of both cells, the batch is the N dimension of v_mfma_f32_16x16x32_bf16 — the recurrent state travels between the
 * workgroups as bf16 rows laid out as the MFMA's B operand; fp32 accumulation, cell state, attention and saved activations,
 * i.e. the arithmetic of t2v_decoder_train_fwd with bf16 packs (packs_bf16 = 1).  Same arena, same dropout masks.
 * scratch: t2v_decoder_train_persist16_scratch_floats(B, T_in, T_out) floats, 16-byte aligned (filled by the call). */
int t2v_decoder_train_persist16_supported(int B, int T_in);
long t2v_decoder_train_persist16_scratch_floats(int B, int T_in, int T_out);
int t2v_decoder_train_fwd_persistent16(const t2v_dec_train_persist_weights* w, const t2v_dec_train_bufs* s, float* scratch,
                                       int B, int T_in, int T_out, float p_att, float p_dec, uint64_t seed, void* stream);

/* Measurement aid for bench.py: re-issues only the selected kernels of a finished forward pass on
 * its saved arena (bit0 = k_lstm_fwd256 (both LSTM cells), bit1 = k_attn_fwd),
 * so their average launch duration can be bracketed with events on `stream`.  Results are
 * bit-identical to the first pass (the kernels are pure functions of the arena). */
int t2v_decoder_replay_fwd_kernels(const t2v_dec_weights* w, const t2v_dec_train_bufs* s,
                                   int B, int T_in, int T_out, float p_att, float p_dec,
                                   uint64_t seed, int kernel_mask, void* stream);

typedef struct t2v_dec_bwd_bufs {
    const float* dHC;  /* (T,B,1536) grad wrt [h_dec_t | ctx_t] coming from the projection */
    float* DGA;   /* (T,B,4096) out: grad wrt attention_rnn pre-activations (== grad of gpre) */
    float* DGD;   /* (T,B,4096) out: grad wrt decoder_rnn pre-activations */
    float* DQ;    /* (T,B,S,128,2) out: per-position-slice partials of the grad wrt the processed query as 8-byte
                     {value, tag} granules ([...,0] = value; sum over dim 2), S = t2v_attn_bwd_slices(T_in) */
    float* DCTX;  /* (T,B,512)  out: grad wrt attention context */
    float* YD;    /* (B,2560) scratch, zeroed by the call */
    float* YA;    /* (B,1536) scratch, zeroed by the call */
    float* DCA;   /* (B,1024) scratch */
    float* DCD;   /* (B,1024) scratch */
    float* GPREV; /* (2,B,S,2,64) scratch: per-slice partial alignment-window gradients, parity double-buffered */
    float* GCUM;  /* (B*S*ceil16(T_in) + 64) scratch: per-workgroup copies of the cumulative-weights gradient, then
                     sync words ([1] != 0 afterwards: a bounded spin timed out) */
    float* DV;    /* (B,S,128) out: per-item, per-slice grad of attention v (sum over dims 0,1 = dv) */
} t2v_dec_bwd_bufs;

/* position slices per item of the attention backward: ceil(T_in / 16) for T_in <= 128, else ceil(T_in / 32) */
int t2v_attn_bwd_slices(int T_in);

/* Hand-written BPTT of the loop above (what autograd does for the reference at train.py:225).
 * On return S holds dpre = grad wrt (q + loc + pm) per (t,b,j,d). */
int t2v_decoder_train_bwd(const t2v_dec_weights* w, const t2v_dec_train_bufs* s,
                          const t2v_dec_bwd_bufs* g, int B, int T_in, int T_out,
                          float p_att, float p_dec, uint64_t seed, void* stream);

/* Measurement aid like t2v_decoder_replay_fwd_kernels for the reverse pass (bit0 = k_lstm_bwd256, bit1 = k_attn_cell_bwd) on the
 * buffers of a finished backward.  Timing only: S already holds dpre, so the replayed attention backward is not a
 * second valid gradient (the launch shapes, operand sizes and hand-offs are those of the real pass). */
int t2v_decoder_replay_bwd_kernels(const t2v_dec_weights* w, const t2v_dec_train_bufs* s,
                                   const t2v_dec_bwd_bufs* g, int B, int T_in, int T_out,
                                   float p_att, float p_dec, uint64_t seed, int kernel_mask, void* stream);

/* Deferred weight gradients of the location layer (model.py:24-28) reduced over the whole decoder pass in one
 * streaming kernel + a fixed-order partial sum (what autograd accumulates step by step at train.py:225), through
 * the fused filter bank:  dW_comb[d][c,k] = sum_{t,b,j} dpre[t,b,j,d] * a_c[t,b,j+k-15], then
 *   d_loc_dense (128,32)  = dW_comb · loc_conv^T,   d_loc_conv (32,2,31) = loc_dense^T · dW_comb.
 * dpre = S after t2v_decoder_train_bwd, al/acum = rows 0..T-1 of AL/ACUM (a_0 / a_1).
 * part_scratch: t2v_attn_wgrad_scratch_floats() floats. */
int t2v_attn_wgrad_scratch_floats(void);
int t2v_attn_wgrad(const float* dpre, const float* al, const float* acum, const float* loc_conv,
                   const float* loc_dense, float* part_scratch, float* d_loc_dense, float* d_loc_conv,
                   int B, int T_in, int T, void* stream);


/* ------------------------------------------------------------------ persistent BPTT (csrc/decoder_train_bwd_persist.hip)
 * The reverse recurrence of the decoder loop as persistent launches for the shapes of the persistent forward
 * (t2v_decoder_bwd_persist_supported: B <= 6, T_in <= 576 — 96-position attention slices beyond 224 symbols —, >= 256 CUs).
 * decoder_rnn's chain does not depend on the attention path, so it runs first for all steps:
 *   t2v_decoder_bwd_dchain : dHC[:, :, :1024] (grad wrt h_dec from the projection), GD, CD (saved by the forward pass)
 *                            -> DGD (T,B,4096), grad wrt decoder_rnn's pre-activations.  256 workgroups x 4 hidden units,
 *                            W_hh_dec^T in registers; the gate-gradient rows travel between CUs through `scratch`
 *                            (t2v_decoder_bwd_dchain_scratch_floats(B, T_out) floats, 16-byte aligned, filled by the call).
 * err_word (1 x uint32, zeroed by the call): != 0 afterwards = a bounded spin timed out. */
int t2v_decoder_bwd_persist_supported(int B, int T_in);
long t2v_decoder_bwd_dchain_scratch_floats(int B, int T_out);
int t2v_decoder_bwd_dchain(const float* w_hh_dec, const float* dHC, const float* GD, const float* CD, float* DGD,
                           float* scratch, uint32_t* err_word, int B, int T_out, float p_dec, uint64_t seed, void* stream);
/* The WHOLE reverse pass as ONE persistent launch (k_achain_bwd): attention_rnn + attention form the per-step dependency
 * chain (all-gather dga(t+1) -> Wcat_att^T columns -> d ctx(t) -> attention(t) backward on position-split workgroups ->
 * dq(t) -> W_q^T, cell backward -> dga(t)); decoder_rnn's chain (cell backward, all-gather of dgd, Wcat_dec^T columns)
 * runs one step ahead on the same workgroups in the shadow of the attention workgroups.  All transposed LSTM weight
 * columns live in registers (read once per pass from the nn.LSTMCell tensors).  Same outputs as t2v_decoder_train_bwd:
 *   DGA, DGD (T,B,4096), DCTX (T,B,512), S overwritten with dpre, DV (B,S,128) per-slice partial dv (sum dims 0,1),
 *   DQP (T,B,S,128) per-slice partial dq rows (sum over dim 2 = dq of a step), S = t2v_attn_bwd_slices(T_in).
 * w: the struct of the persistent forward (bias_dec unused); s: the forward pass's arena (gpre, QP unused);
 * scratch: t2v_decoder_bwd_achain_scratch_floats(B, T_in, T_out) floats, 16-byte aligned; DQP 16-byte aligned. */
long t2v_decoder_bwd_achain_scratch_floats(int B, int T_in, int T_out);
/* position slices per item of the ONE-LAUNCH reverse pass (second-to-last dimension of its DQP (T,B,S,128) and DV (B,S,128)):
 * t2v_attn_bwd_slices(T_in) up to 224 symbols, ceil(T_in / 96) beyond */
int t2v_decoder_bwd_persist_slices(int T_in);
int t2v_decoder_bwd_achain(const t2v_dec_train_persist_weights* w, const float* reserved, const t2v_dec_train_bufs* s,
                           const float* dHC, float* DGA, float* DGD, float* DCTX, float* DV, float* DQP, float* scratch,
                           uint32_t* err_word, int B, int T_in, int T_out, float p_att, float p_dec, uint64_t seed,
                           void* stream);
/* The reverse pass for hparams.bf16_run, B <= 16 (BASELINE configs[4]; the bf16 counterpart of t2v_decoder_bwd_achain: same
 * arithmetic as t2v_decoder_train_bwd with bf16 packs — transposed LSTM weights and gate gradients rounded to bf16 in front of
 * every product, fp32 accumulation, fp32 cell / attention backward).  ONE persistent launch: Wcat^T lives in registers as bf16
 * MFMA tiles, cut into 128-column x 1024-row blocks (48 + 80 workgroups publish partial column sums), 16 + 16 workgroups run
 * the cells of 64 hidden units each, B * S workgroups the attention backward (S = t2v_decoder_bwd_persist16_slices(T_in):
 * slices of 16 positions up to 96 symbols, 32 up to 192, 96 beyond).  Supported when B <= 16, T_in <= 560 (B * S <= 96 always holds).
 * DV (B,S,128), DQP (T_out,B,S,128), scratch: t2v_decoder_bwd_persist16_scratch_floats() floats, 16-byte aligned (filled by
 * the call); dq(t) summed over the slices lies at float offset t2v_decoder_bwd_persist16_dq_offset() of scratch as
 * (T_out, 16, 128).  err_word: set to 1 when a bounded spin gave up (results invalid). */
int t2v_decoder_bwd_persist16_supported(int B, int T_in);
int t2v_decoder_bwd_persist16_slices(int T_in);
/* ... and at this T_out: 1 when every exchange array stays below the 2^31-byte reach of the kernel's buffer offsets (the widest
 * one reaches it at T_out = 3 277) — what a caller asks before its FORWARD pass commits to this reverse pass. */
int t2v_decoder_bwd_persist16_fits(int B, int T_in, int T_out);
long t2v_decoder_bwd_persist16_scratch_floats(int B, int T_in, int T_out);
long t2v_decoder_bwd_persist16_dq_offset(int B, int T_in, int T_out);
int t2v_decoder_bwd_persistent16(const t2v_dec_train_persist_weights* w, const t2v_dec_train_bufs* s, const float* dHC,
                                 float* DGA, float* DGD, float* DCTX, float* DV, float* DQP, float* scratch,
                                 uint32_t* err_word, int B, int T_in, int T_out, float p_att, float p_dec, uint64_t seed,
                                 void* stream);
/* Its preparation (error word, sentinel fills of scratch and DQP: functions of nothing but the buffers, so a training step
 * issues it on a side stream right behind the decoder forward) as a call of its own, and the pass without it. */
int t2v_decoder_bwd_persistent16_prepare(float* DQP, float* scratch, uint32_t* err_word, int B, int T_in, int T_out, void* stream);
int t2v_decoder_bwd_persistent16_prepared(const t2v_dec_train_persist_weights* w, const t2v_dec_train_bufs* s, const float* dHC,
                                          float* DGA, float* DGD, float* DCTX, float* DV, float* DQP, float* scratch,
                                          uint32_t* err_word, int B, int T_in, int T_out, float p_att, float p_dec, uint64_t seed,
                                          void* stream);

/* float offset inside `scratch` of dq(t) summed over the position slices, (T_out,B,128) floats, valid when the pass has ended
 * (-1: shape outside the persistent range) */
long t2v_decoder_bwd_achain_dq_offset(int B, int T_in, int T_out);
/* The same pass as TWO launches (round 4): the attention chain on `stream`, the free-running decoder_rnn chain on `stream_d`
 * (which this call orders behind the preparation launches on `stream`).  DGD is complete when stream_d is — the decoder_rnn
 * weight-gradient GEMMs can be queued behind it and run while the attention chain is still going — everything else when
 * `stream` is; the caller joins stream_d back.  stream_d NULL or == stream: one launch, as t2v_decoder_bwd_achain. */
int t2v_decoder_bwd_achain2(const t2v_dec_train_persist_weights* w, const float* reserved, const t2v_dec_train_bufs* s,
                           const float* dHC, float* DGA, float* DGD, float* DCTX, float* DV, float* DQP, float* scratch,
                           uint32_t* err_word, int B, int T_in, int T_out, float p_att, float p_dec, uint64_t seed,
                           void* stream, void* stream_d);
/* The preparation of that pass as a call of its own (error word, sentinel fills, the factor arrays of both cells — functions
 * of the forward pass's saved activations alone, so a training step issues it right after the decoder forward, on a side
 * stream, next to the Postnet), and the pass without it.  scratch / DQP / err_word: the same buffers in both calls. */
int t2v_decoder_bwd_achain_prepare(const t2v_dec_train_bufs* s, float* DQP, float* scratch, uint32_t* err_word, int B, int T_in,
                                   int T_out, float p_att, float p_dec, uint64_t seed, void* stream);
int t2v_decoder_bwd_achain_prepared(const t2v_dec_train_persist_weights* w, const float* reserved, const t2v_dec_train_bufs* s,
                           const float* dHC, float* DGA, float* DGD, float* DCTX, float* DV, float* DQP, float* scratch,
                           uint32_t* err_word, int B, int T_in, int T_out, float p_att, float p_dec, uint64_t seed,
                           void* stream, void* stream_d);

/* ------------------------------------------------------------------ free-running decode
 * Decoder.inference (model.py:428-464) == the synthesizer loop (synthesizer.py:139-154): steps
 * t_begin..t_end-1, each = attention_rnn → attention → decoder_rnn → 80-mel/gate projection → Prenet of
 * the new frame (dropout p_prenet stays on at inference, model.py:101).  The caller runs chunks and reads
 * `stop_flag` (first t with sigmoid(gate) > gate_threshold for every item, INT_MAX if none) between
 * them.  external_prenet != 0: PRE[t] is supplied by the caller for every step (the per-step
 * `decoder.prenet(x)` + `decoder.decode(x)` call sequence) and no Prenet is evaluated here.
 * Arena rows as in t2v_dec_train_bufs (XS (Tmax+2,B,2560) rows 0,1 zeroed; CA/CD/AL/ACUM row 0 zeroed);
 * PRE[0] = Prenet(go frame) is written by the caller.  B <= 8, T_in <= 4096; QP as in t2v_dec_train_bufs. */
typedef struct t2v_dec_infer_bufs {
    const float* memory;     /* (B,T_in,512) */
    const float* pm;         /* (B,T_in,128) */
    const int32_t* lengths;  /* NULL at inference (mask=None, model.py:441) */
    float* XS; float* CA; float* CD; float* QP; float* AL; float* ACUM;
    float* PRE;              /* (Tmax+1,B,256) prenet outputs */
    float* MEL;              /* (Tmax,B,80) out */
    float* GATE;             /* (Tmax,B)    out */
    int32_t* stop_flag;      /* (1) */
    const float* prenet_w1;  /* (256,256) Prenet layer 1 */
    const float* proj_w;     /* (337,1536) rows 0..79 = linear_projection, row 80 = gate_layer, rows 81..336 = W0·linear_projection
                                (Prenet layer 0 is bias-free and linear in the mel frame: folded so it need not wait for it) */
    const float* proj_b;     /* (337) linear_projection / gate_layer biases, then W0·b_projection */
} t2v_dec_infer_bufs;

int t2v_decoder_infer_steps(const t2v_dec_weights* w, const t2v_dec_infer_bufs* s, int B, int T_in,
                            int t_begin, int t_end, float gate_threshold, float p_prenet,
                            int external_prenet, uint64_t seed, void* stream);

/* The same loop as ONE persistent launch (csrc/decoder_persist.hip): 256 workgroups stay resident for the whole
 * utterance, all weights in registers / LDS, the per-frame state vectors travel between CUs as 4-byte values in a sentinel-filled row per frame,
 * the loop ends on the frame the gate fires (nothing is computed after it).  Runs frames 0..t_end-1 from the zero state;
 * pre_first = Prenet(go frame).  Supported when t2v_decoder_persist_supported(B, T_in) != 0 (B <= 4 and the attention
 * operands of T_in positions fit the 160 KB LDS: T_in <= 224 at B = 1, 192 at B = 2, 160 at B = 3, 128 at B = 4); everything else
 * takes t2v_decoder_infer_steps.
 * Weights are the nn.LSTMCell / LinearNorm tensors themselves (no packing). */
typedef struct t2v_dec_persist_weights {
    const float* w_ih_att; const float* w_hh_att;   /* (4096,768) [prenet | ctx], (4096,1024) */
    const float* w_ih_dec; const float* w_hh_dec;   /* (4096,1536) [h_att | ctx], (4096,1024) */
    const float* bias_att; const float* bias_dec;   /* (4096) b_ih + b_hh */
    const float* wq;        /* (128,1024) query_layer weight */
    const float* wcomb;     /* t2v_fuse_location_weights output */
    const float* v;         /* (128) */
    const float* proj_w;    /* (337,1536) as in t2v_dec_infer_bufs */
    const float* proj_b;    /* (337) */
    const float* prenet_w1; /* (256,256) */
} t2v_dec_persist_weights;
typedef struct t2v_dec_persist_bufs {
    const float* memory;     /* (B,T_in,512) */
    const float* pm;         /* (B,T_in,128) */
    const int32_t* lengths;  /* NULL at inference */
    const float* pre_first;  /* (B,256) */
    float* MEL;              /* (t_end,B,80) out */
    float* GATE;             /* (t_end,B) out */
    float* AL;               /* (t_end+1,B,T_in) out: row t+1 = attention weights of frame t */
    int32_t* stop_flag;      /* (1): first frame on which every item's gate fired (caller presets INT_MAX) */
    void* exchange;          /* t2v_decoder_persist_scratch_floats(B, t_end) floats of exchange rows, 16-byte aligned (filled by the call) */
    uint32_t* err_word;      /* (1): != 0 afterwards: a bounded spin timed out */
} t2v_dec_persist_bufs;
long t2v_decoder_persist_scratch_floats(int B, int t_end);
int t2v_decoder_persist_supported(int B, int T_in);
int t2v_decoder_infer_persistent(const t2v_dec_persist_weights* w, const t2v_dec_persist_bufs* s, int B, int T_in,
                                 int t_end, float gate_threshold, float p_prenet, uint64_t seed, void* stream);

/* ------------------------------------------------------------------ Conv1d + BatchNorm1d + activation
 * The encoder conv bank (model.py:159-177) and the Postnet (model.py:110-148): stride-1 "same" Conv1d as
 * an implicit GEMM on fp32 MFMA, BatchNorm1d (train: biased batch statistics over B*T incl. padded frames;
 * eval: running statistics), tanh / ReLU / none, dropout.  All activations are (B, C, T) fp32.
 *   t2v_conv1d_fwd : Y = conv(X, W) + bias; stat_part ((t2v_conv1d_stat_blocks(B,T,Cin,Cout,KS), Cout, 2) or NULL)
 *                    receives per-column-block partial [sum, sum of squares] per channel.
 *   t2v_bn_act_fwd : out = dropout(act(BN(y)));  act 0 none / 1 tanh / 2 relu.  training != 0 finalises
 *                    the statistics from stat_part, writes mean/rstd for the backward and updates the
 *                    running buffers (momentum, unbiased variance).
 *   t2v_bn_act_bwd : dy (grad wrt the conv output), dgamma, dbeta from dout.  dconv_bias ((M) or NULL) receives the
 *                    gradient of the bias of the convolution that feeds this training-mode BatchNorm, which is
 *                    identically zero (dy has zero mean per channel): written here so the caller launches no fill.
 *   t2v_conv1d_bwd : dX (may be NULL; needs Wt_scratch of W's size) and dW (may be NULL).  W == NULL with dX: Wt_scratch
 *                    already holds the flipped, transposed weight (t2v_conv1d_flip_weights ran earlier in the step); W is not
 *                    read when only dW is asked for.
 *   t2v_conv1d_flip_weights : Wt[i] (Cin, Cout, KS) = W[i] (Cout, Cin, KS) transposed with the taps reversed — the operand
 *                    of the data-gradient convolution — for n <= 16 layers in ONE launch (host arrays of n pointers / dims). */
int t2v_conv1d_stat_blocks(int B, int T, int Cin, int Cout, int KS);
/* the same for t2v_conv1d_fwd_bf16 (bf16_run keeps the tiles of its own kernels) */
int t2v_conv1d_stat_blocks_bf16(int B, int T, int Cin, int Cout, int KS);
int t2v_conv1d_fwd(const float* W, const float* X, const float* bias, float* Y, float* stat_part,
                   int B, int Cin, int T, int Cout, int KS, void* stream);
int t2v_conv1d_dw_scratch_floats(int B, int Cin, int T, int Cout, int KS);   /* 0 -> dw_scratch may be NULL */
int t2v_conv1d_bwd(const float* W, const float* X, const float* dY, float* dX, float* dW, float* Wt_scratch,
                   float* dw_scratch,
                   int B, int Cin, int T, int Cout, int KS, void* stream);
int t2v_conv1d_flip_weights(const float* const* W, float* const* Wt, const int* Cout, const int* Cin, int KS, int n,
                            void* stream);
/* bf16_run variants (BASELINE configs[4]; replace the reference's fp16 path, fp16_optimizer.py / loss_scaler.py):
 * fp32 tensors in and out, operands rounded to bf16 on the way into LDS, fp32 accumulation on bf16 MFMA.
 * Wp_scratch: W's element count x 2 bytes.  Round 5: the weight gradient as well (dY and X rounded to bf16 while staged,
 * v_mfma_f32_16x16x16_bf16; W and Wp_scratch are not needed for a dW-only call, dw_scratch as t2v_conv1d_dw_scratch_floats);
 * T2V_ERR_DIMS unless KS == 5 and the reduction channel count is a multiple of 16. */
int t2v_conv1d_fwd_bf16(const float* W, const float* X, const float* bias, float* Y, float* stat_part,
                        void* Wp_scratch, int B, int Cin, int T, int Cout, int KS, void* stream);
int t2v_conv1d_bwd_bf16(const float* W, const float* X, const float* dY, float* dX, float* dW,
                        void* Wp_scratch, float* dw_scratch, int B, int Cin, int T, int Cout, int KS, void* stream);

int t2v_bn_act_fwd(const float* y, const float* stat_part, int nblk, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, float* mean_out, float* rstd_out, float* out,
                   int B, int M, int T, int act, int training, float p_drop, float momentum, float eps,
                   uint64_t seed, uint32_t rng_stream, uint32_t rng_t, void* stream);
int t2v_bn_act_bwd(const float* y, const float* dout, const float* mean, const float* rstd,
                   const float* gamma, const float* beta, float* dy, float* dgamma, float* dbeta,
                   float* dconv_bias, int B, int M, int T, int act, float p_drop, uint64_t seed,
                   uint32_t rng_stream, uint32_t rng_t, void* stream);
/* the same for eval-mode BatchNorm (running statistics are constants: no batch-statistic terms in dy, dconv_bias =
 * gamma rstd sum(dz) instead of zeros); mean = running_mean, rstd = 1 / sqrt(running_var + eps) */
int t2v_bn_act_bwd_eval(const float* y, const float* dout, const float* mean, const float* rstd,
                   const float* gamma, const float* beta, float* dy, float* dgamma, float* dbeta,
                   float* dconv_bias, int B, int M, int T, int act, float p_drop, uint64_t seed,
                   uint32_t rng_stream, uint32_t rng_t, void* stream);

/* ------------------------------------------------------------------ symbol embedding
 * nn.Embedding(n_symbols, C) of the text encoder (model.py:474-482) as used at model.py:528
 * (`transcript_embedding(text).transpose(1, 2)`): ids (B,T) int64 -> out (B,C,T) channel-major, the layout the first
 * encoder Conv1d reads.  Backward: dW (n_symbols,C) = per-symbol sum of dy (B,C,T) over the positions holding that
 * symbol, ascending position order (deterministic). */
int t2v_embedding_fwd(const long long* ids, const float* W, float* out_bct, int B, int T, int C, int n_symbols, void* stream);
int t2v_embedding_bwd(const long long* ids, const float* dy_bct, float* dW, int B, int T, int C, int n_symbols, void* stream);

/* backward of the GEMM's relu / dropout epilogue (Prenet, model.py:96-99): out = dy * [y != 0] * scale, where y is the
 * forward output (already relu'd and scaled by 1/(1-p)) and scale = 1/(1-p); n contiguous floats, 16-byte aligned. */
int t2v_gemm_epilogue_bwd(const float* dy, const float* y, float* out, size_t n, float scale, void* stream);

/* ------------------------------------------------------------------ column sums (bias gradients)
 * out[j] = sum_i A[i*lda + j] for a row-major (M,N) matrix: what autograd computes as `grad.sum(0)` for the bias of
 * every nn.Linear / LSTM cell on the path (model.py:171-190, 200-236) and for the sum over decoder steps of the
 * processed-memory gradient (model.py:60-64).  Fixed slice boundaries and summation order (bit-reproducible).
 * scratch: t2v_colsum_scratch_floats(M,N) floats (may be 0 -> NULL). */
long t2v_colsum_scratch_floats(long M, long N);
int t2v_colsum(const float* A, long lda, long M, long N, float* scratch, float* out, void* stream);

/* ------------------------------------------------------------------ fused epilogues (round 4, csrc/glue.hip)
 * t2v_mask_outputs: Tacotron2.parse_output (model.py:509-520) in one launch — mel (B,C,T) and mel_post (B,C,T) := 0,
 * gate (B,T) := gate_fill (1e3) at frames t >= lengths[b]; in place (the reference fills `.data`).
 * t2v_reparam_fwd / _bwd: VAE_GST.reparameterize (modules.py:74-81): z = eps * exp(0.5 logvar) + mu over n floats, and
 * dlogvar = dz * eps * exp(0.5 logvar) * 0.5 (dmu = dz needs no launch).
 * t2v_gather_words: *dst[i] = *src[i] for n four-byte device words (16 per launch): the asynchronous error ledger of the
 * host mirror collects the error words of a step's cooperative kernels with it. */
int t2v_mask_outputs(float* mel, float* mel_post, float* gate, const int* lengths, int B, int C, int T, float gate_fill,
                     void* stream);
int t2v_reparam_fwd(const float* eps, const float* mu, const float* logvar, float* z, long n, void* stream);
int t2v_reparam_bwd(const float* dz, const float* eps, const float* logvar, float* dlogvar, long n, void* stream);
int t2v_gather_words(const void* const* src, void* const* dst, int n, void* stream);
/* out[r] = [a[r][0..na) | b[r][0..nb)], rows of float4-aligned widths / strides (lda, ldb in floats): the (h_dec, context) input
 * rows of linear_projection + gate_layer (model.py:385-388) gathered from the decoder arena in one launch */
int t2v_concat2_rows(const float* a, long lda, int na, const float* b, long ldb, int nb, float* out, long rows, void* stream);

/* ------------------------------------------------------------------ encoder BiLSTM recurrence
 * nn.LSTM(512, 256, bidirectional) on a packed sequence (model.py:171-173, 183-190), recurrent part only:
 * gx (2,B,T,1024) = X·W_ih^T + b_ih + b_hh per direction (time-batched GEMM done by the caller), whh
 * (2,1024,256).  Persistent cooperative kernels (16 workgroups, W_hh register-resident for all T steps).
 * y (B,T,512) and dg (2,B,T,1024) are cleared by the calls themselves (padded positions stay zero);
 * hx_scratch (2*2*16*256 8-byte granules = 2*2*2*16*256 floats) / dgx_scratch (2*2*16*16*256 granules = 2*2*2*16*16*256
 * floats: per direction and step parity, the 256 partial recurrent gradients x 16 items of each of 16 producers) are exchange
 * buffers of {value, step tag} granules (zeroed by the call); sync3 = 3 uint32 (zeroed by the call; sync3[2] != 0
 * afterwards means a bounded spin timed out).  gates/cells (saved
 * activations) may be NULL for inference.  B <= 16. */
int t2v_bilstm_fwd(const float* gx, const float* whh, const int32_t* lengths, float* y, float* gates,
                   float* cells, float* hx_scratch, uint32_t* sync3, int B, int T, void* stream);
int t2v_bilstm_bwd(const float* whh, const int32_t* lengths, const float* dy, const float* gates,
                   const float* cells, float* dg, float* dgx_scratch, uint32_t* sync3, int B, int T,
                   void* stream);

/* ------------------------------------------------------------------ dense layers (time-batched)
 * C[i][j] (+)= sum_k A[i*sAi + k*sAk] * B[j*sBj + k*sBk] (+ bias[j]), optional ReLU and dropout epilogue
 * (dropout mask = counter RNG keyed by (seed, rng_stream, rng_t, i*ldc+j)).  Hand-written fp32 MFMA tile
 * for Prenet (model.py:91-102), memory_layer (model.py:290), the hoisted attention_rnn input term and the
 * fused linear_projection+gate_layer (model.py:385-388) — forward (NT) and both gradients (NN / TN). */
int t2v_gemm_f32(const float* A, long sAi, long sAk, const float* B, long sBj, long sBk, const float* bias,
                 float* C, int ldc, int M, int N, int K, int relu, int accumulate, float p_drop,
                 uint64_t seed, uint32_t rng_stream, uint32_t rng_t, void* stream);
/* Same product with split-K for skinny deep-K shapes (weight gradients of small layers with K = T*B): when
 * t2v_gemm_splitk_scratch_floats(M,N,K) > 0 and `splitk_scratch` holds that many floats, the k range is cut over
 * gridDim.z and the partial tiles are summed in a fixed order together with the epilogue (deterministic); otherwise
 * identical to t2v_gemm_f32. */
long t2v_gemm_splitk_scratch_floats(int M, int N, int K);
int t2v_gemm_f32_splitk(const float* A, long sAi, long sAk, const float* B, long sBj, long sBk, const float* bias,
                        float* C, int ldc, int M, int N, int K, int relu, int accumulate, float p_drop,
                        uint64_t seed, uint32_t rng_stream, uint32_t rng_t, float* splitk_scratch, void* stream);

/* Round 6: how the LARGE fp32 products (>= 64 tiles of 128x128, 16-byte-aligned operand runs) are computed.
 * mode 1 (default; T2V_F32_GEMM=native selects 0): "x3" — every fp32 operand is cut exactly into three bf16 values
 * (a = a0 + a1 + a2) and the product is accumulated in fp32 from six v_mfma_f32_32x32x16_bf16 per k-block; the omitted
 * cross terms are <= 2^-25 |a b|, below the rounding of one fp32 product: fp32-class results (same error bound against fp64
 * as the fp32-MFMA kernel, tests/test_gemm_gpu.py) at up to 2.7x the fp32 matrix peak of gfx950 (which has no TF32 path).
 * mode 0: v_mfma_f32_32x32x2_f32 for every product.  Pass -1 to query.  Returns the previous mode.  Process-wide. */
int t2v_gemm_f32_set_mode(int x3);

/* Grouped form of the large fp32 products (round 6): for every group g and part p   C[g][p] (+)= A_g · B_{g,p}^T   with all A_g
 * (M x K) and B_{g,p} (N[p] x K), element strides as in t2v_gemm_f32.  The decoder's four nn.LSTMCell weight gradients
 * (reference model.py:221-226 under autograd) are two groups — A = the gate gradients of a cell, parts = the column blocks
 * [prenet | h_att | ctx] / [h_att + ctx | h_dec] of its input: in x3 mode (t2v_gemm_f32_set_mode) every operand is split ONCE and
 * all tiles run as ONE launch (1 024 tiles = two full rounds of the chip, no k-split); otherwise, or when an N[p] is not a
 * multiple of 128, the products run one by one on t2v_gemm_f32.  ngroups <= 2, nb <= 3.  scratch:
 * t2v_gemm_f32_grouped_scratch_floats() floats, 16-byte aligned. */
typedef struct t2v_gemm_group {
    const float* A;
    long sAi;
    long sAk;
    int nb;
    const float* B[3];
    long sBj[3];
    long sBk[3];
    int N[3];
    float* C[3];
    int ldc[3];
} t2v_gemm_group;
long t2v_gemm_f32_grouped_scratch_floats(const t2v_gemm_group* groups, int ngroups, int M, int K);
int t2v_gemm_f32_grouped(const t2v_gemm_group* groups, int ngroups, int M, int K, int accumulate, float* scratch, void* stream);
/* bf16_run form: the same grouping on ONE bf16 plane per operand (operands rounded to bf16, RNE, once by the split pass; fp32
 * accumulation — the arithmetic of t2v_gemm_bf16). */
long t2v_gemm_bf16_grouped_scratch_floats(const t2v_gemm_group* groups, int ngroups, int M, int K);
int t2v_gemm_bf16_grouped(const t2v_gemm_group* groups, int ngroups, int M, int K, int accumulate, float* scratch, void* stream);

/* ... and the fp32 k = 5 Conv1d forward / data gradient (t2v_conv1d_fwd / t2v_conv1d_bwd) on the same six-product scheme
 * (csrc/conv_x3.hip).  mode 0: never; 1: every eligible shape (KS = 5, Cin % 16 == 0, Cin, Cout >= 64); 2 (default; T2V_CONV_X3
 * presets it): launches of >= 192 tiles of 128 x 128 only — at the B = 6 step's shapes its fixed costs (split passes, channel-split
 * tiles) outweigh the faster loop, at B = 16 it runs a Postnet layer in 152 us against 212.  Only effective while
 * t2v_gemm_f32_set_mode is 1.  t2v_conv1d_stat_blocks answers for the current mode.  Pass -1 to query.  Returns the previous mode. */
int t2v_conv1d_x3_set_mode(int mode);

/* nbatch independent products C_z = A_z · B_z^T (z-th operands at A + z*sAb, B + z*sBb, C + z*sCb; element strides as in
 * t2v_gemm_f32, no bias / epilogue) in one launch.  Replaces the per-item loop that autograd's bmm backward of
 * `attention_context = torch.bmm(attention_weights.unsqueeze(1), memory)` (model.py:84-85) amounts to for d_memory. */
int t2v_gemm_f32_batched(const float* A, long sAb, long sAi, long sAk, const float* B, long sBb, long sBj, long sBk,
                         float* C, long sCb, int ldc, int nbatch, int M, int N, int K, void* stream);
/* same contract, bf16 operands / fp32 accumulate (bf16_run) */
int t2v_gemm_bf16(const float* A, long sAi, long sAk, const float* B, long sBj, long sBk, const float* bias,
                  float* C, int ldc, int M, int N, int K, int relu, int accumulate, float p_drop,
                  uint64_t seed, uint32_t rng_stream, uint32_t rng_t, void* stream);
/* bf16 product with split-K for the 128x128-tile kernel (the deferred LSTM weight gradients of the decoder, model.py:221-226
 * under autograd: 64 .. 384 tiles with K = T*B): when t2v_gemm_bf16_splitk_scratch_floats(M,N,K) > 0 and `splitk_scratch` holds
 * that many floats, the k range is cut over gridDim.z, the partial accumulators are summed in a fixed order (deterministic)
 * with the epilogue; otherwise identical to t2v_gemm_bf16. */
long t2v_gemm_bf16_splitk_scratch_floats(int M, int N, int K);
int t2v_gemm_bf16_splitk(const float* A, long sAi, long sAk, const float* B, long sBj, long sBk, const float* bias,
                         float* C, int ldc, int M, int N, int K, int relu, int accumulate, float p_drop,
                         uint64_t seed, uint32_t rng_stream, uint32_t rng_t, float* splitk_scratch, void* stream);


/* ------------------------------------------------------------------ reference encoder / VAE / loss
 * t2v_conv2d_s2_* : Conv2d 3x3 stride 2 pad 1 of ReferenceEncoder (modules.py:45-57); coord != 0 appends the
 *                   CoordConv channels xx, yy, rr (CoordConv.py:37-74) to x on the fly (w then has Cx+3 inputs).
 *                   BatchNorm2d+ReLU = t2v_bn_act_* on the (B, C, H*W) view with stat_part = NULL.
 * t2v_gru_*       : nn.GRU(256,256) recurrence (modules.py:60-62,78); gi = x·W_ih^T + b_ih outside; hs
 *                   (B,T+1,256) all hidden states (hs[:,0] = 0), gsave (B,T,4,256); dgi/dgh (B,T,768) gradients
 *                   wrt the input-side / hidden-side gate pre-activations.
 * t2v_loss_fwd_bwd: Tacotron2Loss_VAE (loss_function.py:27-44) value + gradients of the total in one launch;
 *                   out4 = [total, recon, kl, kl_weight]; part192 = 192 floats scratch, ticket = zeroed uint32. */
int t2v_conv2d_s2_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Cx, int H, int W,
                      int Cout, int coord, void* stream);
/* dw_scratch: t2v_conv2d_s2_dw_scratch_floats(...) floats (0 -> may be NULL): position-chunk partials of dW */
int t2v_conv2d_s2_dw_scratch_floats(int B, int Cx, int H, int W, int Cout, int coord);
int t2v_conv2d_s2_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* dw_scratch,
                      int B, int Cx, int H, int W, int Cout, int coord, void* stream);

/* The same convolution (forward / weight + data gradients) as batched GEMMs on the MFMA GEMM kernel (round 3; the reference
 * encoder's late layers have <= 39 output positions per item — far too few for a thread-per-output kernel).  `scratch` holds
 * t2v_conv2d_s2_gemm_scratch_floats(...) floats: the forward leaves the im2col matrix col[b][pos][c*9 + kh*3 + kw] there,
 * the backward needs it (same buffer, untouched in between) and overwrites it. */
long t2v_conv2d_s2_gemm_scratch_floats(int B, int Cx, int H, int W, int Cout, int coord);
int t2v_conv2d_s2_fwd_gemm(const float* x, const float* w, const float* bias, float* y, float* scratch, int B, int Cx, int H, int W,
                           int Cout, int coord, void* stream);
int t2v_conv2d_s2_bwd_gemm(const float* x, const float* w, const float* dy, float* dx, float* dw, float* scratch, int B, int Cx, int H,
                           int W, int Cout, int coord, void* stream);
/* xchg: 2*16*256 floats (forward) / 2*16*768 floats (backward) exchange buffer of the 8 cooperating workgroups;
 * sync2: 2 uint32 (zeroed by the call; sync2[1] != 0 afterwards means a bounded spin timed out). */
int t2v_gru_fwd(const float* gi, const float* whh, const float* bhh, float* hs, float* gsave, float* xchg,
                uint32_t* sync2, int B, int T, void* stream);
int t2v_gru_bwd(const float* whh, const float* hs, const float* gsave, const float* dh_last, float* dgi,
                float* dgh, float* xchg, uint32_t* sync2, int B, int T, void* stream);
int t2v_loss_fwd_bwd(const float* mel, const float* post, const float* mel_t, const float* gate,
                     const float* gate_t, const float* mu, const float* logvar, float* dmel, float* dpost,
                     float* dgate, float* dmu, float* dlogvar, float* part192, float* out4, uint32_t* ticket,
                     uint64_t n_mel, int n_gate, int n_lat, float kl_weight, void* stream);

/* ------------------------------------------------------------------ optimiser
 * clip_grad_norm_(params, max_norm) + Adam.step() of the reference loop (train.py:226-229,
 * Adam built at train.py:171-172) fused over one flat fp32 arena.  `grads` holds the SUM over
 * ranks after the all-reduce; inv_world = 1/world_size applies the averaging of
 * distributed.py:162.  partials: (1024) scratch; norm_out[0] receives the pre-clip global
 * L2 norm (the value the reference logs as grad.norm).  All four arenas 16-byte aligned.
 * bc1 = 1 - beta1^t, bc2 = 1 - beta2^t (the caller computes them in double like torch.optim.Adam); with
 * t2v_set_step_params installed, lr / bc1 / sqrt(bc2) come from the device record instead. */
int t2v_clip_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, uint64_t n,
                       float lr, float beta1, float beta2, float eps, float weight_decay,
                       float max_norm, float inv_world, float bc1, float bc2, float* partials,
                       float* norm_out, void* stream);
/* The same step, guarded (round 4): guard[0..guard_n) are error words of the step's cooperative / persistent kernels
 * (device memory).  If any of them is non-zero when the step runs, NOTHING is updated (parameters and both moment arenas keep
 * their values) and norm_out[0] reads NaN: the host finds the error at its next sync, switches the persistent kernels off
 * and runs the iteration again.  guard_n == 0: the plain step. */
int t2v_clip_adam_step_guarded(float* params, float* grads, float* exp_avg, float* exp_avg_sq, uint64_t n,
                               float lr, float beta1, float beta2, float eps, float weight_decay,
                               float max_norm, float inv_world, float bc1, float bc2, float* partials,
                               float* norm_out, const uint32_t* guard, int guard_n, void* stream);

/* ------------------------------------------------------------------ STFT -> mel front end
 * TacotronSTFT.mel_spectrogram (layers.py:75-92): reflect pad n_fft/2, periodic-Hann STFT
 * (stft.py:77-105), magnitude, mel filterbank, log(clamp(.,1e-5)) — batched, per-utterance lengths,
 * zero fill past T_b = n_samples[b]/hop + 1 (the collate pad value, data_utils.py:126).
 * Only n_fft = 1024, hop = 256, n_mel = 80 (T2V_ERR_DIMS otherwise).  Exactly one of wav_f32 /
 * wav_i16 is non-NULL; `scale` multiplies the samples (1/max_wav_value for int16 PCM).
 * Tables are built by the host mirror (layers.TacotronSTFT): window (1024), tw512 = exp(-2πik/512)
 * as (512,2), tw1024 = exp(-2πik/1024) as (513,2), mel filter rows in CSR form. */
int t2v_mel_frontend(const float* wav_f32, const int16_t* wav_i16, const int64_t* n_samples, int B,
                     int n_stride, float scale, int n_fft, int hop, int n_mel, const float* window,
                     const float* tw512, const float* tw1024, const int32_t* mel_start,
                     const int32_t* mel_len, const float* mel_w, int maxw, float* mel_out,
                     int t_stride, void* stream);

#ifdef __cplusplus
}
#endif
#endif
