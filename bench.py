#!/usr/bin/env python
"""bench.py — Tacotron2-VAE training-step throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one full training iteration of the reference loop (train.py:208-229): H2D of the
batch, forward, loss, backward, [gradient all-reduce over RCCL], clip + Adam — on the
BASELINE.json configs[1] workload: batch 6 per GPU, fixed shape T_in=84 symbols / T_out=400
frames (SURVEY.md §8(d) cfg-2 "train-fixed"), fp32, dropout on, synthetic data, random-init
weights from seed 1234.  Multi-GPU is weak scaling (6 utterances per rank).

The single JSON line also carries:
  roofline     — the dominant kernel (k_lstm_fwd256: both decoder LSTM gate GEMVs, weights streamed
                 every time step).  Algorithmic bytes per launch = fp32 weights of the two cells
                 (4096×1536 + 4096×2560)×4 B = 67.1 MB; duration = events on the launch stream
                 around a replay of exactly that kernel's T+1 launches.
  cpu_baseline — the oracle (oracle/t2v_oracle.py, a port of the reference's algorithm with
                 stock torch CPU ops) timed on this box's host cores on the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd'))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

T_IN, T_OUT = 84, 400
LSTM_WEIGHT_BYTES = (4096 * 1536 + 4096 * 2560) * 4
HBM_PEAK_GBS = 8000.0


def synthetic_batch(B, T_in, T_out, seed, lens_in=None, lens_out=None):
    """Collate-layout 7-tuple (reference data_utils.py:88-137) of synthetic data, CPU tensors."""
    g = torch.Generator().manual_seed(seed)
    lens_in = lens_in or [T_in] * B
    lens_out = lens_out or [T_out] * B
    text = torch.zeros(B, T_in, dtype=torch.long)
    mel = torch.zeros(B, 80, T_out)
    gate = torch.zeros(B, T_out)
    for i in range(B):
        text[i, :lens_in[i]] = torch.randint(2, 80, (lens_in[i],), generator=g)
        text[i, lens_in[i] - 1] = 1
        m = torch.randn(80, lens_out[i], generator=g) * 2.0 - 4.0
        mel[i, :, :lens_out[i]] = m.clamp(min=-11.5129, max=2.5)
        gate[i, lens_out[i] - 1:] = 1
    emotions = torch.zeros(B, 4, dtype=torch.long)
    emotions[torch.arange(B), torch.randint(0, 4, (B,), generator=g)] = 1
    return (text, torch.tensor(lens_in), mel, gate, torch.tensor(lens_out), torch.zeros(B, 1, dtype=torch.long),
            emotions)


def _cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for ln in f:
                if ln.lower().startswith('model name'):
                    return ln.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def cpu_baseline(steps=5, warmup=2, threads=None):
    """Oracle train step (fwd+loss+bwd+clip+Adam) on host cores, same workload, bounded sample (SURVEY 8(d):
    median of `steps` timed iterations after `warmup` warm-ups)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    if threads:
        torch.set_num_threads(threads)
    import t2v_oracle as O
    import hparams as HP
    import model as M
    hp = HP.create_hparams()
    torch.manual_seed(hp.seed)
    sd = {k: v.clone() for k, v in M.Tacotron2(hp).state_dict().items()}
    names = [k for k, v in sd.items() if v.dtype == torch.float32 and 'running_' not in k]
    for k in names:
        sd[k].requires_grad_(True)
    mstate = {k: (torch.zeros_like(sd[k]), torch.zeros_like(sd[k])) for k in names}
    text, lin, mel, gate, lout, _, _ = synthetic_batch(6, T_IN, T_OUT, 1234)
    g = torch.Generator().manual_seed(7)
    times = []
    for it in range(warmup + steps):
        t0 = time.perf_counter()
        drop = {}
        for i in range(3):
            drop['enc%d' % i] = torch.rand(6, 512, T_IN, generator=g) >= 0.5
        for i in range(5):
            drop['post%d' % i] = torch.rand(6, 512 if i < 4 else 80, T_OUT, generator=g) >= 0.5
        drop['prenet0'] = torch.rand(T_OUT + 1, 6, 256, generator=g) >= 0.5
        drop['prenet1'] = torch.rand(T_OUT + 1, 6, 256, generator=g) >= 0.5
        drop['lstm'] = [{k: torch.rand(6, 1024, generator=g) >= 0.1 for k in ('att_h', 'att_c', 'dec_h', 'dec_c')}
                        for _ in range(T_OUT)]
        out = O.tacotron2_forward(sd, text, lin, mel, lout, True, None, 0.1, 0.1, drop, 0.5, 0.5)
        loss = O.loss_forward(out, mel, gate, it, 'constant')[0]
        grads = torch.autograd.grad(loss, [sd[k] for k in names], allow_unused=True)
        live = [(k, gr) for k, gr in zip(names, grads) if gr is not None]
        clipped, _ = O.clip_grad_norm([gr for _, gr in live], 1.0)
        with torch.no_grad():
            for (k, _), gr in zip(live, clipped):
                p, m, v = O.adam_step(sd[k], gr, mstate[k][0], mstate[k][1], it + 1)
                sd[k].copy_(p)
                mstate[k] = (m, v)
        times.append(time.perf_counter() - t0)
    t = sorted(times[warmup:])[len(times[warmup:]) // 2]
    return {"value": round(6 * T_OUT / t, 2), "unit": "mel-frames/s", "cores": torch.get_num_threads(),
            "kind": "port", "s_per_it": round(t, 3), "host_cpu": _cpu_model(), "host_nproc": os.cpu_count(),
            "sample": "median of %d timed train steps after %d warm-up(s) of the same B=6,T_in=84,T_out=400 workload, "
                      "oracle/t2v_oracle.py with stock torch CPU fp32 ops, dropout on, %d torch threads"
                      % (steps, warmup, torch.get_num_threads())}


def _cpu_threads_leg(limit_s, threads=None):
    """ONE oracle step with `threads` host threads (None: every host core), in a child process with a time limit.  Default
    run (round 5): a 16-thread probe (a few seconds: shows that more threads than the fastest count do not help these M=6
    GEMVs).  The every-core figure SURVEY 8(d) names is opt-in (--cpu-all-cores): a GPU box that runs its containers under a
    CPU quota throttles 256 spinning OpenMP threads to a crawl — it burnt 75 s of every default run to report `null`."""
    import subprocess
    code = ("import json, os, sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import bench; "
            "r = bench.cpu_baseline(steps=1, warmup=0, threads=%s); print('ALLCORES ' + json.dumps(r))"
            % (ROOT, os.path.join(ROOT, 'tacotron2-vae_amd'), 'os.cpu_count()' if threads is None else str(int(threads))))
    t0 = time.perf_counter()
    try:
        p = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=limit_s)
        for line in p.stdout.split('\n'):
            if line.startswith('ALLCORES '):
                r = json.loads(line[9:])
                return {"value": r["value"], "cores": r["cores"], "s_per_it": r["s_per_it"],
                        "sample": "ONE timed step, no warm-up, %s (child process)"
                                  % ("every host core" if threads is None else "%d torch threads" % threads)}
        return {"value": None, "cores": threads or os.cpu_count(), "error": (p.stderr or p.stdout)[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "cores": threads or os.cpu_count(), "timed_out_after_s": round(time.perf_counter() - t0, 1),
                "note": "one step did not finish inside the limit (CPU quota of the container); "
                        "the figure with the fastest thread count is the one above"}


def decode_cpu_baseline(T_in=200, frames=100, reps=3, threads=8):
    """the CPU figure of cfg-4 beside the GPU one — the oracle's free-running decode loop
    (oracle/t2v_oracle.py:decoder_inference, reference model.py:428-464) on the same 200-symbol input shape, random-init
    weights of seed 1234, style = fc3(z) with z ~ N(0, I) seed 7, `threads` torch threads; median of `reps` runs of
    `frames` frames each after one warm-up run (bounded sample: ≈ 1 s of CPU work)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        import t2v_oracle as O
        import hparams as HP
        import model as M
        hp = HP.create_hparams()
        torch.manual_seed(hp.seed)
        sd = {k: v.detach().clone() for k, v in M.Tacotron2(hp).state_dict().items()}
        g = torch.Generator().manual_seed(1234)
        ids = torch.randint(2, 80, (1, T_in), generator=g)
        z = torch.randn(1, 32, generator=torch.Generator().manual_seed(7))
        with torch.no_grad():
            enc = O.encoder_forward(sd, ids, torch.tensor([T_in]), training=False)
            style = torch.nn.functional.linear(z, sd['vae_gst.fc3.weight'], sd['vae_gst.fc3.bias'])
            memory = enc + style.unsqueeze(1)
            times = []
            for i in range(reps + 1):
                t0 = time.perf_counter()
                O.decoder_inference(sd, memory, max_steps=frames, gate_threshold=1.0, stop_on_gate=False)
                times.append(time.perf_counter() - t0)
        t = sorted(times[1:])[reps // 2]
    finally:
        torch.set_num_threads(prev)
    return {"frames_per_s": round(frames / t, 1), "us_per_frame": round(1e6 * t / frames, 1), "cores": threads, "kind": "port",
            "host_cpu": _cpu_model(),
            "sample": "median of %d runs of %d free-running frames (after one warm-up run), B=1, %d symbols, "
                      "oracle/t2v_oracle.py:decoder_inference, stock torch CPU fp32 ops, %d torch threads"
                      % (reps, frames, T_in, threads)}


def decode_bench(model, T_in=200, steps=800, reps=5):
    """BASELINE.json configs[3]: B=1, 200-symbol utterance, exactly `steps` free-running decoder steps
    (gate ignored), style = fc3(z), z ~ N(0,I) seed 7.  Secondary figure: frames/s of the decode loop."""
    import model as M
    dec = model.decoder
    g = torch.Generator().manual_seed(1234)
    ids = torch.randint(2, 80, (1, T_in), generator=g).cuda()
    z = torch.randn(1, 32, generator=torch.Generator().manual_seed(7)).cuda()
    was_training = model.training
    model.eval()
    old_steps, old_thr = dec.max_decoder_steps, dec.gate_threshold
    dec.max_decoder_steps, dec.gate_threshold = steps, 1.0          # never stop early
    try:
        with torch.no_grad():
            enc = model.encoder.inference(model.transcript_embedding(ids).transpose(1, 2))
            memory = enc + model.vae_gst.fc3(z).unsqueeze(1)
            import contextlib, io
            res = {}
            with contextlib.redirect_stdout(io.StringIO()):
                for name, flag, runs in (("persistent", True, reps), ("launch_per_stage", False, 3)):
                    dec.inference(memory, chunk=steps, persistent=flag)                  # warm-up
                    torch.cuda.synchronize()
                    ts = []
                    for _ in range(runs):
                        t0 = time.perf_counter()
                        mel, _, _ = dec.inference(memory, chunk=steps, persistent=flag)
                        torch.cuda.synchronize()
                        ts.append(time.perf_counter() - t0)
                    res[name] = sorted(ts)[len(ts) // 2]
                    res[name + "_all"] = ts
            dt = res["persistent"]
    finally:
        dec.max_decoder_steps, dec.gate_threshold = old_steps, old_thr
        if was_training:
            model.train()
    return {"frames_per_s": round(steps / dt, 1), "us_per_frame": round(1e6 * dt / steps, 2), "B": 1, "T_in": T_in,
            "steps": steps, "mode": "one persistent launch, weights resident on chip (csrc/decoder_persist.hip)",
            "timing": "median of %d runs (wall clock around Decoder.inference, synchronised)" % reps,
            "us_per_frame_runs": [round(1e6 * t / steps, 2) for t in res["persistent_all"]],
            "launch_per_stage_us_per_frame": round(1e6 * res["launch_per_stage"] / steps, 2),
            "note": "decoder loop only (incl. session set-up: weight packing, memory_layer, one-time weight load); "
                    "encoder/postnet excluded"}


def frontend_bench(B=6, n_samples=102144, reps=20, big=True):
    """STFT->mel front end (k_mel_frontend): B utterances of 102 144 int16 samples (-> 400 frames
    each), events on the launch stream.  Algorithmic bytes per frame = 256 new samples x 2 B + 80 mels x 4 B = 832 B
    (SURVEY 8(d) counts 1 344 B/frame with fp32 samples)."""
    import layers
    import t2v_hip
    stft = layers.TacotronSTFT(1024, 256, 1024, 80, 16000, 0.0, 8000.0)
    g = torch.Generator().manual_seed(0)
    wav = (torch.clamp(0.1 * torch.randn(B, n_samples, generator=g), -1, 1) * 32767).to(torch.int16).cuda()
    n = torch.full((B,), n_samples, dtype=torch.int64)
    tables = stft._tables(wav.device)
    t2v_hip.mel_frontend(wav, n, tables, scale=1.0 / 32768.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        mel = t2v_hip.mel_frontend(wav, n, tables, scale=1.0 / 32768.0, t_stride=n_samples // 256 + 1)
    e1.record()
    torch.cuda.synchronize()
    us = 1000.0 * e0.elapsed_time(e1) / reps
    frames = B * (n_samples // 256 + 1)
    extra = {}
    if big:     # a batch that fills the chip (128 utterances = 51 328 wavefront-frames): the HBM-bound regime of SURVEY 8(d)
        r = frontend_bench(B=128, n_samples=n_samples, reps=10, big=False)
        extra = {"chip_filling_batch": {"B": 128, "us_per_launch": r["us_per_launch"], "frames_per_s": r["frames_per_s"],
                                        "achieved_GBps": r["achieved_GBps"], "frac_of_hbm_peak": round(r["achieved_GBps"] / HBM_PEAK_GBS, 4)}}
    return {**extra, "kernel": "k_mel_frontend", "frames_per_s": round(frames / (us * 1e-6), 1), "us_per_launch": round(us, 2),
            "frames_per_launch": frames, "bytes_per_frame": 832, "achieved_GBps": round(frames * 832 / (us * 1e-6) / 1e9, 2),
            "note": "int16 PCM in HBM -> (B,80,T) log-mel in HBM, one launch per batch; not part of the timed train step "
                    "(the synthetic batch carries mels, like the reference's collate output)"}


def _in_situ_durations():
    """per-kernel average durations of a traced steady-state step (rocprofv3 --kernel-trace over graph replays of the same
    command, committed under profiles/ by tools/round_profile.sh): the in-situ counterpart of the replay timings below"""
    for fn in ('r06_kernel_durations.json', 'r05_kernel_durations.json', 'r04_kernel_durations.json', 'r03_kernel_durations.json', 'r02_kernel_durations.json'):
        try:
            with open(os.path.join(ROOT, 'profiles', fn)) as f:
                return json.load(f), 'profiles/' + fn
        except Exception:
            pass
    return {}, None


def _profile_file(suffix):
    """the newest committed profiles/r0N_<suffix> (this round's evidence when it exists, else the previous round's)"""
    for r in ('r06', 'r05', 'r04', 'r03'):
        fn = os.path.join(ROOT, 'profiles', '%s_%s' % (r, suffix))
        if os.path.isfile(fn):
            return fn
    raise FileNotFoundError(suffix)


def _steady_state_rows():
    """{kernel name without 'void ' and arguments: (launches per step, ms per step)} of profiles/r0N_steady_state.txt"""
    import re
    rows = {}
    with open(_profile_file('steady_state.txt')) as f:
        for l in f.read().split('\n')[2:]:
            t = l.split()
            m = re.search(r'(k_\w+(?:<[^>]*>)?)', l)
            if m is None or len(t) < 4:
                continue
            try:
                c, ms = float(t[-3]), float(t[-2])
            except ValueError:
                continue
            k = m.group(1)
            rows[k] = (rows.get(k, (0, 0))[0] + c, rows.get(k, (0, 0))[1] + ms)
    return rows


def _whole_step_counters(ms_per_step):
    """what the step as a whole does to the two roofs, from the committed PMC passes of the same code (profiles/r04_pmc_*.json,
    tools/pmc_fetch_size.sh / pmc_mfma.sh: separate rocprofv3 --pmc runs) and the traced launches per step: bytes of 128-byte
    lines fetched per step (FETCH_SIZE x 2048, calibrated for polled loads in profiles/r04_fetch_calib.txt) over the kernels the
    PMC pass lists, and the time-weighted MFMA-busy fraction over the whole step"""
    res = {}
    try:
        rows = _steady_state_rows()

        def launches(key):          # "k_conv5_fwd<5>" matches that instantiation, "k_gemm_f32_big" every instantiation
            return sum(c for k, (c, _) in rows.items() if k == key or k.split('<')[0] == key)

        def ms(key):
            return sum(m for k, (_, m) in rows.items() if k == key or k.split('<')[0] == key)
    except Exception:
        return res
    try:
        fsrc = _profile_file('pmc_fetch_size.json')
        with open(fsrc) as f:
            fs = json.load(f)["kernels"]
        tot = sum(e["corrected_bytes_per_launch"] * launches(k) for k, e in fs.items())
        res["measured_fetch_bytes_per_step"] = int(tot)
        res["measured_fetch_frac_of_hbm_peak"] = round(tot / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        res["measured_fetch_kernels"] = sorted(fs)
        res["measured_fetch_source"] = "%s x launches/step of %s" % (os.path.relpath(fsrc, ROOT), os.path.relpath(_profile_file('steady_state.txt'), ROOT))
    except Exception:
        pass
    try:
        msrc = _profile_file('pmc_mfma.json')
        with open(msrc) as f:
            mf = json.load(f)["kernels"]
        busy = sum(e["mfma_busy_frac"] * ms(k) for k, e in mf.items() if "mfma_busy_frac" in e)
        res["mfma_busy_frac_over_step"] = round(busy / ms_per_step, 4)
        # the dense kernels of the step against the MFMA roof of the dtype their MFMAs run in (x3 products: six bf16 MFMAs per fp32
        # product block — `fp32_equiv_tflops` is what the caller gets)
        res["mfma_kernels"] = [
            {"kernel": k, "mfma_dtype": e.get("mfma_dtype"), "mfma_busy_frac": e.get("mfma_busy_frac"), "mfma_tflops_at_2.4GHz": e.get("tflops_at_2.4GHz"),
             "peak_tflops": 2500.0 if e.get("mfma_dtype") == "bf16" else 157.0,
             **({"fp32_equiv_tflops": round(e.get("tflops_at_2.4GHz", 0.0) / 6.0, 1)} if ('x3p' in k or 'conv5_x3' in k) and e.get("mfma_dtype") == "bf16" else {}),
             "ms_per_step": round(ms(k), 3), "dispatches_in_profile": e.get("dispatches")}
            for k, e in sorted(mf.items(), key=lambda kv: -ms(kv[0])) if "mfma_busy_frac" in e and ms(k) > 0]
        res["mfma_busy_source"] = "%s x ms/step of %s" % (os.path.relpath(msrc, ROOT), os.path.relpath(_profile_file('steady_state.txt'), ROOT))
    except Exception:
        pass
    return res


def roofline_table(B, T_in, T, reps=3):
    """Event-timed replays of the per-time-step kernels of the decoder recurrence on the arena of the last step (events
    on the launch stream = torch's current stream).  Algorithmic bytes: what one launch must move at least.
    `avg_launch_us` is the duration of back-to-back replays of exactly that kernel (warm caches, no neighbours);
    `in_situ_us` / `frac_in_situ` come from the committed rocprofv3 trace of the whole step (profiles/), where the kernel
    runs between its real neighbours — the number to recompute `frac` from."""
    import t2v_hip
    f4 = 4
    lstm_bytes = LSTM_WEIGHT_BYTES
    attn_fwd_bytes = f4 * (B * 256 * 128 + B * T_in * 128 + B * T_in * 512 + 2 * B * T_in + 8192        # qp, pm, memory, al/acum, W_comb
                           + B * T_in * 128 + 2 * B * T_in + B * 512)                                    # S, al/acum out, ctx
    attn_bwd_bytes = f4 * (2 * B * T_in * 128 + B * T_in * 512 + B * (1536 + 2560 + 1536)                # S r/w, memory, dHC/YD/YA
                           + 2 * B * 4096 + 4 * B * 1024 + 2 * B * 4096 + 1024 * 128 + 8192)             # GA/GD, cells, DGA/DGD, W_q^T, W_comb
    # the persistent forward: every weight once per PASS, the Prenet term in, every saved activation out
    persist_bytes = (LSTM_WEIGHT_BYTES + f4 * (128 * 1024 + 16384 + B * T_in * 640)                       # weights, W_q, W_comb, memory + pm
                     + f4 * T * B * 4096                                                                 # gpre
                     + f4 * ((T + 2) * B * 2560 + 2 * (T + 1) * B * 1024 + 2 * T * B * 4096              # XS, CA/CD, GA/GD
                             + 2 * (T + 1) * B * T_in + T * B * T_in * 128))                              # AL/ACUM, S
    persistent = t2v_hip.DecoderCore.last_mode == 'persistent'
    legs = []
    if persistent:
        legs.append((t2v_hip.DecoderCore.last_kernel or "k_dec_train_persist", lambda _m: t2v_hip.replay_persistent_forward(), 0, persist_bytes, 1))
    else:
        legs.append(("k_lstm_fwd256", t2v_hip.replay_fwd_kernels, 1, lstm_bytes, T + 1))
        legs.append(("k_attn_fwd", t2v_hip.replay_fwd_kernels, 2, attn_fwd_bytes, T))
    # the persistent reverse pass: every transposed weight column once per PASS, the saved activations in, the gate
    # gradients / dpre / dctx / dq out
    persist_bwd_bytes = (LSTM_WEIGHT_BYTES + f4 * (128 * 1024 + 16384 + B * T_in * 512)
                         + f4 * (T * B * 1536 + (T + 2) * B * 2560 + 2 * (T + 1) * B * 1024 + 2 * T * B * 4096   # dHC, XS, CA/CD, GA/GD
                                 + (T + 1) * B * T_in + T * B * T_in * 128)                                      # AL, S in
                         + f4 * (2 * T * B * 4096 + T * B * 512 + T * B * T_in * 128 + T * B * 8 * 128))        # DGA/DGD, DCTX, dpre, dq
    if t2v_hip.DecoderCore.last_bwd_mode == 'persistent':
        legs.append((t2v_hip.DecoderCore.last_bwd_kernel or "k_achain_bwd", lambda _m: t2v_hip.replay_persistent_backward(), 0, persist_bwd_bytes, 1))
    else:
        legs.append(("k_lstm_bwd256", t2v_hip.replay_bwd_kernels, 1, lstm_bytes, T))
        legs.append(("k_attn_cell_bwd", t2v_hip.replay_bwd_kernels, 2, attn_bwd_bytes, T + 1))
    situ, situ_src = _in_situ_durations()
    rows = []
    for name, fn, mask, nbytes, per_step in legs:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(mask)
        torch.cuda.synchronize()
        ev0.record()
        n = 0
        for _ in range(reps):
            n += fn(mask)
        ev1.record()
        torch.cuda.synchronize()
        us = 1000.0 * ev0.elapsed_time(ev1) / n
        gbs = nbytes / (us * 1e-6) / 1e9
        row = {"kernel": name, "bound": "hbm", "algorithmic_bytes_per_launch": int(nbytes),
               "avg_launch_us": round(us, 3), "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": round(gbs / HBM_PEAK_GBS, 4), "launches_per_step": per_step, "launches_timed": n,
               "timing": "HIP events around back-to-back replays of this kernel alone"}
        if name in situ:
            row["in_situ_us"] = situ[name]
            row["frac_in_situ"] = round(nbytes / (situ[name] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
            row["in_situ_source"] = situ_src
        if name == "k_achain_bwd":
            row["us_per_time_step"] = round(us / T, 3)
            row["note"] = ("ONE launch for the whole reverse pass (%d time steps): all-gather of (dc, dh) -> transposed weight columns "
                           "in registers -> attention backward on position-split workgroups -> cell backward.  A chain of dependent "
                           "hand-offs between CUs: HBM bandwidth is not what bounds it (the `frac` against 8 TB/s is reported because "
                           "the contract asks for it)" % T)
            # what does bound it (tools/micro/hop_latency.hip, tools/dbg/persist_bwd_prof.py): a CU pulls ~11 B/cycle
            # from beyond its L2, a word crosses the chip in ~0.45 us, and a step needs three dependent hand-offs, a 48 KB row per
            # attention_rnn workgroup on the chain and ~3.8 us of dependent arithmetic (context GEMV, attention slice, cell)
            # round 4 (profiles/r04_bwd_persist_timeline.txt): FOUR dependent hand-offs per step — (dc, dh) row -> attention_rnn
            # workgroups, context gradient -> attention slices, partial dq -> slice 0 of the item, summed dq -> attention_rnn
            # workgroups — and, between them, the context-column GEMV (1.2 us), the attention slice (1.2 us) and the cell (1.0 us)
            # round 6 (profiles/r06_bwd_persist_timeline.txt): the model on BOTH hop costs.  A word crosses the chip in
            # 0.45 us (profiles/r03_hop_latency.txt: two workgroups, nothing else running) — the floor of an ideal hand-off; what a
            # dependent look costs INSIDE this kernel, with >= 128 workgroups polling and the factor DMA in the same memory pipe, is
            # ~1 us (round 5; the r06 timeline: context gradient published -> seen by the attention slices 1.3-1.7 us,
            # dq published -> gathered by the attention_rnn workgroups through slice 0 2.2 us = two hops).  Four dependent hand-offs,
            # the 48 KB (dc, dh) row at a CU's 11 B/cycle, 3.4 us of dependent arithmetic (context-column GEMV 1.2, attention slice
            # 1.2, cell 1.0)
            fetch_us = 48 * 1024 / 11.0 / 2400.0
            floor_us = 4 * 0.45 + fetch_us + 3.4
            floor_loaded_us = 4 * 1.0 + fetch_us + 3.4
            row["latency_model"] = {"hand_offs_per_step": 4, "hand_off_us_idle_chip": 0.45, "hand_off_us_under_load": 1.0,
                                    "row_bytes_on_chain_per_cu": 48 * 1024, "cu_fetch_bytes_per_cycle": 11, "dependent_compute_us": 3.4,
                                    "floor_us_per_step": round(floor_us, 2), "floor_us_per_step_loaded_hops": round(floor_loaded_us, 2),
                                    "achieved_us_per_step": round(us / T, 2), "frac_of_floor": round(floor_us / (us / T), 3),
                                    "frac_of_loaded_hop_floor": round(floor_loaded_us / (us / T), 3),
                                    "source": "profiles/r06_bwd_persist_timeline.txt (this code), profiles/r03_hop_latency.txt (idle-chip hop)"}
        if name == "k_bwd_persist16":
            row["us_per_time_step"] = round(us / T, 3)
            row["note"] = ("bf16_run, B <= 16: ONE launch for the whole reverse pass (%d time steps) — Wcat^T as register-resident bf16 MFMA "
                           "tiles cut 128 columns x 1024 rows (48 + 80 workgroups publish partial column sums), 16 + 16 cell workgroups, "
                           "B*S attention-backward workgroups; per step a chain of four hand-offs (gate gradients -> partial sums -> context "
                           "gradient -> dq -> cells), latency-bound: the `frac` against 8 TB/s is reported because the contract asks for it"
                           % T)
        if name == "k_dec_train_persist16":
            row["us_per_time_step"] = round(us / T, 3)
            row["note"] = ("bf16_run, B <= 16: ONE launch for all %d time steps — the LSTM weights as register-resident bf16 MFMA tiles "
                           "(8 hidden units of both cells per workgroup, the batch is the N dimension of v_mfma_f32_16x16x32_bf16), the "
                           "state rows exchanged in MFMA-operand order and polled straight into registers; a chain of dependent hand-offs "
                           "(attention_rnn -> attention -> attention_rnn), latency-bound by construction" % T)
        if name == "k_dec_train_persist":
            row["us_per_time_step"] = round(us / T, 3)
            row["note"] = ("ONE launch for all %d time steps: a chain of dependent hand-offs between CUs (attention_rnn -> "
                           "attention -> attention_rnn), latency-bound by construction; its 67 MB of LSTM weights are read "
                           "once per pass into registers instead of once per step" % T)
        rows.append(row)
    rows.sort(key=lambda r: -r["avg_launch_us"] * r["launches_per_step"])
    return rows


def run_workload(args, world, rank, bf16, koemo, steps, warmup, graph, eager_steps=0):
    """one timed configuration; returns (engine, result dict)"""
    import hparams as HP
    import t2v_hip
    import train as TR
    bpg = 16 if bf16 else 6
    hp = HP.create_hparams("batch_size=%d,anneal_function=constant%s%s" % (
        bpg, ",distributed_run=True" if world > 1 else "", ",bf16_run=True" if bf16 else ""))
    torch.manual_seed(hp.seed)
    torch.cuda.manual_seed(hp.seed)
    engine = TR.TrainEngine(hp, world_size=world, graph=graph, force_dist=bool(getattr(args, 'force_dist', False)))
    koemo_in, koemo_out = [84, 80, 71, 66, 50, 37], [400, 380, 350, 300, 260, 200]
    if koemo:
        batch = synthetic_batch(bpg, T_IN, T_OUT, 1234 + rank, lens_in=koemo_in, lens_out=koemo_out)
    else:
        batch = synthetic_batch(bpg, T_IN, T_OUT, 1234 + rank)
    batch = tuple(t.pin_memory() for t in batch)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    it = 0
    startup = 0
    with engine.stream_context():
        if engine.use_graph:
            # graph priming is start-up cost like building the model: the shape is captured the third time it is seen,
            # so these extra untimed steps make sure neither the warm-up nor the timed region contains the capture
            startup = engine.GRAPH_AFTER + 1 + args.settle
            for _ in range(startup):
                engine.step(batch, it)
                it += 1
        for _ in range(warmup):
            engine.step(batch, it)
            it += 1
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = engine.step(batch, it)[0]
            it += 1
        sync()
        elapsed = time.perf_counter() - t0
        # a rank's own clock up to ITS last step's completion would need a sync before the barrier; what a straggler shows
        # in is the per-rank time to issue + finish its steps, taken around the same region with a device sync only
        eager_ms = None
        if engine.use_graph and eager_steps > 0:
            # the same step issued eagerly (one host launch per kernel) by the same engine, in the same run: a replayed graph
            # whose branches the executor decided to serialise shows up as graph >> eager
            engine.use_graph = False
            try:
                for _ in range(2):
                    engine.step(batch, it)
                    it += 1
                sync()
                t1 = time.perf_counter()        # (host clock from a drained queue: includes the issue time of the first step)
                for _ in range(eager_steps):
                    engine.step(batch, it)
                    it += 1
                sync()
                eager_ms = 1000.0 * (time.perf_counter() - t1) / eager_steps
            finally:
                engine.use_graph = True
    rank_ms = None
    if world > 1:
        mine = torch.tensor([elapsed], device='cpu' if dist.get_backend() == 'gloo' else 'cuda', dtype=torch.float64)
        allt = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allt, mine)
        per = [1000.0 * float(t.item()) / steps for t in allt]
        rank_ms = {"min": round(min(per), 3), "max": round(max(per), 3), "per_rank": [round(x, 3) for x in per]}
        elapsed = max(float(t.item()) for t in allt)
    final_loss = float(loss.item())
    t2v_hip.check_async_errors()     # any bounded-spin timeout inside the timed steps invalidates the run
    frames = (sum(koemo_out) if koemo else bpg * T_OUT) * world      # koemo: valid (unpadded) frames, like the metric
    ms = 1000.0 * elapsed / steps
    res = {"value": round(frames / (elapsed / steps), 1), "ms_per_step": round(ms, 3), "frames_per_step": frames,
           "final_loss": round(final_loss, 5), "step_mode": ("hip-graph replay of forward + backward, then one eager all-reduce and the fused clip + Adam"
                         if getattr(engine, 'graph_ddp', False) else "hip-graph replay") if engine.use_graph else "eager launches",
           "startup_steps": startup, "batch_per_gpu": bpg,
           "decoder_forward": t2v_hip.DecoderCore.last_mode, "decoder_backward": t2v_hip.DecoderCore.last_bwd_mode}
    if getattr(engine, 'graph_fallbacks', 0):
        # the engine's replay watchdog (train.TrainEngine._probe_end) found the captured graph slower than the eager step and
        # dropped it: the timed steps above were eager launches
        rp, eg = list(engine._no_graph.values())[-1]
        res["step_mode"] = "eager launches (replay watchdog: the captured graph replayed in %.2f ms, the eager step took %.2f ms)" % (rp, eg)
        res["graph_watchdog"] = {"fallbacks": int(engine.graph_fallbacks), "replay_ms": round(rp, 3), "eager_ms": round(eg, 3)}
    elif engine.use_graph and getattr(engine, 'graph_watchdog', False):
        res["graph_watchdog"] = {"fallbacks": 0}
    if eager_ms is not None:
        res["graph_vs_eager_ms"] = {"graph": round(ms, 3), "eager": round(eager_ms, 3), "eager_steps": eager_steps,
                                    "graph_replay_serialised": bool(ms > eager_ms + 0.5),
                                    "note": "the same engine, the same step, issued eagerly right after the timed replays: a "
                                            "graph whose branches were serialised by the executor would read graph >> eager"}
    if rank_ms is not None:
        res["ms_per_step_ranks"] = rank_ms
        # what every rank's engine ended up running: the replay watchdog and the persistent-kernel time-out path
        # are rank-local decisions — one rank that fell back to eager launches makes all others wait at the all-reduce, and the line
        # has to say so
        mine = {"rank": rank, "step_mode": res["step_mode"].split(' (')[0], "graph_watchdog": res.get("graph_watchdog"),
                "decoder_forward": res["decoder_forward"], "decoder_backward": res["decoder_backward"],
                "recoveries": int(getattr(engine, 'recoveries', 0))}
        allm = [None] * world
        dist.all_gather_object(allm, mine)
        res["ranks"] = allm
        keyf = lambda m: (m["step_mode"], m["decoder_forward"], m["decoder_backward"], (m["graph_watchdog"] or {}).get("fallbacks", 0), m["recoveries"])
        res["ranks_disagree"] = len({keyf(m) for m in allm}) > 1
        if res["ranks_disagree"] and rank == 0:
            print("bench.py: WARNING — the ranks did not all run the same kind of step (replay watchdog fall-back or persistent-kernel "
                  "recovery on some of them): the slowest form sets the pace at the all-reduce: %s" % json.dumps(allm), file=sys.stderr, flush=True)
    if engine.allreduce is not None:
        res["allreduce_exposed_ms"] = round(engine.allreduce.exposed_ms(), 3)
        res["allreduce_buckets"] = [(b[0], 4 * (b[2] - b[1])) for b in engine.allreduce.buckets]
    return engine, res


WORKLOADS = {
    "headline": "configs[1]: Tacotron2-VAE fp32 train step (fwd+loss+bwd+clip+Adam), B=6/GPU fixed shape T_in=84 T_out=400, "
                "dropout on, random-init seed 1234",
    "koemo": "configs[1], koemo length profile: fp32 train step, B=6/GPU ragged (T_in,T_out) = (84,400),(80,380),(71,350),"
             "(66,300),(50,260),(37,200), valid frames counted, dropout on",
    "bf16": "configs[4]: bf16_run train step (bf16 MFMA wide Conv1d fwd/dx + time-batched linears + LSTM dW GEMMs; fp32 "
            "master/BN/recurrence), B=16/GPU fixed shape T_in=84 T_out=400, dropout on, random-init seed 1234",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-steps', type=int, default=5)
    ap.add_argument('--cpu-warmup', type=int, default=2)
    ap.add_argument('--cpu-threads', type=int, default=8,
                    help='torch CPU threads for the baseline leg (the M=6 GEMVs of this model stop scaling\n'
                         'around 8 threads on this EPYC host: 8 → 3.7 s/it, 16 → 4.2, 32 → 7.4, all cores ≈ 40)')
    ap.add_argument('--cpu-all-cores-timeout', type=float, default=75.0)
    ap.add_argument('--cpu-all-cores', action='store_true',
                    help='also time ONE oracle step with every host core (opt-in: ≈40 s on an unthrottled 128-core host, cut off\n'
                         'after --cpu-all-cores-timeout on the quota-limited GPU boxes)')
    ap.add_argument('--cpu-probe-threads', type=int, default=16,
                    help='second CPU figure of the default run: ONE oracle step with this many threads (0 = skip)')
    ap.add_argument('--eager-steps', type=int, default=30,
                    help='eager steps timed next to the graph replays (graph_vs_eager_ms; 0 = skip).  30: the first eager steps '
                         'after a sync refill the launch queue — with 5 steps that start-up read as +0.4 ms per step')
    ap.add_argument('--no-decode', action='store_true')
    ap.add_argument('--no-secondary', action='store_true',
                    help='skip the secondary workloads (koemo length profile, bf16 B=16) and the front-end leg')
    ap.add_argument('--settle', type=int, default=40,
                    help='extra untimed start-up steps after the graph capture (reported as config.startup_steps)')
    ap.add_argument('--no-graph', action='store_true',
                    help='run the step eagerly (one host launch per kernel) instead of replaying the captured HIP graph')
    ap.add_argument('--launch', action='store_true',
                    help='self-launch under torch.distributed.run even for --gpus 1 (exercises the multi-GPU entry on one GPU)')
    ap.add_argument('--force-dist', action='store_true',
                    help='initialise RCCL and run the bucketed gradient all-reduce even in a 1-rank world (tests)')
    ap.add_argument('--koemo', action='store_true', help='make the koemo length profile the headline workload')
    ap.add_argument('--bf16', action='store_true', help='make BASELINE configs[4] (bf16_run, B=16 per GPU) the headline workload')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if (args.gpus > 1 or args.launch) and 'WORLD_SIZE' not in os.environ:
        # self-launch (replaces the reference's multiproc.py:1-23): one process per GPU under torch.distributed.run,
        # rendezvous on 127.0.0.1 (the container hostname may not resolve)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    # T2V_BENCH_SHARE_GPU=1 + T2V_BENCH_BACKEND=gloo (tests only): N ranks of this entry on ONE GPU — RCCL refuses two ranks on one
    # device, so the plumbing of the N > 1 line (self-launch, rendezvous, barrier + max over ranks, per-rank keys) is exercised over
    # gloo with device tensors staged through the host (distributed.all_reduce_sum); production is one GPU per rank over RCCL
    share = os.environ.get('T2V_BENCH_SHARE_GPU', '0') == '1'
    torch.cuda.set_device(0 if share else local)
    if world > 1 or args.force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group(os.environ.get('T2V_BENCH_BACKEND', 'nccl'), init_method='env://', world_size=world, rank=rank)

    import t2v_hip
    t2v_hip.load_library()
    t2v_hip.DecoderCore.keep_last = True       # the roofline leg replays the last step's kernels on its arena
    kind = 'bf16' if args.bf16 else ('koemo' if args.koemo else 'headline')
    engine, res = run_workload(args, world, rank, args.bf16, args.koemo and not args.bf16, args.steps, args.warmup,
                               not args.no_graph, eager_steps=args.eager_steps)
    bpg = res["batch_per_gpu"]
    out = {
        "metric": "mel-frames/s (train step, batch=%d, 80-mel)" % bpg, "value": res["value"],
        "unit": "mel-frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.bf16 else "f32",
        "data": "synthetic",
        "config": {"workload": WORKLOADS[kind], "step_mode": res["step_mode"], "decoder_forward": res["decoder_forward"], "decoder_backward": res["decoder_backward"],
                   "startup_steps": res["startup_steps"],
                   "global_batch": bpg * world, "frames_per_step": res["frames_per_step"],
                   "parallelism": "dp%d" % world,
                   "f32_dense_products": ("fp32-MFMA only (T2V_F32_GEMM=native)" if not t2v_hip.set_f32_gemm_mode(None) else
                                          "large GEMMs (>= 64 tiles of 128x128) as six bf16 MFMAs on exactly 3-way-split fp32 operands "
                                          "(x3: fp32-class error, tests/test_gemm_gpu.py::test_x3_gemm_is_fp32_class), everything else "
                                          "v_mfma_f32_*_f32")},
        "final_loss": res["final_loss"],
    }
    if "graph_vs_eager_ms" in res:
        out["graph_vs_eager_ms"] = res["graph_vs_eager_ms"]
    if "graph_watchdog" in res:
        out["graph_watchdog"] = res["graph_watchdog"]
    if "ms_per_step_ranks" in res:      # per-rank step time next to the max the value is computed from: a straggler is visible
        out["ms_per_step_ranks"] = res["ms_per_step_ranks"]
        out["ranks"] = res["ranks"]
        out["ranks_disagree"] = res["ranks_disagree"]
    if dist.is_initialized():
        out["rccl_ranks"] = dist.get_world_size()
        if "allreduce_exposed_ms" in res:
            out["allreduce_exposed_ms"] = res["allreduce_exposed_ms"]
            out["allreduce_buckets_bytes"] = res["allreduce_buckets"]

    if rank == 0:
        # ---- roofline leg: the four per-time-step kernels of the decoder recurrence (87 % of the GPU time of a step)
        rows = roofline_table(bpg, T_IN, T_OUT)
        top = rows[0]                   # the kernel with the most time per step
        traffic, tsrc = None, None      # HBM bytes per launch from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE, corrected)
        for fn in ('r06_pmc_fetch_size.json', 'r05_pmc_fetch_size.json', 'r04_pmc_fetch_size.json', 'r03_pmc_fetch_size.json', 'r02_pmc_fetch_size.json', 'r01_pmc_fetch_size.json'):
            try:
                with open(os.path.join(ROOT, 'profiles', fn)) as f:
                    traffic = json.load(f)["kernels"][top["kernel"]]["corrected_bytes_per_launch"]
                tsrc = "profiles/%s (separate rocprofv3 --pmc pass, tools/pmc_fetch_size.sh; not measured in this run)" % fn
                break
            except Exception:
                pass
        e2e_bytes = 58.9e9   # compulsory bytes of one cfg-2 iteration
        rec_us = sum(r["avg_launch_us"] * r["launches_per_step"] for r in rows) / T_OUT
        stream_rows = [r for r in rows if r["kernel"].startswith("k_lstm_")]
        out["roofline"] = {"kernel": top["kernel"], "bound": "hbm", "achieved": top["achieved"], "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": top["frac"], "traffic": traffic, "traffic_source": tsrc,
                           "avg_launch_us": top["avg_launch_us"],
                           "algorithmic_bytes_per_launch": top["algorithmic_bytes_per_launch"],
                           "launches_timed": top["launches_timed"],
                           "frac_in_situ": top.get("frac_in_situ"), "in_situ_us": top.get("in_situ_us"),
                           "in_situ_source": top.get("in_situ_source"),
                           "note": top.get("note") if top.get("note") else
                                   "the 67 MB weight stream of a launch is re-read every time step and is served by the "
                                   "256 MiB Infinity Cache, not by HBM proper; 8 TB/s is the HBM3E peak the guide prices against",
                           "latency_model": top.get("latency_model"), "us_per_time_step": top.get("us_per_time_step"),
                           "kernels": rows,
                           "forward_mode": t2v_hip.DecoderCore.last_mode, "backward_mode": t2v_hip.DecoderCore.last_bwd_mode,
                           "recurrence_us_per_time_step": round(rec_us, 2),
                           # NOT a utilisation: SURVEY 8(d) prices an iteration at 58.9 GB under the launch-per-step formulation
                           # (67 MB of LSTM weights re-read 801 times); the persistent kernels keep the weights in registers and
                           # do not move those bytes at all — this is the step's speed relative to that streaming model
                           "speed_vs_survey_streaming_model": round(e2e_bytes / (out["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                           if kind == 'headline' else None}
        out["roofline"].update(_whole_step_counters(out["ms_per_step"]))
        if not args.no_decode:
            out["decode"] = decode_bench(engine.model)
            dec_bytes = 72.86e6      # 72.35 MB of recurrent weights + 0.51 MB of memory / processed memory per frame
            dgbs = dec_bytes / (out["decode"]["us_per_frame"] * 1e-6) / 1e9
            out["roofline"]["kernels"].append({
                "kernel": "k_decode_persist", "bound": "hbm", "algorithmic_bytes_per_launch": int(dec_bytes * out["decode"]["steps"]),
                "avg_launch_us": round(out["decode"]["us_per_frame"] * out["decode"]["steps"], 1), "achieved": round(dgbs, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(dgbs / HBM_PEAK_GBS, 4), "launches_per_step": 0,
                "us_per_frame": out["decode"]["us_per_frame"],
                "note": "free-running decode, one launch per utterance (not part of the train step): priced against what a "
                        "per-frame weight stream would have to move (SURVEY 8(d): 9.1 us per frame at 8 TB/s); the kernel itself "
                        "keeps the weights in registers and is bound by the chain of hand-offs between CUs",
                "timing": "wall clock around Decoder.inference, %d frames" % out["decode"]["steps"]})
        if world == 1 and not args.no_secondary and not args.force_dist:
            out["frontend"] = frontend_bench()
            fe = out["frontend"]
            for tag, us, frames in (("B=6", fe["us_per_launch"], fe["frames_per_launch"]),
                                    ("B=128", fe["chip_filling_batch"]["us_per_launch"], 128 * (102144 // 256 + 1))):
                gbs = frames * 832 / (us * 1e-6) / 1e9
                out["roofline"]["kernels"].append({
                    "kernel": "k_mel_frontend (%s)" % tag, "bound": "hbm", "algorithmic_bytes_per_launch": frames * 832,
                    "avg_launch_us": us, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 4), "launches_per_step": 0,
                    "note": "STFT->mel front end, one launch per batch (not part of the timed train step): int16 PCM in, log-mel "
                            "out = 832 B per frame; the FFT butterflies make it compute/latency-bound long before HBM",
                    "timing": "HIP events around back-to-back launches"})
            sec = {}
            del engine
            t2v_hip.DecoderCore.keep_last = False
            t2v_hip.DecoderCore.last_call = t2v_hip.DecoderCore.last_bwd = t2v_hip.DecoderCore.last_persist = t2v_hip.DecoderCore.last_bwd_persist = None
            torch.cuda.empty_cache()
            for name, (b16, ko) in (("koemo", (False, True)), ("bf16", (True, False))):
                if name == kind:
                    continue
                t2v_hip.release_step_params()
                _, r2 = run_workload(args, world, rank, b16, ko, max(10, args.steps // 2), 3, not args.no_graph)
                r2["workload"] = WORKLOADS[name]
                r2["unit"] = "mel-frames/s"
                if name == 'bf16':
                    try:        # its own roofline rows: MFMA counters of the bf16 step, separate PMC pass
                        bsrc = _profile_file('pmc_mfma_bf16.json')
                        with open(bsrc) as f:
                            mb = json.load(f)["kernels"]
                        r2["roofline_kernels"] = [
                            {"kernel": k, "bound": "mfma", "achieved": e.get("tflops_at_2.4GHz"), "unit": "TFLOP/s",
                             "peak": 2500.0 if e.get("mfma_dtype", "bf16" if 'bf16' in k else "f32") == "bf16" else 157.0,
                             "mfma_dtype": e.get("mfma_dtype"), "mfma_busy_frac": e.get("mfma_busy_frac"),
                             "dispatches_in_profile": e.get("dispatches"), "source": os.path.relpath(bsrc, ROOT)}
                            for k, e in mb.items() if "mfma_busy_frac" in e]
                    except Exception:
                        pass
                sec[name] = r2
                t2v_hip.set_bf16(False)
            out["secondary"] = sec
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(steps=args.cpu_steps, warmup=args.cpu_warmup, threads=args.cpu_threads)
            out["speedup_vs_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
            if args.cpu_probe_threads and args.cpu_probe_threads != args.cpu_threads:
                out["cpu_baseline"]["probe_%d_threads" % args.cpu_probe_threads] = _cpu_threads_leg(30.0, args.cpu_probe_threads)
            if args.cpu_all_cores:              # SURVEY 8(d) says "all cores": opt-in, see _cpu_threads_leg
                out["cpu_baseline"]["all_cores"] = _cpu_threads_leg(args.cpu_all_cores_timeout)
            if "decode" in out:                 # cfg-4 gets its CPU figure beside it as well
                out["decode"]["cpu_baseline"] = decode_cpu_baseline(threads=args.cpu_threads)
                out["decode"]["speedup_vs_cpu"] = round(out["decode"]["frames_per_s"] / out["decode"]["cpu_baseline"]["frames_per_s"], 1)
        # the secondary headline figures as scalars at the FRONT of the line (the driver stores a tail of stdout,
        # and configs[4]'s number used to sit in the part that was cut off)
        front = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step") if k in out}
        sec = out.get("secondary", {})
        if "bf16" in sec:
            front["bf16_ms_per_step"] = sec["bf16"]["ms_per_step"]
            front["bf16_frames_per_s"] = sec["bf16"]["value"]
        if "koemo" in sec:
            front["koemo_ms_per_step"] = sec["koemo"]["ms_per_step"]
        if "decode" in out:
            front["decode_us_per_frame"] = out["decode"]["us_per_frame"]
        front.update({k: v for k, v in out.items() if k not in front})
        print(json.dumps(front))
    if dist.is_initialized():
        # orderly teardown of a rank: captured graphs and their arenas go first, then the communicator; the process then
        # leaves without running the remaining library destructors (HIP graph / RCCL teardown order at interpreter exit
        # aborted one run in a few with exit code -6 AFTER the result line had been printed)
        import gc
        torch.cuda.synchronize()
        dist.barrier()
        t2v_hip.DecoderCore.last_call = t2v_hip.DecoderCore.last_bwd = t2v_hip.DecoderCore.last_persist = t2v_hip.DecoderCore.last_bwd_persist = None
        del engine
        gc.collect()
        torch.cuda.synchronize()
        dist.destroy_process_group()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == '__main__':
    main()
