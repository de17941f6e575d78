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

B_PER_GPU, T_IN, T_OUT = 6, 84, 400
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


def cpu_baseline(steps=2, warmup=1, threads=None):
    """Oracle train step (fwd+loss+bwd+clip+Adam) on host cores, same workload, bounded sample."""
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
    text, lin, mel, gate, lout, _, _ = synthetic_batch(B_PER_GPU, T_IN, T_OUT, 1234)
    g = torch.Generator().manual_seed(7)
    times = []
    for it in range(warmup + steps):
        t0 = time.perf_counter()
        drop = {}
        for i in range(3):
            drop['enc%d' % i] = torch.rand(B_PER_GPU, 512, T_IN, generator=g) >= 0.5
        for i in range(5):
            drop['post%d' % i] = torch.rand(B_PER_GPU, 512 if i < 4 else 80, T_OUT, generator=g) >= 0.5
        drop['prenet0'] = torch.rand(T_OUT + 1, B_PER_GPU, 256, generator=g) >= 0.5
        drop['prenet1'] = torch.rand(T_OUT + 1, B_PER_GPU, 256, generator=g) >= 0.5
        drop['lstm'] = [{k: torch.rand(B_PER_GPU, 1024, generator=g) >= 0.1 for k in ('att_h', 'att_c', 'dec_h', 'dec_c')}
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
    return {"value": round(B_PER_GPU * T_OUT / t, 2), "unit": "mel-frames/s", "cores": torch.get_num_threads(),
            "kind": "port", "s_per_it": round(t, 3),
            "sample": "%d timed train steps (after %d warm-up) of the same B=6,T_in=84,T_out=400 workload, "
                      "oracle/t2v_oracle.py with stock torch CPU fp32 ops, dropout on" % (steps, warmup)}


def decode_bench(model, T_in=200, steps=800):
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
            with contextlib.redirect_stdout(io.StringIO()):
                dec.inference(memory, chunk=steps)                  # warm-up
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                mel, _, _ = dec.inference(memory, chunk=steps)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
    finally:
        dec.max_decoder_steps, dec.gate_threshold = old_steps, old_thr
        if was_training:
            model.train()
    return {"frames_per_s": round(steps / dt, 1), "us_per_frame": round(1e6 * dt / steps, 2), "B": 1, "T_in": T_in,
            "steps": steps, "note": "decoder loop only (incl. weight packing + memory_layer); encoder/postnet excluded"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-steps', type=int, default=2)
    ap.add_argument('--cpu-threads', type=int, default=8,
                    help='torch CPU threads for the baseline leg (the M=6 GEMVs of this model stop scaling\n'
                         'around 8 threads on this EPYC host: 8 → 3.7 s/it, 16 → 4.2, 32 → 7.4, all cores ≈ 40)')
    ap.add_argument('--no-decode', action='store_true')
    ap.add_argument('--koemo', action='store_true',
                    help='secondary workload of SURVEY 8(d): the ragged koemo length profile instead of the fixed shape\n'
                         '((T_in,T_out) = (84,400),(80,380),(71,350),(66,300),(50,260),(37,200); valid frames counted)')
    ap.add_argument('--settle', type=int, default=40,
                    help='extra untimed start-up steps after the graph capture (the first ~40 replays after process start\n'
                         'run ~4 %% slower than the steady state on this box; reported as config.startup_steps)')
    ap.add_argument('--no-graph', action='store_true',
                    help='run the step eagerly (one host launch per kernel) instead of replaying the captured HIP graph')
    ap.add_argument('--bf16', action='store_true',
                    help='BASELINE configs[4] instead of the headline config: bf16_run=True, B=16 per GPU')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
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
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group('nccl', init_method='env://', world_size=world, rank=rank)

    import hparams as HP
    import t2v_hip
    import train as TR
    t2v_hip.load_library()
    t2v_hip.DecoderCore.keep_last = True       # the roofline leg replays the last forward's kernels on its arena
    global B_PER_GPU
    if args.bf16:
        B_PER_GPU = 16
    hp = HP.create_hparams("batch_size=%d,anneal_function=constant%s%s" % (
        B_PER_GPU, ",distributed_run=True" if world > 1 else "", ",bf16_run=True" if args.bf16 else ""))
    torch.manual_seed(hp.seed)
    torch.cuda.manual_seed(hp.seed)
    engine = TR.TrainEngine(hp, world_size=world, graph=not args.no_graph)
    if args.koemo and not args.bf16:
        koemo_in, koemo_out = [84, 80, 71, 66, 50, 37], [400, 380, 350, 300, 260, 200]
        batch = synthetic_batch(B_PER_GPU, T_IN, T_OUT, 1234 + rank, lens_in=koemo_in, lens_out=koemo_out)
    else:
        batch = synthetic_batch(B_PER_GPU, T_IN, T_OUT, 1234 + rank)
    batch = tuple(t.pin_memory() for t in batch)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    it = 0
    with engine.stream_context():
        if engine.use_graph:
            # graph priming is start-up cost like building the model: the shape is captured the third time it is seen,
            # so three extra untimed steps make sure neither the warm-up nor the timed region contains the capture
            for _ in range(engine.GRAPH_AFTER + 1 + args.settle):
                engine.step(batch, it)
                it += 1
        for _ in range(args.warmup):
            engine.step(batch, it)
            it += 1
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = engine.step(batch, it)[0]
            it += 1
        sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device='cuda', dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    final_loss = float(loss.item())
    t2v_hip.check_async_errors()     # any bounded-spin timeout inside the timed steps invalidates the run

    ms_per_step = 1000.0 * elapsed / args.steps
    frames = B_PER_GPU * T_OUT * world
    if args.koemo and not args.bf16:
        frames = sum(koemo_out) * world          # valid (unpadded) frames, like the metric's definition
    value = frames / (elapsed / args.steps)

    out = {
        "metric": "mel-frames/s (train step, batch=%d, 80-mel)" % B_PER_GPU, "value": round(value, 1),
        "unit": "mel-frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.bf16 else "f32",
        "data": "synthetic",
        "config": {"workload": ("configs[4]: bf16_run train step (bf16 MFMA wide Conv1d fwd/dx + time-batched linears "
                                "+ LSTM dW GEMMs; fp32 master/BN/recurrence), B=16/GPU fixed shape T_in=84 T_out=400, "
                                "dropout on, random-init seed 1234") if args.bf16 else
                               ("configs[1], koemo length profile: fp32 train step, B=6/GPU ragged (T_in,T_out) = (84,400),"
                                "(80,380),(71,350),(66,300),(50,260),(37,200), valid frames counted, dropout on"
                                if args.koemo else
                                "configs[1]: Tacotron2-VAE fp32 train step (fwd+loss+bwd+clip+Adam), "
                                "B=6/GPU fixed shape T_in=84 T_out=400, dropout on, random-init seed 1234"),
                   "step_mode": "hip-graph replay" if engine.use_graph else "eager launches",
                   "startup_steps": (engine.GRAPH_AFTER + 1 + args.settle) if engine.use_graph else 0,
                   "global_batch": B_PER_GPU * world, "frames_per_step": frames,
                   "parallelism": "dp%d" % world},
        "final_loss": round(final_loss, 5),
    }

    if rank == 0:
        # ---- roofline leg: replay only k_lstm_fwd256's T+1 launches of the last forward, events on
        # the launch stream (torch's current stream is the stream the library launches on).
        reps = 3
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t2v_hip.replay_fwd_kernels(1)
        torch.cuda.synchronize()
        ev0.record()
        n = 0
        for _ in range(reps):
            n += t2v_hip.replay_fwd_kernels(1)
        ev1.record()
        torch.cuda.synchronize()
        us = 1000.0 * ev0.elapsed_time(ev1) / n
        achieved = LSTM_WEIGHT_BYTES / (us * 1e-6) / 1e9
        traffic = None      # HBM bytes per launch from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE, corrected)
        try:
            with open(os.path.join(ROOT, 'profiles', 'r01_pmc_fetch_size.json')) as f:
                traffic = json.load(f)["kernels"]["k_lstm_fwd256"]["corrected_bytes_per_launch"]
        except Exception:
            pass
        out["roofline"] = {"kernel": "k_lstm_fwd256", "bound": "hbm", "achieved": round(achieved, 1),
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                           "traffic": traffic, "avg_launch_us": round(us, 3),
                           "algorithmic_bytes_per_launch": LSTM_WEIGHT_BYTES, "launches_timed": n}
        e2e_bytes = 58.9e9   # compulsory bytes of one cfg-2 iteration
        out["roofline"]["end_to_end_frac"] = round(e2e_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if not args.no_decode:
            out["decode"] = decode_bench(engine.model)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(steps=args.cpu_steps, threads=args.cpu_threads)
            out["speedup_vs_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
