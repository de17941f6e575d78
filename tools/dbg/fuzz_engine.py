"""Whole optimiser steps of the training engine (fp32 and bf16_run, eager and graph engine) on random batch shapes — finite losses,
no error word, eager == graph bit for bit — and free-running decode at random (B, T_in) on both decode paths:
`python tools/dbg/fuzz_engine.py [seed]` from the repo root."""
import os, sys, random
sys.path.insert(0, os.path.join(os.getcwd(), 'tacotron2-vae_amd'))
sys.path.insert(0, os.getcwd())
import torch
import hparams as HP
import train as TR
import t2v_hip
from bench import synthetic_batch
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 2)
bad = 0
for it in range(14):
    bf16 = rng.random() < (float(os.environ.get("FUZZ_BF16", "0.5")))
    B = rng.choice([1, 2, 3, 6, 7, 12, 16] if bf16 else [1, 2, 3, 5, 6])
    T_in = rng.choice([1, 5, 16, 17, 33, 84, 130, 192, 200, 224, 230])
    T_out = rng.randint(2, 24)
    lens_in = sorted([rng.randint(1, T_in) for _ in range(B)], reverse=True); lens_in[0] = T_in
    lens_out = [rng.randint(1, T_out) for _ in range(B)]; lens_out[0] = T_out
    res = {}
    try:
        for graph in (False, True):
            hp = HP.create_hparams("batch_size=%d,anneal_function=constant%s" % (B, ",bf16_run=True" if bf16 else ""))
            torch.manual_seed(hp.seed); torch.cuda.manual_seed(hp.seed)
            eng = TR.TrainEngine(hp, graph=graph)
            eng.graph_watchdog = False
            eng.model.vae_gst.eps_override = torch.full((B, 32), 0.125, device='cuda')
            batch = synthetic_batch(B, T_in, T_out, 3 + it, lens_in=lens_in, lens_out=lens_out)
            with eng.stream_context():
                out = [eng.step(batch, i) for i in range(5)]
            torch.cuda.synchronize()
            t2v_hip.check_async_errors()
            res[graph] = ([float(o[0]) for o in out], eng.optimizer.params.clone())
            assert all(torch.isfinite(torch.tensor(res[graph][0])))
            eng.close()
        same = res[False][0] == res[True][0] and torch.equal(res[False][1], res[True][1])
        if not same:
            bad += 1
        print("ok  " if same else "DIFF", "bf16" if bf16 else "fp32", B, T_in, T_out, t2v_hip.DecoderCore.last_kernel, t2v_hip.DecoderCore.last_bwd_kernel,
              "" if same else (res[False][0], res[True][0]), flush=True)
    except Exception as e:
        bad += 1
        print("FAIL", "bf16" if bf16 else "fp32", B, T_in, T_out, repr(e)[:300], flush=True)
    t2v_hip.set_bf16(False)
print("engine fuzz failures:", bad)
