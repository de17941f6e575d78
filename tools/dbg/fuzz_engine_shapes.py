"""One training engine fed a stream of batches whose shapes recur irregularly (more distinct shapes than the engine keeps graphs for:
capture, replay, eviction, re-capture, the replay watchdog's comparison steps) against the eager engine on the same stream of
batches — the two trajectories must agree bit for bit: `python tools/dbg/fuzz_engine_shapes.py [seed] [--bf16]`."""
import os, sys, random
sys.path.insert(0, os.path.join(os.getcwd(), 'tacotron2-vae_amd'))
sys.path.insert(0, os.getcwd())
import torch
import hparams as HP
import train as TR
import t2v_hip
from bench import synthetic_batch
args = [a for a in sys.argv[1:] if not a.startswith('--')]
bf16 = '--bf16' in sys.argv
rng = random.Random(int(args[0]) if args else 1)
B = 16 if bf16 else 6
shapes = []
for i in range(11):
    T_in = rng.choice([5, 17, 33, 60, 84, 100, 130, 190])
    T_out = rng.randint(3, 30)
    Bs = rng.choice([B, B, max(1, B // 2), B - 1])
    shapes.append((Bs, T_in, T_out, sorted([rng.randint(1, T_in) for _ in range(Bs - 1)] + [T_in], reverse=True),
                   [T_out] + [rng.randint(1, T_out) for _ in range(Bs - 1)]))
order = [rng.randrange(len(shapes)) if rng.random() < 0.7 else rng.randrange(3) for _ in range(70)]
batches = {i: synthetic_batch(s[0], s[1], s[2], 10 + i, lens_in=s[3], lens_out=s[4]) for i, s in enumerate(shapes)}
res = {}
for graph in (False, True):
    hp = HP.create_hparams("batch_size=%d,anneal_function=constant%s" % (B, ",bf16_run=True" if bf16 else ""))
    torch.manual_seed(hp.seed); torch.cuda.manual_seed(hp.seed)
    eng = TR.TrainEngine(hp, graph=graph)
    losses = []
    # (one eps tensor per batch size, alive for the whole run: a captured graph keeps the ADDRESS of the tensor it was captured with)
    eps = {b: torch.full((b, 32), 0.125, device='cuda') for b in {s[0] for s in shapes}}
    with eng.stream_context():
        for it, k in enumerate(order):
            eng.model.vae_gst.eps_override = eps[shapes[k][0]]
            losses.append(eng.step(batches[k], it)[0].clone())
    torch.cuda.synchronize()
    t2v_hip.check_async_errors()
    res[graph] = ([float(x) for x in losses], eng.optimizer.params.clone(), len(getattr(eng, '_graphs', {})), getattr(eng, 'graph_fallbacks', 0))
    eng.close()
    t2v_hip.set_bf16(False)
same = res[False][0] == res[True][0] and torch.equal(res[False][1], res[True][1])
first = next((i for i, (a, b) in enumerate(zip(res[False][0], res[True][0])) if a != b), None)
print("bf16" if bf16 else "fp32", "70 steps over %d shapes: graphs held %d, watchdog fall-backs %d, eager == graph: %s%s" % (
    len(shapes), res[True][2], res[True][3], same, "" if same else " (first difference at step %s, shape %s)" % (first, shapes[order[first]][:3] if first is not None else None)))
print("final loss", res[False][0][-1], res[True][0][-1], "all finite:", all(x == x for x in res[True][0]))
