"""Host time of one graph replay of the training step (how long hipGraphLaunch keeps the calling thread busy) next to
the GPU time of the step: a replay that needs more host time than the GPU needs for the step makes the step host-bound."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import torch, bench, hparams as HP, train as TR
hp = HP.create_hparams("batch_size=6,anneal_function=constant")
torch.manual_seed(hp.seed)
eng = TR.TrainEngine(hp, graph=True)
batch = tuple(t.pin_memory() for t in bench.synthetic_batch(6, bench.T_IN, bench.T_OUT, 1234))
with eng.stream_context():
    for it in range(8):
        eng.step(batch, it)
    torch.cuda.synchronize()
    (key, entry), = eng._graphs.items()
    g = entry[0]
    host = []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        host.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("single replay: host call %.2f ms, until the GPU is done %.2f ms (median of 10)" % (
    sorted(h[0] for h in host)[5], sorted(h[1] for h in host)[5]))
print("10 replays back to back: host calls returned after %.2f ms each, GPU done after %.2f ms each" % ((t1 - t0) * 100, (t2 - t0) * 100))
