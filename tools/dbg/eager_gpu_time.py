import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import torch, bench, hparams as HP, train as TR
hp = HP.create_hparams("batch_size=6,anneal_function=constant")
eng = TR.TrainEngine(hp, world_size=1, graph=False)
batch = tuple(t.pin_memory() for t in bench.synthetic_batch(6, bench.T_IN, bench.T_OUT, 1234))
N = 18
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
host = []
with eng.stream_context():
    for it in range(N):
        t0 = time.perf_counter()
        evs[it][0].record()
        eng.step(batch, it)
        evs[it][1].record()
        host.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
for it in range(4, N):
    print('step %2d: host issue %.1f ms, GPU start->end %.1f ms, GPU end(prev)->start %.1f ms' % (
        it, host[it], evs[it][0].elapsed_time(evs[it][1]), evs[it - 1][1].elapsed_time(evs[it][0])))
