import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import torch, bench, hparams as HP, train as TR, model as M
print('torch threads', torch.get_num_threads(), 'interop', torch.get_num_interop_threads())
hp = HP.create_hparams("batch_size=6,anneal_function=constant")
eng = TR.TrainEngine(hp, world_size=1, graph=False)
batch = tuple(t.pin_memory() for t in bench.synthetic_batch(6, bench.T_IN, bench.T_OUT, 1234))
orig = torch.Tensor.copy_
log = []
def timed(self, src, *a, **k):
    t0 = time.perf_counter(); r = orig(self, src, *a, **k); dt = time.perf_counter() - t0
    log.append((dt * 1e3, tuple(self.shape), str(self.device), str(src.device), bool(k.get('non_blocking', False))))
    return r
with eng.stream_context():
    for it in range(8):
        eng.step(batch, it)
    torch.Tensor.copy_ = timed
    ocat = torch.cat
    def tcat(*a, **k):
        t0 = time.perf_counter(); r = ocat(*a, **k); log.append(((time.perf_counter() - t0) * 1e3, tuple(r.shape), 'cat', str(r.device), False)); return r
    torch.cat = tcat
    t0 = time.perf_counter()
    eng.step(batch, 8)          # NOT drained: the queue still holds the previous steps
    t1 = time.perf_counter()
    torch.Tensor.copy_ = orig; torch.cat = ocat
    torch.cuda.synchronize()
print('step host time %.2f ms' % ((t1 - t0) * 1e3))
for e in sorted(log, reverse=True)[:12]:
    print('%.3f ms  dst %s %s <- %s nb=%s' % e)
