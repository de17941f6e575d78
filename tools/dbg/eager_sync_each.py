import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import torch, bench, hparams as HP, train as TR
hp = HP.create_hparams("batch_size=6,anneal_function=constant")
eng = TR.TrainEngine(hp, world_size=1, graph=False)
TO = int(os.environ.get('TOUT', bench.T_OUT))
batch = tuple(t.pin_memory() for t in bench.synthetic_batch(6, bench.T_IN, TO, 1234))
import gc
if os.environ.get('NOGC'):
    gc.disable()
ts = []
with eng.stream_context():
    for it in range(int(os.environ.get('NSTEP', 22))):
        t0 = time.perf_counter()
        out = eng.step(batch, it)
        float(out[0].item())              # the reference loop reads the loss back every iteration (train.py:230)
        ts.append((time.perf_counter() - t0) * 1e3)
print('per-step wall with loss.item() each step (ms):', ' '.join('%.1f' % t for t in ts[4:]))
