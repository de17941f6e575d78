import os, sys, time, torch
sys.path.insert(0, '/root/repo/tacotron2-vae_amd'); sys.path.insert(0, '/root/repo')
import hparams as HP, train as TR, model as M
from bench import synthetic_batch
for mode in (True, False, True, False):
    M.Tacotron2.overlap_branches = mode
    hp = HP.create_hparams("batch_size=6,anneal_function=constant")
    torch.manual_seed(1234)
    eng = TR.TrainEngine(hp)
    batch = tuple(t.pin_memory() for t in synthetic_batch(6, 84, 400, 1234))
    for it in range(5): eng.step(batch, it)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in range(30): eng.step(batch, 5 + it)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print('overlap_branches=%s: %.3f ms/step' % (mode, dt * 1e3))
    del eng
