"""How long does the HOST need to enqueue one train step (no device sync inside the loop)?"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import hparams as HP, train as TR
from bench import synthetic_batch
hp = HP.create_hparams("batch_size=6,anneal_function=constant")
torch.manual_seed(1234)
eng = TR.TrainEngine(hp)
batch = tuple(t.pin_memory() for t in synthetic_batch(6, 84, 400, 1234))
for it in range(5): eng.step(batch, it)
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(10): eng.step(batch, 5 + it)
t_host = (time.perf_counter() - t0) / 10
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / 10
print('host enqueue %.2f ms/step, wall %.2f ms/step' % (t_host * 1e3, t_all * 1e3))
# phase split on the host
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for it in range(5): eng.step(batch, 20 + it)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime'); st.print_stats(22)
