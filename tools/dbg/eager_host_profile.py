"""Host-side profile of EAGER training steps (ragged batches never repeat a shape, so real training runs eagerly)."""
import cProfile, pstats, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import torch, bench, hparams as HP, train as TR
hp = HP.create_hparams("batch_size=6,anneal_function=constant")
torch.manual_seed(hp.seed)
eng = TR.TrainEngine(hp, world_size=1, graph=False)
batch = tuple(t.pin_memory() for t in bench.synthetic_batch(6, bench.T_IN, bench.T_OUT, 1234))
with eng.stream_context():
    for it in range(5):
        eng.step(batch, it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(5, 15):
        eng.step(batch, it)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('10 eager steps: host issue %.2f ms/step, incl. drain %.2f ms/step' % ((t1 - t0) * 100, (t2 - t0) * 100))
    pr = cProfile.Profile()
    pr.enable()
    for it in range(15, 20):
        eng.step(batch, it)
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(8)
st.print_callers('copy_')
