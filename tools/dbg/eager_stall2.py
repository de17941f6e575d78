import os, sys, time, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import torch, bench, hparams as HP, train as TR
hp = HP.create_hparams("batch_size=6,anneal_function=constant")
eng = TR.TrainEngine(hp, world_size=1, graph=False)
batch = tuple(t.pin_memory() for t in bench.synthetic_batch(6, bench.T_IN, bench.T_OUT, 1234))
shown = 0
with eng.stream_context():
    for it in range(30):
        pr = cProfile.Profile(); pr.enable()
        t0 = time.perf_counter()
        out = eng.step(batch, it)
        t1 = time.perf_counter()
        float(out[0].item())
        t2 = time.perf_counter()
        pr.disable()
        if it > 3 and (t2 - t0) > 0.04 and shown < 3:
            shown += 1
            s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(5)
            print('step %d: issue %.1f ms, item() %.1f ms' % (it, (t1 - t0) * 1e3, (t2 - t1) * 1e3)); print('\n'.join(s.getvalue().split('\n')[6:14]))
