import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import torch, bench, hparams as HP, train as TR
hp = HP.create_hparams("batch_size=6,anneal_function=constant")
eng = TR.TrainEngine(hp, world_size=1, graph=False)
batch = tuple(t.pin_memory() for t in bench.synthetic_batch(6, bench.T_IN, bench.T_OUT, 1234))
with eng.stream_context():
    for it in range(24):
        t0 = time.perf_counter()
        eng.step(batch, it)
        dt = (time.perf_counter() - t0) * 1e3
        st = torch.cuda.memory_stats()
        print('step %2d host %.1f ms  allocated %.2f GB reserved %.2f GB  hipMalloc calls %d  gc counts %s' % (
            it, dt, torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30, st['num_device_alloc'], gc.get_count()))
    torch.cuda.synchronize()
