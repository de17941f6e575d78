import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import torch, hparams as HP, train as TR
from bench import synthetic_batch
hp = HP.create_hparams("batch_size=6,anneal_function=constant")
torch.manual_seed(1234)
eng = TR.TrainEngine(hp, graph=True)
batch = tuple(t.pin_memory() for t in synthetic_batch(6, 84, 400, 1234))
with eng.stream_context():
    for it in range(4): eng.step(batch, it)
    torch.cuda.synchronize()
    for blk in range(15):
        t0 = time.perf_counter()
        for i in range(20): eng.step(batch, 10 + blk * 20 + i)
        torch.cuda.synchronize()
        print('replays %3d-%3d: %.3f ms/step' % (blk * 20, blk * 20 + 19, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
        if blk == 2:
            print('-- one trivial kernel on the default stream')
            with torch.cuda.stream(torch.cuda.default_stream()):
                z = torch.zeros(16, device='cuda'); z += 1
            torch.cuda.synchronize()
        if blk == 5:
            print('-- 3000 trivial kernels on the default stream')
            with torch.cuda.stream(torch.cuda.default_stream()):
                for i in range(3000): z += 1
            torch.cuda.synchronize()
        if blk == 8:
            print('-- eager engine: 3 steps on the default stream')
            eng2 = TR.TrainEngine(hp, graph=False)
            with torch.cuda.stream(torch.cuda.default_stream()):
                for i in range(3): eng2.step(batch, i)
            torch.cuda.synchronize()
        if blk == 11:
            print('-- eager engine: 20 steps on the default stream')
            with torch.cuda.stream(torch.cuda.default_stream()):
                for i in range(20): eng2.step(batch, i)
            torch.cuda.synchronize()
