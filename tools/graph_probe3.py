import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import torch, hparams as HP, train as TR
from bench import synthetic_batch
hp = HP.create_hparams("batch_size=6,anneal_function=constant")
torch.manual_seed(1234)
eng = TR.TrainEngine(hp, graph=True)
batch = tuple(t.pin_memory() for t in synthetic_batch(6, 84, 400, 1234))
def smi(tag):
    out = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True).stdout
    keep = [l.strip() for l in out.split('\n') if any(k in l for k in ('sclk', 'mclk', 'fclk', 'socclk', 'Power'))]
    print(tag, ' | '.join(k.split(':', 1)[-1].strip() if 'GPU[' in k else k for k in keep)[:400], flush=True)
with eng.stream_context():
    for it in range(4): eng.step(batch, it)
    torch.cuda.synchronize()
    for blk in range(8):
        t0 = time.perf_counter()
        for i in range(20):
            eng.step(batch, 10 + blk * 20 + i)
            if i == 10: smi('   clocks:')
        torch.cuda.synchronize()
        print('replays %3d-%3d: %.3f ms/step' % (blk * 20, blk * 20 + 19, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
        if blk == 3:
            print('-- 3000 trivial kernels on the default stream')
            z = torch.zeros(16, device='cuda')
            with torch.cuda.stream(torch.cuda.default_stream()):
                for i in range(3000): z += 1
            torch.cuda.synchronize()
