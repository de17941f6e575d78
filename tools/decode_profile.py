"""Per-kernel breakdown of the free-running decode loop (run under rocprofv3 --kernel-trace)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import torch, hparams as HP, model as M, bench
hp = HP.create_hparams()
torch.manual_seed(hp.seed)
m = M.Tacotron2(hp).cuda().eval()
print(bench.decode_bench(m))
