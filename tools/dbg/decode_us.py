"""us per frame of the persistent decode (bench.py's decode leg only): python tools/dbg/decode_us.py [repeats]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd')); sys.path.insert(0, ROOT)
import torch
import hparams as HP, model as M
import bench
torch.manual_seed(1234)
m = M.Tacotron2(HP.create_hparams()).cuda().eval()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    d = bench.decode_bench(m)
    print("decode %.2f us/frame  (%s)" % (d["us_per_frame"], d.get("mode", "")[:40]))
