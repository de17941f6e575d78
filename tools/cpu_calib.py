"""Pick the torch CPU thread count for bench.py's cpu_baseline leg (run on the GPU box's host)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tacotron2-vae_amd'))
import bench
for th in (8, 16, 32):
    r = bench.cpu_baseline(steps=1, warmup=1, threads=th)
    print(th, 'threads:', r['s_per_it'], 's/it', r['value'], 'frames/s', flush=True)
