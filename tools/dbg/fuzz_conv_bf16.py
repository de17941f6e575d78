"""Random shapes through the bf16 Conv1d tests of tests/test_bf16_gpu.py (forward, data gradient, weight gradient):
`python tools/dbg/fuzz_conv_bf16.py` from the repo root."""
import os, sys, random
sys.path.insert(0, os.path.join(os.getcwd(), 'tacotron2-vae_amd'))
sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch
import t2v_hip
import test_bf16_gpu as TB
rng = random.Random(5)
bad = 0
t2v_hip.set_bf16(True)
for it in range(30):
    B = rng.choice([1, 2, 3, 6, 16])
    Cin = rng.choice([128, 160, 256, 512])
    Cout = rng.choice([128, 144, 256, 512])
    T = rng.choice([5, 16, 37, 80, 84, 95, 96, 97, 161, 400])
    try:
        TB.test_conv_bf16_forward_and_data_gradient(None, B, Cin, Cout, T)
        TB.test_conv_bf16_weight_gradient(B, Cin, Cout, T)
        print("ok  ", B, Cin, Cout, T, flush=True)
    except Exception as e:
        bad += 1
        print("FAIL", B, Cin, Cout, T, repr(e)[:200], flush=True)
t2v_hip.set_bf16(False)
print("conv fuzz failures:", bad)
