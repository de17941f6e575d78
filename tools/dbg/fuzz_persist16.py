"""Random shapes through tests/test_decoder_persist16_gpu.py::test_persistent16_forward_matches_launch_per_step_bf16 (bf16 persistent
forward + reverse pass against the launch-per-step bf16 loop, reproducibility): `python tools/dbg/fuzz_persist16.py [seed]` from the repo root."""
import os, sys, random, traceback
sys.path.insert(0, os.path.join(os.getcwd(), 'tacotron2-vae_amd'))
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch
import test_decoder_persist16_gpu as T16
import t2v_hip as H
lib = H.load_library()
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
bad = 0
n = 0
while n < 36:
    B = rng.choice([1, 2, 5, 7, 9, 10, 11, 13, 14, 15, 16])
    T_in = rng.choice([1, 2, 3, 15, 16, 17, 31, 33, 47, 64, 65, 83, 95, 96, 97, 111, 128, 129, 160, 191, 192, 193, 200, 223, 224])
    T = rng.randint(2, 24)
    ragged = rng.random() < 0.7
    if lib.t2v_decoder_train_persist16_supported(B, T_in) != 1:
        continue
    n += 1
    try:
        T16.test_persistent16_forward_matches_launch_per_step_bf16(B, T_in, T, ragged)
        print("ok  ", B, T_in, T, ragged, "bwd16" if lib.t2v_decoder_bwd_persist16_supported(B, T_in) == 1 else "", flush=True)
    except Exception as e:
        bad += 1
        print("FAIL", B, T_in, T, ragged, repr(e)[:300], flush=True)
print("failures:", bad)
