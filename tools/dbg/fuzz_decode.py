"""Random (T_in, B) through tests/test_inference_gpu.py::test_short_and_limit_texts_both_decode_paths_agree (free-running decode:
persistent kernel == launch-per-stage path == CPU oracle, 24 frames): `python tools/dbg/fuzz_decode.py [seed]` from the repo root."""
import os, sys, random
sys.path.insert(0, os.path.join(os.getcwd(), 'tacotron2-vae_amd'))
sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch
import hparams as HP
import model as M
import test_inference_gpu as TI
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 6)
hp = HP.create_hparams("max_decoder_steps=24")
M.drop_rate = 0.0
torch.manual_seed(hp.seed)
m = M.Tacotron2(hp).cuda().eval()
bad = 0
for it in range(16):
    T_in = rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 47, 64, 83, 100, 129, 200, 223, 224, 225, 300])
    B = rng.choice([1, 1, 2, 3, 4, 5, 8, 9])
    try:
        TI.test_short_and_limit_texts_both_decode_paths_agree(m, T_in, B)
        print("ok  ", T_in, B, flush=True)
    except Exception as e:
        bad += 1
        print("FAIL", T_in, B, repr(e)[:200], flush=True)
print("decode fuzz failures:", bad)
