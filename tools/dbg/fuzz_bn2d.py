"""Random shapes through tests/test_conv_bn_gpu.py::test_conv2d_s2_bn_relu_matches_torch (reference-encoder layer: strided conv,
BatchNorm2d incl. the split of wide channels over several workgroups, ReLU; output, running statistics, five gradients):
`python tools/dbg/fuzz_bn2d.py` from the repo root."""
import os, sys, random
sys.path.insert(0, os.path.join(os.getcwd(), 'tacotron2-vae_amd'))
sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import test_conv_bn_gpu as TC
rng = random.Random(9)
bad = 0
for it in range(24):
    B = rng.choice([1, 2, 5, 6, 9, 16])
    Cx, coord = rng.choice([(1, True), (32, False), (5, True), (64, False)])
    H = rng.choice([9, 50, 100, 199, 200, 400])
    W = rng.choice([5, 20, 40, 41, 80])
    Cout = rng.choice([6, 32, 64, 128])
    if B * Cx * H * W > 3e6:
        continue
    try:
        TC.test_conv2d_s2_bn_relu_matches_torch(B, Cx, H, W, Cout, coord, True)
        print("ok  ", B, Cx, H, W, Cout, coord, flush=True)
    except Exception as e:
        bad += 1
        print("FAIL", B, Cx, H, W, Cout, coord, repr(e)[:200], flush=True)
print("bn2d fuzz failures:", bad)
