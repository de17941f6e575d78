"""Random shapes / operand forms / accumulate flags through t2v_hip.gemm in bf16 and fp32 mode (all kernels incl. the split-K forms)
against fp64 products of the (rounded) operands; every product twice (bit-reproducibility): `python tools/dbg/fuzz_gemm.py`."""
import os, sys, random
sys.path.insert(0, os.path.join(os.getcwd(), 'tacotron2-vae_amd'))
import torch
import t2v_hip
lib = t2v_hip.load_library()
rng = random.Random(3)
bad = 0
for mode in (True, False):
    t2v_hip.set_bf16(mode)
    for it in range(60):
        M = rng.choice([4, 37, 64, 128, 132, 256, 500, 1024, 1156, 2052, 4096, 6400])
        N = rng.choice([5, 80, 81, 128, 132, 256, 512, 1028, 1536])
        K = rng.choice([3, 32, 80, 96, 100, 256, 504, 515, 1024, 2400, 4096, 6400])
        if M * N * K > 3e10:
            continue
        ta, tb = rng.random() < 0.5, rng.random() < 0.5
        acc = rng.random() < 0.3
        g = torch.Generator().manual_seed(it)
        A = (torch.randn(K, M, generator=g).t() if ta else torch.randn(M, K, generator=g)).cuda()
        B = (torch.randn(K, N, generator=g).t() if tb else torch.randn(N, K, generator=g)).cuda()
        bias = torch.randn(N, generator=g).cuda() if rng.random() < 0.5 else None
        out0 = torch.randn(M, N, generator=g).cuda()
        out = out0.clone()
        t2v_hip.gemm(A, B, bias, out=out, accumulate=acc)
        out2 = out0.clone()
        t2v_hip.gemm(A, B, bias, out=out2, accumulate=acc)
        ref = A.double() @ B.double().t() + (bias.double() if bias is not None else 0) + (out0.double() if acc else 0)
        refb = A.bfloat16().double() @ B.bfloat16().double().t() + (bias.double() if bias is not None else 0) + (out0.double() if acc else 0)
        scale = ref.abs().max().item() + 1e-9
        e_full = (out.double() - ref).abs().max().item() / scale
        e_b = (out.double() - refb).abs().max().item() / scale
        ok = torch.equal(out, out2) and (e_b < 2e-3 or e_full < 1e-4) and e_full < 3e-2
        if not ok:
            bad += 1
            print("FAIL bf16=%s" % mode, M, N, K, ta, tb, acc, e_full, e_b, torch.equal(out, out2), flush=True)
print("gemm fuzz failures:", bad)
