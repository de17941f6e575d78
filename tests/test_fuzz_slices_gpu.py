"""Seeded slices of the fuzz passes of tools/dbg/fuzz_*.py as `-m gpu` tests: random shapes
through the bodies of the parity tests, each slice sized for ~30 s on the GPU box, so the "no failure" the rounds relied on is
evidence the driver's GPU run records.  The seeds are fixed: a failure names its shape."""
import os
import random
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_fuzz_slice_persist16_forward_and_reverse_pass():
    """bf16 persistent forward + reverse pass against the launch-per-step bf16 loop (and bit-reproducibility) on 10 random shapes
    with B 1…16, T_in 1…560 incl. the tile / slice edges and the long forms of the attention roles, ragged lengths, state dropout on (tools/dbg/fuzz_persist16.py)."""
    import t2v_hip as H
    import test_decoder_persist16_gpu as T16
    lib = H.load_library()
    rng = random.Random(20260)
    n = with_bwd = 0
    while n < 10:
        B = rng.choice([1, 2, 5, 7, 9, 10, 11, 13, 14, 15, 16])
        T_in = rng.choice([1, 2, 3, 15, 16, 17, 31, 33, 47, 64, 65, 83, 95, 96, 97, 111, 128, 129, 160, 191, 192, 193, 200, 223, 224,
                           225, 288, 289, 300, 384, 385, 480, 512, 513, 555, 560])
        T = rng.randint(2, 24)
        ragged = rng.random() < 0.7
        if lib.t2v_decoder_train_persist16_supported(B, T_in) != 1:
            continue
        n += 1
        with_bwd += lib.t2v_decoder_bwd_persist16_supported(B, T_in) == 1
        try:
            T16.test_persistent16_forward_matches_launch_per_step_bf16(B, T_in, T, ragged)
        except Exception as e:
            raise AssertionError("persist16 fuzz shape (B=%d, T_in=%d, T=%d, ragged=%s): %r" % (B, T_in, T, ragged, e))
    assert with_bwd >= 4        # (the slice must exercise the persistent reverse pass as well)


def test_fuzz_slice_decode_paths_against_oracle():
    """free-running decode: persistent kernel == launch-per-stage path == CPU oracle over 24 frames on 8 random (T_in, B) incl. the
    limits of the persistent launch (tools/dbg/fuzz_decode.py)."""
    import hparams as HP
    import model as M
    import test_inference_gpu as TI
    rng = random.Random(20261)
    hp = HP.create_hparams("max_decoder_steps=24")
    old = M.drop_rate
    M.drop_rate = 0.0
    try:
        torch.manual_seed(hp.seed)
        m = M.Tacotron2(hp).cuda().eval()
        for it in range(8):
            T_in = rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 47, 64, 83, 100, 129, 200, 223, 224, 225, 300])
            B = rng.choice([1, 1, 2, 3, 4, 5, 8, 9])
            try:
                TI.test_short_and_limit_texts_both_decode_paths_agree(m, T_in, B)
            except Exception as e:
                raise AssertionError("decode fuzz case (T_in=%d, B=%d): %r" % (T_in, B, e))
    finally:
        M.drop_rate = old


def test_fuzz_slice_engine_recurring_shapes_graph_equals_eager():
    """one engine fed 36 steps over 7 irregularly recurring batch shapes (capture, replay, the watchdog's comparison steps) against
    the eager engine on the same stream of batches: bit-identical trajectories (tools/dbg/fuzz_engine_shapes.py)."""
    import hparams as HP
    import t2v_hip
    import train as TR
    from bench import synthetic_batch
    rng = random.Random(20262)
    B = 6
    shapes = []
    for i in range(7):
        T_in = rng.choice([5, 17, 33, 60, 84, 100, 130, 190, 300, 555])       # (300, 555: the long forms of the persistent kernels)
        T_out = rng.randint(3, 24)
        Bs = rng.choice([B, B, max(1, B // 2), B - 1])
        shapes.append((Bs, T_in, T_out, sorted([rng.randint(1, T_in) for _ in range(Bs - 1)] + [T_in], reverse=True),
                       [T_out] + [rng.randint(1, T_out) for _ in range(Bs - 1)]))
    order = [rng.randrange(len(shapes)) if rng.random() < 0.6 else rng.randrange(2) for _ in range(36)]
    batches = {i: synthetic_batch(s[0], s[1], s[2], 10 + i, lens_in=s[3], lens_out=s[4]) for i, s in enumerate(shapes)}
    res = {}
    for graph in (False, True):
        hp = HP.create_hparams("batch_size=%d,anneal_function=constant" % B)
        torch.manual_seed(hp.seed)
        torch.cuda.manual_seed(hp.seed)
        eng = TR.TrainEngine(hp, graph=graph)
        eps = {b: torch.full((b, 32), 0.125, device='cuda') for b in {s[0] for s in shapes}}
        losses = []
        with eng.stream_context():
            for it, k in enumerate(order):
                eng.model.vae_gst.eps_override = eps[shapes[k][0]]
                losses.append(eng.step(batches[k], it)[0].clone())
        torch.cuda.synchronize()
        t2v_hip.check_async_errors()
        res[graph] = ([float(x) for x in losses], eng.optimizer.params.clone(), len(eng._graphs))
        eng.close()
    assert res[True][2] >= 1, "the slice must have captured and replayed at least one graph"
    assert all(x == x for x in res[True][0])
    first = next((i for i, (a, b) in enumerate(zip(res[False][0], res[True][0])) if a != b), None)
    assert first is None, ("first difference at step", first, shapes[order[first]][:3])
    assert torch.equal(res[False][1], res[True][1])


@pytest.mark.parametrize('bf16', [False, True], ids=['f32', 'bf16'])
def test_fuzz_slice_gemm_shapes_and_operand_forms(bf16):
    """random shapes / operand forms / accumulate flags through t2v_hip.gemm (every kernel it dispatches to, split-K forms included)
    against fp64 products of the (rounded) operands; every product twice: bit-reproducible (tools/dbg/fuzz_gemm.py)."""
    import t2v_hip
    rng = random.Random(20263 + int(bf16))
    t2v_hip.set_bf16(bf16)
    try:
        done = 0
        it = 0
        while done < 40:
            it += 1
            M = rng.choice([4, 37, 64, 128, 132, 256, 500, 1024, 1156, 2052, 4096, 6400])
            N = rng.choice([5, 80, 81, 128, 132, 256, 512, 1028, 1536, 2560])
            K = rng.choice([3, 32, 80, 96, 100, 256, 504, 515, 1024, 2400, 4096, 6400])
            if M * N * K > 3e10:
                continue
            done += 1
            ta, tb = rng.random() < 0.5, rng.random() < 0.5
            acc = rng.random() < 0.3
            g = torch.Generator().manual_seed(it)
            A = (torch.randn(K, M, generator=g).t() if ta else torch.randn(M, K, generator=g)).cuda()
            Bm = (torch.randn(K, N, generator=g).t() if tb else torch.randn(N, K, generator=g)).cuda()
            bias = torch.randn(N, generator=g).cuda() if rng.random() < 0.5 else None
            out0 = torch.randn(M, N, generator=g).cuda()
            out = out0.clone()
            t2v_hip.gemm(A, Bm, bias, out=out, accumulate=acc)
            out2 = out0.clone()
            t2v_hip.gemm(A, Bm, bias, out=out2, accumulate=acc)
            extra = (bias.double() if bias is not None else 0) + (out0.double() if acc else 0)
            ref = A.double() @ Bm.double().t() + extra
            scale = ref.abs().max().item() + 1e-9
            e_full = (out.double() - ref).abs().max().item() / scale
            case = (M, N, K, ta, tb, acc, bias is not None)
            assert torch.equal(out, out2), ('not reproducible', case)
            if bf16:
                refb = A.bfloat16().double() @ Bm.bfloat16().double().t() + extra
                e_b = (out.double() - refb).abs().max().item() / scale
                assert (e_b < 2e-3 or e_full < 1e-4) and e_full < 3e-2, (case, e_full, e_b)
            else:
                # fp32 products: fp32 round-off through K terms in a different summation order than the fp64 reference
                assert e_full < 3e-6, (case, e_full)
    finally:
        t2v_hip.set_bf16(False)
