"""The small fused launches of round 4 (csrc/glue.hip) against the tensor-op chains they replace: output masking
(reference model.py:509-520), VAE reparameterisation (modules.py:74-81) with its gradient, the gather of the projection's
input rows (model.py:385-388 torch.cat) and the batched error-word gather of the asynchronous ledger."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_mask_outputs_matches_masked_fill():
    import t2v_hip as H
    from utils import get_mask_from_lengths
    g = torch.Generator().manual_seed(3)
    B, Cn, T = 5, 80, 37
    mel, post, gate = torch.randn(B, Cn, T, generator=g).cuda(), torch.randn(B, Cn, T, generator=g).cuda(), torch.randn(B, T, generator=g).cuda()
    lens = torch.tensor([37, 36, 20, 1, 0])
    pad = ~get_mask_from_lengths(lens.cuda(), T)
    ref = (mel.masked_fill(pad.unsqueeze(1), 0.0), post.masked_fill(pad.unsqueeze(1), 0.0), gate.masked_fill(pad, 1e3))
    H.mask_outputs(mel, post, gate, lens.to(device='cuda', dtype=torch.int32))
    torch.cuda.synchronize()
    for a, r in zip((mel, post, gate), ref):
        assert torch.equal(a, r)


def test_reparam_matches_autograd():
    import t2v_hip as H
    g = torch.Generator().manual_seed(5)
    eps, mu, lv = (torch.randn(6, 32, generator=g).cuda() for _ in range(3))
    w = torch.randn(6, 32, generator=g).cuda()
    mu1, lv1 = mu.clone().requires_grad_(True), lv.clone().requires_grad_(True)
    ((eps * torch.exp(0.5 * lv1) + mu1) * w).sum().backward()
    mu2, lv2 = mu.clone().requires_grad_(True), lv.clone().requires_grad_(True)
    z = H.Reparam.apply(eps, mu2, lv2)
    (z * w).sum().backward()
    torch.cuda.synchronize()
    assert (z - (eps * torch.exp(0.5 * lv) + mu)).abs().max().item() < 1e-6
    assert torch.equal(mu2.grad, mu1.grad)
    assert (lv2.grad - lv1.grad).abs().max().item() < 1e-6 * max(1.0, lv1.grad.abs().max().item())


def test_concat2_rows_gathers_strided_halves():
    import t2v_hip as H
    lib = H.load_library()
    g = torch.Generator().manual_seed(7)
    XS = torch.randn(9, 3, 2560, generator=g).cuda()            # the decoder arena's row layout
    T = 7
    ref = torch.cat((XS[2:T + 2, :, 1536:], XS[1:T + 1, :, 1024:1536]), 2)
    out = torch.empty(T, 3, 1536, device='cuda')
    H._check(lib.t2v_concat2_rows(H._p(XS[2:, :, 1536:]), 2560, 1024, H._p(XS[1:, :, 1024:]), 2560, 512, H._p(out), T * 3,
                                  H._stream()), 't2v_concat2_rows')
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    assert lib.t2v_concat2_rows(H._p(XS), 2560, 1022, H._p(XS), 2560, 512, H._p(out), 3, H._stream()) != 0     # widths must be float4s


def test_error_words_are_gathered_in_one_launch_and_found():
    import t2v_hip as H
    H.check_async_errors()
    words = [torch.zeros(1, device='cuda', dtype=torch.int32) for _ in range(20)]
    words[13].fill_(1)
    for i, w in enumerate(words):
        H._err_note('unit test word %d' % i, w)
    with pytest.raises(H.T2VHipError) as e:
        H.check_async_errors()
    assert e.value.labels == ['unit test word 13']
    H.check_async_errors()          # the ledger is clean again
