"""Encoder BiLSTM cooperative kernels vs torch's packed nn.LSTM on CPU (forward + all gradients)."""
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lens,T", [([20, 17, 9], 20), ([84, 80, 71, 66, 50, 37], 84), ([5], 5), ([33] * 16, 33),
                                    ([1, 1], 1)])
def test_bilstm_matches_packed_lstm(lens, T):
    import t2v_hip
    B = len(lens)
    torch.manual_seed(3)
    lstm = nn.LSTM(512, 256, 1, batch_first=True, bidirectional=True)
    g = torch.Generator().manual_seed(B + T)
    x = torch.randn(B, T, 512, generator=g)
    wo = torch.randn(B, T, 512, generator=g)
    lengths = torch.tensor(lens)
    cx = x.clone().requires_grad_(True)
    packed = nn.utils.rnn.pack_padded_sequence(cx, lengths, batch_first=True)
    out, _ = lstm(packed)
    ref, _ = nn.utils.rnn.pad_packed_sequence(out, batch_first=True, total_length=T)
    (ref * wo).sum().backward()

    dev = 'cuda'
    P = {k: v.detach().clone().to(dev).requires_grad_(True) for k, v in lstm.named_parameters()}
    gx = x.clone().to(dev).requires_grad_(True)
    y = t2v_hip.BiLSTM.apply(gx, lengths.to(dev).int(), P['weight_ih_l0'], P['weight_hh_l0'], P['bias_ih_l0'],
                             P['bias_hh_l0'], P['weight_ih_l0_reverse'], P['weight_hh_l0_reverse'],
                             P['bias_ih_l0_reverse'], P['bias_hh_l0_reverse'], True)
    (y * wo.to(dev)).sum().backward()
    torch.cuda.synchronize()
    assert (y.cpu() - ref).abs().max() < 2e-5
    for b, n in enumerate(lens):
        assert float(y[b, n:].abs().max()) == 0.0 if n < T else True
    assert (gx.grad.cpu() - cx.grad).abs().max() < 2e-3 * cx.grad.abs().max() + 1e-6
    for k, p in lstm.named_parameters():
        d = (P[k].grad.cpu() - p.grad).abs().max().item()
        assert d < 2e-3 * p.grad.abs().max().item() + 1e-6, (k, d)


def test_bilstm_deterministic_and_no_timeout():
    import t2v_hip
    torch.manual_seed(0)
    lstm = nn.LSTM(512, 256, 1, batch_first=True, bidirectional=True).cuda()
    x = torch.randn(6, 84, 512, device='cuda')
    lengths = torch.tensor([84, 80, 71, 66, 50, 37], device='cuda', dtype=torch.int32)
    args = [p.detach() for p in (lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0,
                                 lstm.weight_ih_l0_reverse, lstm.weight_hh_l0_reverse, lstm.bias_ih_l0_reverse,
                                 lstm.bias_hh_l0_reverse)]
    outs = [t2v_hip.BiLSTM.apply(x, lengths, *args, False) for _ in range(20)]
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0], o) for o in outs[1:])
