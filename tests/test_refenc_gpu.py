"""Reference-encoder kernels (CoordConv + Conv2d s2 + BN2d + ReLU, GRU), fused loss: HIP vs oracle/torch CPU."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def test_vae_gst_matches_oracle():
    import hparams as HP
    import modules as MD
    import t2v_oracle as O
    hp = HP.create_hparams()
    torch.manual_seed(5)
    vae = MD.VAE_GST(hp)
    g = torch.Generator().manual_seed(1)
    mel = torch.randn(3, 80, 137, generator=g) * 2 - 4
    eps = torch.randn(3, 32, generator=g)
    sd = {'vae_gst.' + k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and 'running' not in k)
          for k, v in vae.state_dict().items()}
    cm = mel.clone().requires_grad_(True)
    ref = O.vae_gst_forward(sd, cm, True, eps)
    wo = [torch.randn_like(t) for t in ref]
    sum((r * w).sum() for r, w in zip(ref, wo)).backward()

    vae = vae.cuda().train()
    vae.eps_override = eps.cuda()
    gm = mel.clone().cuda().requires_grad_(True)
    out = vae(gm)
    sum((o * w.cuda()).sum() for o, w in zip(out, wo)).backward()
    torch.cuda.synchronize()
    for o, r, name in zip(out, ref, ('style', 'mu', 'logvar', 'z')):
        assert (o.cpu() - r).abs().max() < 5e-5, name
    assert (gm.grad.cpu() - cm.grad).abs().max() < 2e-3 * cm.grad.abs().max()
    gmax = max(v.grad.abs().max().item() for v in sd.values() if v.grad is not None)
    for k, p in vae.named_parameters():
        rg = sd['vae_gst.' + k].grad
        if rg is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k      # dead CoordConv parent tensors (B-2)
            continue
        scale = max(rg.abs().max().item(), 1e-3 * gmax)    # conv biases ahead of a BatchNorm: exactly-zero gradient
        assert (p.grad.cpu() - rg).abs().max().item() < 3e-3 * scale, (k, (p.grad.cpu() - rg).abs().max().item(), scale)
    # eval mode: z = mu, running statistics
    vae.eval()
    sd_eval = {'vae_gst.' + k: v.detach().cpu() for k, v in vae.state_dict().items()}
    with torch.no_grad():
        e_out = vae(mel.cuda())
        e_ref = O.vae_gst_forward(sd_eval, mel, False)
    assert (e_out[0].cpu() - e_ref[0]).abs().max() < 5e-5 and torch.equal(e_out[1], e_out[3])


def test_fused_loss_matches_oracle():
    import hparams as HP
    import t2v_oracle as O
    from loss_function import Tacotron2Loss_VAE
    g = torch.Generator().manual_seed(0)
    B, T = 6, 400
    mel, post = torch.randn(B, 80, T, generator=g), torch.randn(B, 80, T, generator=g)
    gate = torch.randn(B, T, generator=g) * 3
    gate[0, -3:] = 1e3
    mu, logvar = torch.randn(B, 32, generator=g), torch.randn(B, 32, generator=g) * 0.3
    mel_t, gate_t = torch.randn(B, 80, T, generator=g), (torch.rand(B, T, generator=g) > 0.9).float()
    cpu = [t.clone().requires_grad_(True) for t in (mel, post, gate, mu, logvar)]
    ref = O.loss_forward([cpu[0], cpu[1], cpu[2], None, cpu[3], cpu[4]], mel_t, gate_t, 20000, 'logistic')
    ref[0].backward()
    dev = [t.clone().cuda().requires_grad_(True) for t in (mel, post, gate, mu, logvar)]
    crit = Tacotron2Loss_VAE(HP.create_hparams("anneal_function=logistic"))
    out = crit([dev[0], dev[1], dev[2], None, dev[3], dev[4]], (mel_t.cuda(), gate_t.cuda()), 20000)
    out[0].backward()
    for a, b in zip(out[:3], ref[:3]):
        assert abs(float(a) - float(b)) < 1e-5 * abs(float(b)) + 1e-6
    assert out[3] == ref[3]
    for d, c in zip(dev, cpu):
        assert (d.grad.cpu() - c.grad).abs().max() < 1e-5 * c.grad.abs().max() + 1e-9


@pytest.mark.parametrize("B,T", [(6, 7), (16, 16), (1, 1), (3, 2)])
def test_gru_last_matches_torch(B, T):
    """cooperative GRU kernels (8 workgroups, W_hh in registers) vs nn.GRU: last hidden state and all gradients."""
    import t2v_hip
    torch.manual_seed(B * 10 + T)
    gru = torch.nn.GRU(256, 256, batch_first=True)
    x = torch.randn(B, T, 256)
    wo = torch.randn(B, 256)
    xr = x.clone().requires_grad_(True)
    _, h = gru(xr)
    (h[0] * wo).sum().backward()
    params = [p.detach().clone().cuda().requires_grad_(True) for p in (gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0)]
    xg = x.clone().cuda().requires_grad_(True)
    out = t2v_hip.GRULast.apply(xg, *params)
    (out * wo.cuda()).sum().backward()
    torch.cuda.synchronize()
    assert (out.cpu() - h[0]).abs().max().item() < 2e-5
    refs = [xr.grad, gru.weight_ih_l0.grad, gru.weight_hh_l0.grad, gru.bias_ih_l0.grad, gru.bias_hh_l0.grad]
    for name, got, ref in zip(('dx', 'dw_ih', 'dw_hh', 'db_ih', 'db_hh'), [xg.grad] + [p.grad for p in params], refs):
        scale = ref.abs().max().item() + 1e-6
        assert (got.cpu() - ref).abs().max().item() < 2e-4 * scale + 1e-6, name
    # same bits on a second run (the exchange protocol has no data race)
    xg2 = x.clone().cuda().requires_grad_(True)
    out2 = t2v_hip.GRULast.apply(xg2, *[p.detach().clone().requires_grad_(True) for p in params])
    assert torch.equal(out, out2)
