"""The one-launch persistent teacher-forced decoder forward (csrc/decoder_train_persist.hip) against the launch-per-step
loop (decoder_fwd.hip + attn_fwd.hip) on the same inputs, state dropout ON: both draw the same counter-based masks, so
every saved array of the arena (XS, CA, CD, GA, GD, AL, ACUM, S) and every gradient of the backward pass — the
launch-per-step BPTT and the one-launch persistent reverse pass (csrc/decoder_train_bwd_persist.hip) alike — must
agree to fp32 summation order — the persistent path is a re-scheduling of the same arithmetic (reference loop
model.py:415-421), not an approximation."""
import pytest
import torch

pytestmark = pytest.mark.gpu

NAMES = ('gpre', 'memory', 'pm', 'lengths', 'XS', 'CA', 'CD', 'GA', 'GD', 'QP', 'AL', 'ACUM', 'S')


def _run(dec, mode, mem0, mels, lens, T, bwd=False):
    import t2v_hip as H
    H.DecoderCore.persistent = mode
    H.DecoderCore.persistent_bwd = bwd
    dec._calls = 0
    for q in dec.parameters():
        q.grad = None
    mem = mem0.clone().requires_grad_(True)
    mel, gate, al = dec(mem, mels, lens)
    used = H.DecoderCore.last_mode
    keep = H.DecoderCore.last_call[3]
    arena = {n: keep[i].clone() for i, n in enumerate(NAMES) if torch.is_tensor(keep[i]) and n in ('XS', 'CA', 'CD', 'GA', 'GD', 'AL', 'ACUM', 'S')}
    (mel.sum() + 0.3 * gate.sum() + 0.01 * (mel * mel).sum()).backward()
    torch.cuda.synchronize()
    H.check_async_errors()
    grads = {n: q.grad.clone() for n, q in dec.named_parameters() if q.grad is not None}
    grads['memory'] = mem.grad.clone()
    return used, mel.detach(), gate.detach(), al.detach(), arena, grads, H.DecoderCore.last_bwd_mode


@pytest.mark.parametrize("B,T_in,T,ragged", [(6, 84, 400, True), (6, 84, 40, False), (6, 84, 25, True), (1, 5, 7, False), (2, 16, 9, True),
                                             (4, 84, 12, True), (5, 130, 8, True), (6, 224, 5, True), (3, 200, 6, True),
                                             (6, 1, 4, False),
                                             # the long form (round 6): 224 < T_in <= 560, W_q / processed memory in registers, 96-position
                                             # slices in the reverse pass — koemo's longest sentence (555 symbols) at the headline batch, the
                                             # first length past the short form, a length of exactly one / of several full slices, one plane
                                             (6, 555, 30, True), (6, 225, 6, True), (4, 288, 9, True), (5, 560, 5, False), (2, 513, 7, True),
                                             (6, 400, 120, True)])
def test_persistent_forward_equals_launch_per_step(B, T_in, T, ragged):
    import hparams as HP
    import model as M
    import t2v_hip as H
    assert H.load_library().t2v_decoder_train_persist_supported(B, T_in) == 1
    old_drop, old_keep, old_mode = M.drop_rate, H.DecoderCore.keep_last, H.DecoderCore.persistent
    M.drop_rate = 0.0           # Prenet dropout is keyed by a per-call counter; the LSTM state dropout below stays ON
    H.DecoderCore.keep_last = True
    try:
        torch.manual_seed(0)
        dec = M.Decoder(HP.create_hparams()).cuda().train()
        dec.p_attention_dropout = dec.p_decoder_dropout = 0.1
        g = torch.Generator().manual_seed(1)
        mem0 = (torch.randn(B, T_in, 512, generator=g) * 0.5).cuda()
        mels = torch.randn(B, 80, T, generator=g).cuda()
        lens = torch.tensor([max(1, T_in - 7 * i) for i in range(B)] if ragged else [T_in] * B).cuda()
        a = _run(dec, False, mem0, mels, lens, T)
        b = _run(dec, True, mem0, mels, lens, T)
        c = _run(dec, True, mem0, mels, lens, T, bwd=True)       # + the one-launch persistent reverse pass
        assert a[0] == 'launch-per-step' and b[0] == 'persistent'
        assert a[6] == 'launch-per-step' and b[6] == 'launch-per-step' and c[6] == 'persistent'
        for i, name in ((1, 'mel'), (2, 'gate'), (3, 'alignments')):
            assert (a[i] - b[i]).abs().max().item() < 2e-6, name
        for n in a[4]:
            x, y = a[4][n], b[4][n]
            if n == 'XS':               # row T+1 carries h_dec(T-1) only; its other columns are never written
                x, y = torch.cat((x[:T + 1].flatten(), x[T + 1][:, 1536:].flatten())), torch.cat((y[:T + 1].flatten(), y[T + 1][:, 1536:].flatten()))
            assert not torch.isnan(y).any(), n
            assert (x - y).abs().max().item() < 5e-6 * max(1.0, x.abs().max().item()), n
        gmax = max(v.abs().max().item() for v in a[5].values())
        for n in a[5]:
            scale = a[5][n].abs().max().item()
            # (T_in = 1: alpha == 1, so the query / location / memory_layer gradients are rounding noise around zero)
            assert (a[5][n] - b[5][n]).abs().max().item() < 2e-5 * scale + 1e-6 * gmax + 1e-7, (n, scale, gmax)
            assert (a[5][n] - c[5][n]).abs().max().item() < 2e-5 * scale + 1e-6 * gmax + 1e-7, ('persistent backward', n, scale, gmax)
    finally:
        M.drop_rate, H.DecoderCore.keep_last, H.DecoderCore.persistent = old_drop, old_keep, old_mode
        H.DecoderCore.persistent_bwd = None
        H.DecoderCore.last_call = H.DecoderCore.last_bwd = None


@pytest.mark.parametrize("B,T_in,T", [(6, 576, 6), (3, 561, 9)])
def test_persistent_reverse_pass_beyond_the_forward_range(B, T_in, T):
    """561 … 576 symbols: the forward pass is out of range (560: LDS of the memory slice) and takes the launch-per-step loop, the
    reverse pass (96-position slices, six per item) still runs as one launch on the arena that loop saved"""
    import hparams as HP
    import model as M
    import t2v_hip as H
    lib = H.load_library()
    assert lib.t2v_decoder_train_persist_supported(B, T_in) == 0 and lib.t2v_decoder_bwd_persist_supported(B, T_in) == 1
    old_drop, old_keep, old_mode = M.drop_rate, H.DecoderCore.keep_last, H.DecoderCore.persistent
    M.drop_rate = 0.0
    H.DecoderCore.keep_last = True
    try:
        torch.manual_seed(0)
        dec = M.Decoder(HP.create_hparams()).cuda().train()
        dec.p_attention_dropout = dec.p_decoder_dropout = 0.1
        g = torch.Generator().manual_seed(1)
        mem0 = (torch.randn(B, T_in, 512, generator=g) * 0.5).cuda()
        mels = torch.randn(B, 80, T, generator=g).cuda()
        lens = torch.tensor([max(1, T_in - 7 * i) for i in range(B)]).cuda()
        a = _run(dec, None, mem0, mels, lens, T, bwd=False)
        c = _run(dec, None, mem0, mels, lens, T, bwd=True)
        assert a[0] == 'launch-per-step' and c[0] == 'launch-per-step' and a[6] == 'launch-per-step' and c[6] == 'persistent'
        assert torch.equal(a[1], c[1])
        gmax = max(v.abs().max().item() for v in a[5].values())
        for n in a[5]:
            scale = a[5][n].abs().max().item()
            assert (a[5][n] - c[5][n]).abs().max().item() < 2e-5 * scale + 1e-6 * gmax + 1e-7, (n, scale, gmax)
    finally:
        M.drop_rate, H.DecoderCore.keep_last, H.DecoderCore.persistent = old_drop, old_keep, old_mode
        H.DecoderCore.persistent_bwd = None
        H.DecoderCore.last_call = H.DecoderCore.last_bwd = None


def test_persistent_range_and_fallback():
    """outside B <= 6 / T_in <= 560 the library says so and the wrapper takes the launch-per-step loop"""
    import hparams as HP
    import model as M
    import t2v_hip as H
    lib = H.load_library()
    assert lib.t2v_decoder_train_persist_supported(6, 224) == 1
    assert lib.t2v_decoder_train_persist_supported(7, 84) == 0
    assert lib.t2v_decoder_train_persist_supported(6, 225) == 1 and lib.t2v_decoder_train_persist_supported(6, 560) == 1
    assert lib.t2v_decoder_train_persist_supported(6, 561) == 0 and lib.t2v_decoder_train_persist_supported(1, 561) == 0
    assert lib.t2v_decoder_bwd_persist_supported(6, 560) == 1 and lib.t2v_decoder_bwd_persist_supported(6, 577) == 0
    assert lib.t2v_decoder_bwd_persist_slices(224) == 7 and lib.t2v_decoder_bwd_persist_slices(225) == 3
    assert lib.t2v_decoder_bwd_persist_slices(555) == 6
    assert lib.t2v_decoder_train_persist_scratch_floats(6, 84, 400) == 402 * 2 * 2560 * 4 + 400 * 6 * 8 * 96
    torch.manual_seed(0)
    dec = M.Decoder(HP.create_hparams()).cuda().train()
    with torch.no_grad():
        dec(torch.randn(7, 30, 512, device='cuda'), torch.randn(7, 80, 3, device='cuda'), torch.full((7,), 30, device='cuda'))
    assert H.DecoderCore.last_mode == 'launch-per-step'
    with torch.no_grad():
        dec(torch.randn(2, 30, 512, device='cuda'), torch.randn(2, 80, 3, device='cuda'), torch.full((2,), 30, device='cuda'))
    assert H.DecoderCore.last_mode == 'persistent'
    torch.cuda.synchronize()
    H.check_async_errors()


def test_persistent_kernels_survive_a_neighbour_that_holds_cus():
    """A 64-workgroup spinner on a side stream occupies CUs while the persistent forward / reverse pass start (a
    communication kernel or another process on the GPU).  The 256 workgroups of a persistent launch then do not become resident
    together: the ones that are wait for the others inside their bounded spins.  The pass must finish with the SAME bits as an
    undisturbed one and without a time-out."""
    import ctypes as C
    import hparams as HP
    import model as M
    import t2v_hip as H
    hp = HP.create_hparams()
    torch.manual_seed(3)
    dec = M.Decoder(hp).cuda().train()
    old_drop = M.drop_rate
    M.drop_rate = 0.0           # Prenet dropout is keyed by a per-call counter; the LSTM state dropout stays ON
    B, T_in, T = 6, 84, 60
    g = torch.Generator().manual_seed(5)
    mem0 = (torch.randn(B, T_in, 512, generator=g) * 0.5).cuda()
    mels = torch.randn(B, 80, T, generator=g).cuda()
    lens = torch.tensor([84, 80, 71, 66, 50, 37], device='cuda')
    H.DecoderCore.keep_last = True
    old = (H.DecoderCore.persistent, H.DecoderCore.persistent_bwd)
    lib = H.load_library()
    side = torch.cuda.Stream()
    try:
        ref = _run(dec, True, mem0, mels, lens, T, bwd=True)
        assert ref[0] == 'persistent' and ref[6] == 'persistent'
        for spin_us in (300, 3000):
            # the spinner goes first: its 64 workgroups sit on 64 CUs when the persistent launch arrives
            H._check(lib.t2v_debug_spin(64, spin_us, C.c_void_p(side.cuda_stream)), 't2v_debug_spin')
            got = _run(dec, True, mem0, mels, lens, T, bwd=True)
            # ... and again in front of the reverse pass only (the forward of _run has drained the first one by then)
            assert got[0] == 'persistent' and got[6] == 'persistent'
            assert torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2])
            for n in ref[5]:
                assert torch.equal(got[5][n], ref[5][n]), n
        side.synchronize()
    finally:
        M.drop_rate = old_drop
        H.DecoderCore.persistent, H.DecoderCore.persistent_bwd = old
        H.DecoderCore.keep_last = False
        H.DecoderCore.last_call = H.DecoderCore.last_bwd = None
