"""Free-running decode on the GPU vs the reference's golden inference run and vs the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def model():
    import hparams as HP
    import model as M
    hp = HP.create_hparams("max_decoder_steps=24")
    old = M.drop_rate
    M.drop_rate = 0.0
    torch.manual_seed(hp.seed)
    m = M.Tacotron2(hp).cuda().eval()
    yield m
    M.drop_rate = old


def test_inference_matches_reference_golden(model, golden_dir):
    g = np.load(os.path.join(golden_dir, 'inference.npz'))
    with torch.no_grad():
        ids = torch.from_numpy(g['ids']).cuda()
        emb = model.transcript_embedding(ids).transpose(1, 2)
        enc = model.encoder.inference(emb)
        style = model.vae_gst.fc3(torch.from_numpy(g['z']).cuda())
        memory = enc + style.unsqueeze(1)
        assert (memory.cpu() - torch.from_numpy(g['memory'])).abs().max() < 1e-4
        mel, gate, al = model.decoder.inference(memory)
        post = mel + model.postnet(mel)
    assert tuple(mel.shape) == g['mel'].shape and tuple(gate.shape) == g['gate'].shape
    assert tuple(al.shape) == g['align'].shape
    assert (mel.cpu() - torch.from_numpy(g['mel'])).abs().max() < 2e-4
    assert (mel.cpu() - torch.from_numpy(g['mel'])).abs().mean() < 1e-4
    assert (gate.cpu() - torch.from_numpy(g['gate'])).abs().max() < 2e-4
    assert (al.cpu() - torch.from_numpy(g['align'])).abs().max() < 2e-5
    assert torch.equal(al.cpu().argmax(-1), torch.from_numpy(g['align']).argmax(-1))    # alignment path
    assert (post.cpu() - torch.from_numpy(g['post'])).abs().max() < 5e-4


def test_stepwise_decode_api_equals_inference(model, golden_dir):
    """synthesizer.py:135-154 call sequence: initialize_decoder_states → prenet → decode per step."""
    g = np.load(os.path.join(golden_dir, 'inference.npz'))
    dec = model.decoder
    memory = torch.from_numpy(g['memory']).cuda()
    with torch.no_grad():
        mel_a, gate_a, al_a = dec.inference(memory)
        x = dec.get_go_frame(memory)
        dec.initialize_decoder_states(memory, mask=None)
        mels, gates, als = [], [], []
        for _ in range(mel_a.shape[2]):
            m, gt, al = dec.decode(dec.prenet(x))
            mels.append(m.clone()), gates.append(gt.clone()), als.append(al.clone())
            x = m
        mel_b, gate_b, al_b = dec.parse_decoder_outputs(mels, gates, als)
    # the per-step path evaluates the Prenet with torch (rocBLAS) instead of inside k_proj_prenet:
    # same math, different summation order
    assert (mel_a - mel_b).abs().max() < 2e-5 and (al_a - al_b).abs().max() < 1e-6
    assert (gate_a - gate_b).abs().max() < 2e-5
    assert dec.attention_hidden.shape == (1, 1024) and dec.attention_weights_cum.shape == (1, 30)


def test_gate_stop_and_batch(model):
    """Stop rule: force the gate bias high → stops after the first frame; B=3 runs to max steps and
    matches the oracle."""
    import t2v_oracle as O
    dec = model.decoder
    g = torch.Generator().manual_seed(2)
    memory = (torch.randn(3, 41, 512, generator=g) * 0.3).cuda()
    with torch.no_grad():
        mel, gate, al = dec.inference(memory, chunk=5)
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        o_mel, o_gate, o_al = O.decoder_inference(sd, memory.cpu(), max_steps=24)
        assert mel.shape == o_mel.shape
        assert (mel.cpu() - o_mel).abs().max() < 2e-4 and (al.cpu() - o_al).abs().max() < 2e-5
        old = dec.gate_layer.linear_layer.bias.clone()
        dec.gate_layer.linear_layer.bias.fill_(50.0)
        mel1, gate1, _ = dec.inference(memory[:1])
        dec.gate_layer.linear_layer.bias.copy_(old)
    assert mel1.shape[2] == 1 and gate1.shape == (1, 1, 1)


def _cfg4_model(hh_scale=1.0, gate_bias=None, max_steps=400):
    import hparams as HP
    import model as M
    hp = HP.create_hparams("max_decoder_steps=%d" % max_steps)
    torch.manual_seed(hp.seed)
    m = M.Tacotron2(hp).cuda().eval()
    with torch.no_grad():
        m.decoder.attention_rnn.weight_hh.mul_(hh_scale)
        m.decoder.decoder_rnn.weight_hh.mul_(hh_scale)
        if gate_bias is not None:
            m.decoder.gate_layer.linear_layer.bias.fill_(float(gate_bias))
    return m


@pytest.mark.parametrize("case", ['plain', 'lively'])
def test_gate_terminated_inference_matches_reference(golden_dir, case):
    """BASELINE configs[3] size (200 symbols, B = 1): the reference's own stop rule (model.py:453) ends the run — the
    HIP decode loop must stop at the same frame, with the same alignment path (fixture: oracle/gen_golden.py d2)."""
    import model as M
    g = np.load(os.path.join(golden_dir, 'inference_gate_stop.npz'))
    old = M.drop_rate
    M.drop_rate = 0.0
    try:
        m = _cfg4_model(float(g[case + '_hh_scale'][0]), float(g[case + '_gate_bias'][0]))
        with torch.no_grad():
            ids = torch.from_numpy(g['ids']).cuda()
            enc = m.encoder.inference(m.transcript_embedding(ids).transpose(1, 2))
            memory = enc + m.vae_gst.fc3(torch.from_numpy(g['z']).cuda()).unsqueeze(1)
            mel, gate, al = m.decoder.inference(memory, chunk=16)
        n = int(g[case + '_n_frames'][0])
        assert mel.shape[2] == n and n < 400, (mel.shape, n)
        tol = 2e-4 if case == 'plain' else 2e-2            # 'lively': x6 recurrent weights amplify fp32 round-off
        assert (mel.cpu() - torch.from_numpy(g[case + '_mel'])).abs().max() < tol
        assert (mel.cpu() - torch.from_numpy(g[case + '_mel'])).abs().mean() < tol / 4
        assert torch.equal(al.cpu().argmax(-1).to(torch.int16), torch.from_numpy(g[case + '_align_argmax']))
        assert (al.cpu().max(-1).values - torch.from_numpy(g[case + '_align_max'])).abs().max() < tol
        assert (al.cpu()[0, :4] - torch.from_numpy(g[case + '_align_head'])).abs().max() < 2e-5
    finally:
        M.drop_rate = old


def test_cfg4_decode_800_steps_matches_oracle():
    """BASELINE configs[3] at size: B = 1, 200 symbols, exactly 800 free-running decoder steps (gate ignored), HIP loop vs
    the CPU oracle: alignment argmax path equal, attention weights L1, mel."""
    import model as M
    import t2v_oracle as O
    old = M.drop_rate
    M.drop_rate = 0.0
    try:
        m = _cfg4_model(max_steps=800)
        m.decoder.gate_threshold = 1.0                      # never stop early
        g = torch.Generator().manual_seed(1234)
        ids = torch.randint(2, 80, (1, 200), generator=g)
        z = torch.randn(1, 32, generator=torch.Generator().manual_seed(7))
        with torch.no_grad():
            enc = m.encoder.inference(m.transcript_embedding(ids.cuda()).transpose(1, 2))
            memory = enc + m.vae_gst.fc3(z.cuda()).unsqueeze(1)
            import contextlib, io
            with contextlib.redirect_stdout(io.StringIO()):
                mel, gate, al = m.decoder.inference(memory, chunk=800)
            sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
            o_mel, o_gate, o_al = O.decoder_inference(sd, memory.cpu(), max_steps=800, gate_threshold=1.0)
        assert mel.shape == (1, 80, 800) and o_mel.shape == (1, 80, 800)
        assert torch.equal(al.cpu().argmax(-1), o_al.argmax(-1))                    # alignment path over all 800 frames
        assert (al.cpu() - o_al).abs().sum(-1).max() < 1e-3                         # per-frame L1 of the attention weights
        assert (mel.cpu() - o_mel).abs().max() < 5e-4 and (mel.cpu() - o_mel).abs().mean() < 1e-4
    finally:
        M.drop_rate = old


def test_more_than_eight_utterances_are_decoded_in_groups(model):
    """B = 11: groups of 8 + 3, each through the kernels it qualifies for, same values as decoding the groups by hand"""
    g = torch.Generator().manual_seed(77)
    memory = (torch.randn(11, 40, 512, generator=g) * 0.5).cuda()
    dec = model.decoder
    old_thr = dec.gate_threshold
    dec.gate_threshold = 1.0                                        # never stop: all frames of max_decoder_steps
    try:
        with torch.no_grad():
            mel, gate, al = dec.inference(memory)
            parts = [dec.inference(memory[:8]), dec.inference(memory[8:])]
    finally:
        dec.gate_threshold = old_thr
    n = parts[0][0].size(2)
    assert mel.shape == (11, 80, n) and gate.shape == (11, n, 1) and al.shape == (11, n, 40)
    for i, ref in enumerate((torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts]), torch.cat([p[2] for p in parts]))):
        assert (ref - (mel, gate, al)[i]).abs().max().item() < 2e-5


@pytest.mark.parametrize("T_in,B", [(1, 1), (3, 1), (16, 2), (224, 1), (225, 1), (40, 8)])
def test_short_and_limit_texts_both_decode_paths_agree(model, T_in, B):
    """Free-running decode at the edges of the persistent kernel's range (one symbol; one 16-position tile; T_in = 224 is
    its LDS limit, 225 falls back to the launch-per-stage path): persistent == launch-per-stage == oracle."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
    import t2v_oracle as O
    g = torch.Generator().manual_seed(100 + T_in)
    memory = (torch.randn(B, T_in, 512, generator=g) * 0.5).cuda()
    dec = model.decoder
    old_thr = dec.gate_threshold
    dec.gate_threshold = 1.0                                        # never stop: all 24 frames
    try:
        with torch.no_grad():
            mel_a, gate_a, al_a = dec.inference(memory, persistent=False)
            mel_b, gate_b, al_b = dec.inference(memory)             # persistent when supported
    finally:
        dec.gate_threshold = old_thr
    assert mel_a.shape == mel_b.shape == (B, 80, 24)
    assert (mel_a - mel_b).abs().max() < 2e-5 and (gate_a - gate_b).abs().max() < 2e-5
    assert (al_a - al_b).abs().max() < 2e-6
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    o_mel, o_gate, o_al = O.decoder_inference(sd, memory.cpu(), max_steps=24, stop_on_gate=False)
    assert (mel_b.cpu() - o_mel).abs().max() < 2e-4
    assert (al_b.cpu() - o_al).abs().max() < 2e-5
