"""Pins oracle/t2v_oracle.py to the reference: every array here was produced by the reference's own
code (oracle/gen_golden.py).  Also pins the product's boundary modules (init recipe, state_dict
layout).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch


def _digest(t):
    t = t.detach().double()
    return [float(t.sum()), float(t.abs().sum()), float((t * t).sum())]


@pytest.fixture(scope='module')
def model_sd():
    import hparams as HP
    import model as M
    hp = HP.create_hparams()
    torch.manual_seed(hp.seed)
    m = M.Tacotron2(hp)
    return hp, m, {k: v.detach().clone() for k, v in m.state_dict().items()}


@pytest.fixture(scope='module')
def digests(golden_dir):
    with open(os.path.join(golden_dir, 'train_step_digests.json')) as f:
        return json.load(f)


def test_init_recipe_bit_exact(model_sd, digests):
    """seed 1234 + same container order ⇒ the reference's step-0 weights (Appendix B-16)."""
    _, _, sd = model_sd
    assert list(sd.keys()) == list(digests['init'].keys())
    for k, v in sd.items():
        assert _digest(v) == digests['init'][k], k


def test_state_dict_manifest_and_checkpoint_schema(model_sd, golden_dir):
    _, m, sd = model_sd
    with open(os.path.join(golden_dir, 'checkpoint_schema.json')) as f:
        sch = json.load(f)
    assert [[k, list(v.shape), str(v.dtype).replace('torch.', '')] for k, v in sd.items()] == sch['state_dict']
    assert len(sch['state_dict']) == 142
    assert sch['top_keys'] == ['iteration', 'state_dict', 'optimizer', 'learning_rate']
    # parameters that never get a gradient in the reference == the ones our arena optimiser skips
    import optim
    names = [n for n, _ in m.named_parameters()]
    assert len(names) == sch['n_parameters'] == 100
    live = [i for i, n in enumerate(names) if not optim.is_dead_param(n)]
    assert live == sch['optimizer_state_indices'] and len(live) == 94


@pytest.fixture(scope='module')
def oracle_step(model_sd, golden_dir):
    import t2v_oracle as O
    _, _, sd0 = model_sd
    g = np.load(os.path.join(golden_dir, 'train_step.npz'))
    sd = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and 'running_' not in k else v.clone())
          for k, v in sd0.items()}
    text, lin = torch.from_numpy(g['text']), torch.from_numpy(g['input_lengths'])
    mel, gate, lout = torch.from_numpy(g['mel']), torch.from_numpy(g['gate']), torch.from_numpy(g['output_lengths'])
    out = O.tacotron2_forward(sd, text, lin, mel, lout, True, torch.from_numpy(g['eps']))
    loss, recon, kl, w = O.loss_forward(out, mel, gate, 0, 'constant')
    loss.backward()
    return g, sd, out, (loss, recon, kl, w)


def test_oracle_forward_matches_reference(oracle_step):
    g, sd, out, (loss, recon, kl, w) = oracle_step
    for i, name in enumerate(['out_mel', 'out_post', 'out_gate', 'out_align', 'out_mu', 'out_logvar', 'out_z']):
        ref = torch.from_numpy(g[name])
        assert (out[i].detach() - ref).abs().max().item() < 5e-5, name
    assert abs(float(loss) - g['scalars'][0]) < 1e-4 * abs(g['scalars'][0])
    assert abs(float(kl) - g['scalars'][2]) < 1e-4 * abs(g['scalars'][2]) + 1e-6
    assert w == g['scalars'][3] == 0.001


def test_oracle_gradients_match_reference(oracle_step, digests):
    g, sd, out, _ = oracle_step
    gd = digests['grads_step0']
    for k in digests['no_grad_params']:
        assert sd[k].grad is None, k
    gmax = max(np.sqrt(v['digest'][2]) for v in gd.values())
    for k, v in gd.items():
        gr = sd[k].grad
        assert gr is not None, k
        head = gr.reshape(-1)[:8].double().numpy()
        scale = max(np.sqrt(v["digest"][2]), 1e-4 * gmax)      # tensor L2 norm (zero-gradient conv biases → tiny)
        assert np.abs(head - np.array(v['head'])).max() < 1e-3 * scale + 1e-7, k
        assert abs(np.sqrt(float((gr.double() ** 2).sum())) - np.sqrt(v['digest'][2])) < 2e-3 * scale + 1e-7, k


def test_oracle_inference_matches_reference(model_sd, golden_dir):
    import t2v_oracle as O
    _, _, sd = model_sd
    g = np.load(os.path.join(golden_dir, 'inference.npz'))
    with torch.no_grad():
        mel, gate, al = O.decoder_inference(sd, torch.from_numpy(g['memory']), max_steps=24)
        post = mel + O.postnet_forward(sd, mel, training=False)
    assert mel.shape == torch.Size(g['mel'].shape)
    assert (mel - torch.from_numpy(g['mel'])).abs().max() < 1e-4
    assert (al - torch.from_numpy(g['align'])).abs().max() < 1e-5
    assert (post - torch.from_numpy(g['post'])).abs().max() < 1e-4
    assert torch.equal(al.argmax(-1), torch.from_numpy(g['align']).argmax(-1))     # alignment path
    # encoder.inference + fc3 path that produced `memory`
    emb_out = O.encoder_forward(sd, torch.from_numpy(g['ids']), torch.tensor([g['ids'].shape[1]]), training=False)
    style = torch.from_numpy(g['z']) @ sd['vae_gst.fc3.weight'].t() + sd['vae_gst.fc3.bias']
    assert (emb_out + style[:, None] - torch.from_numpy(g['memory'])).abs().max() < 1e-4


def test_oracle_mel_frontend_matches_reference(golden_dir):
    import t2v_oracle as O
    g = np.load(os.path.join(golden_dir, 'mel_frontend.npz'))
    basis = O.slaney_mel_basis()
    assert (basis - torch.from_numpy(g['mel_basis'])).abs().max() < 1e-7
    for name in ('speech', 'noise'):
        y = torch.from_numpy(g[name + '_wav'].astype(np.float32) / 32768.0)[None]
        mel = O.mel_spectrogram(y)[0]
        ref = torch.from_numpy(g[name + '_mel'])
        assert mel.shape == ref.shape
        assert (mel - ref).abs().mean() < 1e-5 and (mel - ref).abs().max() < 2e-3, name


def test_oracle_two_adam_steps(model_sd, golden_dir, digests):
    """clip_grad_norm_ + Adam trajectory: two full steps of the oracle land on the reference's weights."""
    import t2v_oracle as O
    _, _, sd0 = model_sd
    g = np.load(os.path.join(golden_dir, 'train_step.npz'))
    names = [k for k, v in sd0.items() if v.dtype == torch.float32 and 'running_' not in k]
    sd = {k: v.clone() for k, v in sd0.items()}
    for k in names:
        sd[k].requires_grad_(True)
    state = {}
    text, lin = torch.from_numpy(g['text']), torch.from_numpy(g['input_lengths'])
    mel, gate, lout = torch.from_numpy(g['mel']), torch.from_numpy(g['gate']), torch.from_numpy(g['output_lengths'])
    for it in range(2):
        out = O.tacotron2_forward(sd, text, lin, mel, lout, True, torch.from_numpy(g['eps']))
        loss = O.loss_forward(out, mel, gate, it, 'constant')[0]
        grads = torch.autograd.grad(loss, [sd[k] for k in names], allow_unused=True)
        live = [(k, gr) for k, gr in zip(names, grads) if gr is not None]
        clipped, total = O.clip_grad_norm([gr for _, gr in live], 1.0)
        assert abs(float(total) - g['grad_norms'][it]) < 2e-3 * g['grad_norms'][it]
        assert abs(float(loss) - g['losses'][it]) < 2e-3 * abs(g['losses'][it])
        with torch.no_grad():
            for (k, _), gr in zip(live, clipped):
                m, v = state.get(k, (torch.zeros_like(gr), torch.zeros_like(gr)))
                p, m, v = O.adam_step(sd[k], gr, m, v, it + 1)
                sd[k].copy_(p)
                state[k] = (m, v)
    gd = digests['grads_step0']
    gmax = max(np.sqrt(v['digest'][2]) for v in gd.values())
    for k in names:
        ref = digests['after_2_steps'][k]
        got = _digest(sd[k])
        tol = 2e-4 * ref[1] + 1e-6
        if k in gd and np.sqrt(gd[k]['digest'][2]) < 1e-5 * gmax:
            # conv biases feeding a BatchNorm have an exactly-zero true gradient; Adam turns the fp32
            # noise into ±lr steps whose sign is implementation-defined → bound by 2 steps · lr · numel
            tol += 2 * 1e-3 * sd[k].numel()
        assert abs(got[1] - ref[1]) <= tol, k


def _grad_sample_check(g, grad_of, tol_rel, who):
    """whole-tensor pins: flattened gradient sampled with the stored stride (stride 1 = the full tensor)"""
    names, strides = [str(x) for x in g['grad_sample_names']], [int(x) for x in g['grad_sample_strides']]
    assert len(names) >= 6
    for i, (k, st) in enumerate(zip(names, strides)):
        ref = torch.from_numpy(g['grad_sample_%d' % i]).double()
        got = grad_of(k).detach().cpu().reshape(-1)[::st].double()
        assert got.shape == ref.shape, (who, k)
        scale = float(ref.abs().max())
        assert float((got - ref).abs().max()) < tol_rel * scale + 1e-9, (who, k, float((got - ref).abs().max()), scale)
        # a sign flip or a permutation of the tensor cannot hide behind a norm: correlation with the reference
        assert float((got * ref).sum()) > 0.999 * float((ref * ref).sum()), (who, k)


def test_oracle_full_gradient_pins(oracle_step):
    """six key parameters, sampled over the WHOLE tensor, against the reference's gradients"""
    g, sd, _, _ = oracle_step
    _grad_sample_check(g, lambda k: sd[k].grad, 2e-3, 'oracle')


@pytest.fixture(scope='module')
def oracle_long_step(model_sd, golden_dir):
    import t2v_oracle as O
    _, _, sd0 = model_sd
    g = np.load(os.path.join(golden_dir, 'train_step_long.npz'))
    sd = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and 'running_' not in k else v.clone())
          for k, v in sd0.items()}
    text, lin = torch.from_numpy(g['text']), torch.from_numpy(g['input_lengths'])
    mel, gate, lout = torch.from_numpy(g['mel']), torch.from_numpy(g['gate']), torch.from_numpy(g['output_lengths'])
    out = O.tacotron2_forward(sd, text, lin, mel, lout, True, torch.from_numpy(g['eps']))
    loss, recon, kl, w = O.loss_forward(out, mel, gate, 0, 'constant')
    loss.backward()
    return g, sd, out, loss


def test_oracle_long_text_step_matches_reference(oracle_long_step):
    """T_in = 300 (> 256: koemo reaches 555 symbols; the reference's attention is unbounded, model.py:67-88)"""
    g, sd, out, loss = oracle_long_step
    assert g['text'].shape == (2, 300)
    for i, name in enumerate(['out_mel', 'out_post', 'out_gate', 'out_align']):
        assert (out[i].detach() - torch.from_numpy(g[name])).abs().max().item() < 5e-5, name
    assert abs(float(loss) - g['scalars'][0]) < 1e-4 * abs(g['scalars'][0])
    names = [str(x) for x in g['grad_names']]
    gmax = float(g['grad_norms'].max())
    for k, ref in zip(names, g['grad_norms']):
        assert abs(float(sd[k].grad.norm()) - float(ref)) < 2e-3 * max(float(ref), 1e-4 * gmax) + 1e-7, k
    _grad_sample_check(g, lambda k: sd[k].grad, 2e-3, 'oracle-long')


@pytest.mark.parametrize("case", ['plain', 'lively'])
def test_oracle_gate_terminated_inference(model_sd, golden_dir, case):
    """BASELINE configs[3] size (200 symbols): the run ends because the reference's own stop rule fired
    (model.py:453), not because max_decoder_steps was reached."""
    import t2v_oracle as O
    _, _, sd0 = model_sd
    g = np.load(os.path.join(golden_dir, 'inference_gate_stop.npz'))
    sd = {k: v.clone() for k, v in sd0.items()}
    fac = float(g[case + '_hh_scale'][0])
    sd['decoder.attention_rnn.weight_hh'] *= fac
    sd['decoder.decoder_rnn.weight_hh'] *= fac
    sd['decoder.gate_layer.linear_layer.bias'] = torch.from_numpy(g[case + '_gate_bias']).clone()
    ids = torch.from_numpy(g['ids'])
    with torch.no_grad():
        memory = O.encoder_forward(sd, ids, torch.tensor([ids.shape[1]]), training=False)
        memory = memory + (torch.from_numpy(g['z']) @ sd['vae_gst.fc3.weight'].t() + sd['vae_gst.fc3.bias'])[:, None]
        mel, gate, al = O.decoder_inference(sd, memory, max_steps=400)
    n = int(g[case + '_n_frames'][0])
    assert mel.shape[2] == n and n < 400                                  # stopped on the gate, at the same frame
    tol = 1e-4 if case == 'plain' else 5e-3                               # 'lively': x6 recurrent weights amplify round-off
    assert (mel - torch.from_numpy(g[case + '_mel'])).abs().max() < tol
    assert torch.equal(al.argmax(-1).to(torch.int16), torch.from_numpy(g[case + '_align_argmax']))
    assert (al.max(-1).values - torch.from_numpy(g[case + '_align_max'])).abs().max() < tol


@pytest.mark.parametrize("case", ['ratios', 'ref_audio'])
def test_oracle_synthesize_call_sequence(model_sd, golden_dir, case):
    """the reference's `Synthesizer.synthesize` statements (synthesizer.py:112-160; fixture (h) of
    oracle/gen_golden.py, produced by the real reference) restated with the oracle: emotion-ratio mix of the centroids through
    fc3 (run ends at max_decoder_steps = 24) and reference-audio conditioning (the stop rule ends the run)."""
    import t2v_oracle as O
    _, _, sd0 = model_sd
    g = np.load(os.path.join(golden_dir, 'synthesize.npz'))
    sd = {k: v.clone() for k, v in sd0.items()}
    sd['decoder.gate_layer.linear_layer.bias'] = torch.from_numpy(g[case + '_gate_bias']).clone()
    ids = torch.from_numpy(g['ids'])
    with torch.no_grad():
        enc = O.encoder_forward(sd, ids, torch.tensor([ids.shape[1]]), training=False)
        if case == 'ref_audio':
            mel_ref = O.mel_spectrogram(torch.from_numpy(g['ref_wav'].astype(np.float32) / 32768.0)[None])
            style, _, _, _ = O.vae_gst_forward(sd, mel_ref, training=False)
            memory = enc + style[:, None]
        else:
            zs, em, r = g['zs'], g['emotions'], g['ratios']
            cent = [zs[em == i].mean(0) for i in range(4)]                     # neu, sad, ang, hap
            mix = r[0] * cent[0] + r[1] * cent[1] + r[2] * cent[3] + r[3] * cent[2]      # call order: (neu, sad, hap, ang)
            memory = enc + (torch.FloatTensor(mix) @ sd['vae_gst.fc3.weight'].t() + sd['vae_gst.fc3.bias'])
        mel, gate, al = O.decoder_inference(sd, memory, max_steps=24)
        post = mel + O.postnet_forward(sd, mel, training=False)
    want = torch.from_numpy(g[case + '_post'])
    assert post.shape == want.shape and (case == 'ratios') == (want.shape[2] == 24)
    assert (post - want).abs().max() < 2e-4
    assert (al - torch.from_numpy(g[case + '_align'])).abs().max() < 1e-5


def test_koemo_text_front_end_hash(golden_dir):
    """every unique sentence of the koemo filelists through OUR text front end, one SHA-256 against the reference's
    (needs the read-only reference's filelists: data, present in the build container only)"""
    import hashlib
    from text import text_to_sequence
    ref_dir = os.environ.get('T2V_REFERENCE', '/root/reference')
    names = ['koemo_spk_emo_all_%s.txt' % s for s in ('train', 'valid', 'test')]
    if not all(os.path.isfile(os.path.join(ref_dir, 'filelists', n)) for n in names):
        pytest.skip("reference filelists not available here")
    with open(os.path.join(golden_dir, 'koemo_ids_sha256.json')) as f:
        gold = json.load(f)
    sents = []
    for n in names:
        with open(os.path.join(ref_dir, 'filelists', n), encoding='utf-8') as f:
            sents += [ln.strip().split('|')[1] for ln in f if ln.strip()]
    uniq = sorted(set(sents))
    assert len(uniq) == gold['unique_sentences'] and gold['skipped_need_nltk'] == 0
    h = hashlib.sha256()
    longest = 0
    for t in uniq:
        ids = text_to_sequence(t, ['korean_cleaners'])
        longest = max(longest, len(ids))
        h.update((t + '\t' + ','.join(str(i) for i in ids) + '\n').encode('utf-8'))
    assert longest == gold['max_symbols'] == 555
    assert h.hexdigest() == gold['sha256']
