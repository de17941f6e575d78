"""GPU parity of the whole train step against the reference's golden vectors (tests/golden, written
by oracle/gen_golden.py from the real reference): forward outputs, loss, raw gradients, two
clip+Adam steps through the fused HIP optimiser, checkpoint interchange."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _digest(t):
    t = t.detach().double().cpu()
    return [float(t.sum()), float(t.abs().sum()), float((t * t).sum())]


@pytest.fixture(scope='module')
def run(golden_dir):
    import hparams as HP
    import model as M
    import train as TR
    g = np.load(os.path.join(golden_dir, 'train_step.npz'))
    with open(os.path.join(golden_dir, 'train_step_digests.json')) as f:
        dig = json.load(f)
    hp = HP.create_hparams("anneal_function=constant,p_attention_dropout=0.0,p_decoder_dropout=0.0")
    old = M.drop_rate
    M.drop_rate = 0.0
    torch.manual_seed(hp.seed)
    eng = TR.TrainEngine(hp)
    eng.model.vae_gst.eps_override = torch.from_numpy(g['eps']).cuda()
    batch = (torch.from_numpy(g['text']), torch.from_numpy(g['input_lengths']), torch.from_numpy(g['mel']),
             torch.from_numpy(g['gate']), torch.from_numpy(g['output_lengths']),
             torch.zeros(2, 1, dtype=torch.long), torch.from_numpy(g['emotions']))
    # step 0 by hand (to look at outputs and raw gradients), then the optimiser
    opt = eng.optimizer
    opt.zero_grad()
    x, y = eng.model.parse_batch(batch)
    y_pred = eng.model(x)
    loss, recon, kl, w = eng.criterion(y_pred, y, 0)
    loss.backward()
    grads = {n: p.grad.detach().clone().cpu() for n, p in eng.model.named_parameters() if p.grad is not None}
    gn0 = float(opt.step().item())
    l1, _, _, _, gn1 = eng.step(batch, 1)
    torch.cuda.synchronize()
    res = dict(g=g, dig=dig, eng=eng, y_pred=[t.detach().cpu() for t in y_pred[:7]], loss=float(loss), kl=float(kl),
               recon=float(recon), grads=grads, gn=[gn0, float(gn1.item())], l1=float(l1.item()), batch=batch, hp=hp)
    yield res
    M.drop_rate = old


def test_forward_outputs_match_reference(run):
    g = run['g']
    tol = dict(out_mel=1e-4, out_post=2e-4, out_gate=1e-4, out_align=2e-5, out_mu=1e-5, out_logvar=1e-5, out_z=1e-5)
    for i, name in enumerate(['out_mel', 'out_post', 'out_gate', 'out_align', 'out_mu', 'out_logvar', 'out_z']):
        ref = torch.from_numpy(g[name])
        d = (run['y_pred'][i] - ref).abs()
        assert d.max().item() < tol[name], (name, d.max().item())
    # BASELINE.json: mel-L1 vs the reference CPU path < 1e-4
    assert (run['y_pred'][0] - torch.from_numpy(g['out_mel'])).abs().mean().item() < 1e-4
    assert (run['y_pred'][1] - torch.from_numpy(g['out_post'])).abs().mean().item() < 1e-4
    assert abs(run['loss'] - g['scalars'][0]) < 1e-4 * abs(g['scalars'][0])
    assert abs(run['kl'] - g['scalars'][2]) < 1e-4 * abs(g['scalars'][2])


def test_gradients_match_reference(run):
    gd = run['dig']['grads_step0']
    gmax = max(np.sqrt(v['digest'][2]) for v in gd.values())
    assert set(gd.keys()) <= set(run['grads'].keys())
    for k, v in gd.items():
        gr = run['grads'][k]
        scale = max(np.sqrt(v['digest'][2]), 1e-4 * gmax)
        head = gr.reshape(-1)[:8].double().numpy()
        assert np.abs(head - np.array(v['head'])).max() < 2e-3 * scale + 1e-7, k
        assert abs(np.sqrt(float((gr.double() ** 2).sum())) - np.sqrt(v['digest'][2])) < 2e-3 * scale + 1e-7, k
    # tensors the reference never touches stay untouched here too (Appendix B-7)
    for k in run['dig']['no_grad_params']:
        assert k not in run['grads'] or float(run['grads'][k].abs().max()) == 0.0, k


def test_two_optimizer_steps_match_reference(run):
    g, dig = run['g'], run['dig']
    assert abs(run['gn'][0] - g['grad_norms'][0]) < 2e-3 * g['grad_norms'][0]
    assert abs(run['gn'][1] - g['grad_norms'][1]) < 5e-3 * g['grad_norms'][1]
    assert abs(run['l1'] - g['losses'][1]) < 2e-3 * abs(g['losses'][1])
    gd = dig['grads_step0']
    gmax = max(np.sqrt(v['digest'][2]) for v in gd.values())
    sd = run['eng'].model.state_dict()
    for k, ref in dig['after_2_steps'].items():
        if not sd[k].dtype.is_floating_point or 'running_' in k:
            continue
        # Adam's first steps move every element by ~lr whatever the gradient magnitude, so an element whose
        # gradient is at the fp32 noise floor can land 2*lr away; 3e-4 of the digest covers a handful of those
        tol = 3e-4 * ref[1] + 1e-6
        if k in gd and np.sqrt(gd[k]['digest'][2]) < 1e-5 * gmax:
            tol += 2 * 1e-3 * sd[k].numel()        # zero-gradient conv biases: Adam amplifies fp32 noise
        assert abs(_digest(sd[k])[1] - ref[1]) <= tol, k
    for k in dig['no_grad_params']:                 # dead tensors keep their initial value
        assert _digest(sd[k]) == dig['init'][k], k
    # BatchNorm buffers: 2 updates with momentum 0.1
    assert int(sd['postnet.convolutions.0.1.num_batches_tracked']) == 2


def test_checkpoint_interchange(run, golden_dir, tmp_path):
    """save_checkpoint writes the reference's dict; torch.optim.Adam can load our optimizer state and
    FlatAdam can load it back."""
    import train as TR
    with open(os.path.join(golden_dir, 'checkpoint_schema.json')) as f:
        sch = json.load(f)
    eng = run['eng']
    path = str(tmp_path / 'checkpoint_1')
    TR.save_checkpoint(eng.model, eng.optimizer, 1e-3, 1, path)
    ck = torch.load(path, map_location='cpu', weights_only=False)
    assert list(ck.keys()) == sch['top_keys']
    assert [[k, list(v.shape), str(v.dtype).replace('torch.', '')] for k, v in ck['state_dict'].items()] == sch['state_dict']
    osd = ck['optimizer']
    assert sorted(osd['state'].keys()) == sch['optimizer_state_indices']
    assert sorted(next(iter(osd['state'].values())).keys()) == sch['optimizer_state_keys']
    assert len(osd['param_groups'][0]['params']) == sch['n_parameters']
    # a stock torch Adam over a reference-layout model accepts it
    import hparams as HP
    import model as M
    m2 = M.Tacotron2(HP.create_hparams())
    m2.load_state_dict(ck['state_dict'])
    adam = torch.optim.Adam(m2.parameters(), lr=1e-3, weight_decay=1e-6)
    adam.load_state_dict(osd)
    # and our optimiser round-trips its own file
    before = eng.optimizer.exp_avg.clone()
    eng.optimizer.exp_avg.zero_()
    TR.load_checkpoint(path, eng.model, eng.optimizer)
    assert torch.equal(eng.optimizer.exp_avg, before) and eng.optimizer.step_count == 2


def test_fused_adam_matches_oracle_formula():
    """k_sumsq + k_clip_adam vs the oracle's clip_grad_norm/adam_step on random tensors."""
    import ctypes as C
    import t2v_hip
    import t2v_oracle as O
    lib = t2v_hip.load_library()
    n = 1000003
    g = torch.Generator().manual_seed(0)
    p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g) * 0.01
    m, v = torch.randn(n, generator=g) * 0.01, torch.rand(n, generator=g) * 1e-4
    clipped, total = O.clip_grad_norm([gr], 1.0)
    p_ref, m_ref, v_ref = O.adam_step(p, clipped[0], m, v, 3)
    dp, dg, dm, dv = (t.cuda() for t in (p, gr, m, v))
    part, norm = torch.zeros(1024, device='cuda'), torch.zeros(1, device='cuda')
    rc = lib.t2v_clip_adam_step(C.c_void_p(dp.data_ptr()), C.c_void_p(dg.data_ptr()), C.c_void_p(dm.data_ptr()),
                                C.c_void_p(dv.data_ptr()), n, 1e-3, 0.9, 0.999, 1e-8, 1e-6, 1.0, 1.0, 1.0 - 0.9 ** 3, 1.0 - 0.999 ** 3,
                                C.c_void_p(part.data_ptr()), C.c_void_p(norm.data_ptr()),
                                C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert abs(float(norm) - float(total)) < 1e-5 * float(total)
    assert (dp.cpu() - p_ref).abs().max() < 2e-6
    assert (dm.cpu() - m_ref).abs().max() < 1e-7 and (dv.cpu() - v_ref).abs().max() < 1e-9


@pytest.mark.parametrize('graph', [True, False], ids=['graph_engine', 'eager_engine'])
def test_branch_overlap_is_bit_identical_to_single_stream(golden_dir, graph):
    """The engine's multi-stream schedule (t2v_hip.Overlap: encoder ‖ reference encoder ‖ Prenet -> gpre in the forward,
    every weight gradient on the deferred-work streams in the backward) vs the single-stream schedule: same bits in
    the outputs and in every gradient, in the graph engine (shadow leaves + autograd.grad) and in the eager engine
    (loss.backward(): AccumulateGrad must adopt the deferred gradients without touching them) — also a race detector
    for the cross-stream hand-offs."""
    import hparams as HP
    import model as M
    import train as TR
    g = np.load(os.path.join(golden_dir, 'train_step.npz'))
    old = M.drop_rate
    M.drop_rate = 0.0
    res = {}
    try:
        for mode in (True, False, True):
            M.Tacotron2.overlap_branches = mode
            hp = HP.create_hparams("anneal_function=constant,p_attention_dropout=0.0,p_decoder_dropout=0.0")
            torch.manual_seed(hp.seed)
            eng = TR.TrainEngine(hp, graph=graph)
            eng.model.vae_gst.eps_override = torch.from_numpy(g['eps']).cuda()
            batch = (torch.from_numpy(g['text']), torch.from_numpy(g['input_lengths']), torch.from_numpy(g['mel']),
                     torch.from_numpy(g['gate']), torch.from_numpy(g['output_lengths']),
                     torch.zeros(2, 1, dtype=torch.long), torch.from_numpy(g['emotions']))
            for it in range(3):
                out = eng.step(batch, it)
            torch.cuda.synchronize()
            cur = (float(out[0]), eng.optimizer.grads.clone(), eng.optimizer.params.clone())
            if mode in res:
                assert cur[0] == res[mode][0] and torch.equal(cur[1], res[mode][1])
            res[mode] = cur
        assert res[True][0] == res[False][0]
        assert torch.equal(res[True][1], res[False][1]) and torch.equal(res[True][2], res[False][2])
    finally:
        M.drop_rate = old
        M.Tacotron2.overlap_branches = True


def test_koemo_shape_parity_against_oracle():
    """BASELINE.json configs[1] at its full size with the koemo length profile of SURVEY 8(d) — (T_in,T_out) =
    (84,400),(80,380),(71,350),(66,300),(50,260),(37,200): HIP forward + loss + backward against the CPU oracle
    (the validated restatement of the reference) on the same weights, dropout off, same epsilon."""
    import sys
    import hparams as HP
    import model as M
    import t2v_oracle as O
    import train as TR
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import synthetic_batch
    lens_in, lens_out = [84, 80, 71, 66, 50, 37], [400, 380, 350, 300, 260, 200]
    old = M.drop_rate
    M.drop_rate = 0.0
    try:
        hp = HP.create_hparams("batch_size=6,anneal_function=constant,p_attention_dropout=0.0,p_decoder_dropout=0.0")
        torch.manual_seed(hp.seed)
        eng = TR.TrainEngine(hp)
        batch = synthetic_batch(6, 84, 400, 77, lens_in=lens_in, lens_out=lens_out)
        eps = torch.randn(6, hp.z_latent_dim, generator=torch.Generator().manual_seed(5))
        eng.model.vae_gst.eps_override = eps.cuda()
        sd = {k: v.detach().cpu().clone() for k, v in eng.model.state_dict().items()}
        eng.optimizer.zero_grad()
        x, y = eng.model.parse_batch(batch)
        y_pred = eng.model(x)
        loss = eng.criterion(y_pred, y, 0)[0]
        loss.backward()
        torch.cuda.synchronize()
        # ---- oracle on the host (8 threads: the fastest setting for these M=6 GEMVs on the EPYC host)
        nthreads = torch.get_num_threads()
        torch.set_num_threads(8)
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and 'running_' not in k}
        osd = dict(sd)
        osd.update(leaves)
        text, lin, mel, gate, lout = batch[0].long(), batch[1].long(), batch[2].float(), batch[3].float(), batch[4].long()
        o = O.tacotron2_forward(osd, text, lin, mel, lout, training=True, eps=eps)
        o_loss = O.loss_forward(o, mel, gate, 0, anneal_function='constant')[0]
        o_loss.backward()
        torch.set_num_threads(nthreads)
        assert abs(float(loss) - float(o_loss)) < 1e-4 * abs(float(o_loss))
        assert (y_pred[0].detach().cpu() - o[0].detach()).abs().mean().item() < 1e-4          # mel-L1 (BASELINE.json)
        assert (y_pred[1].detach().cpu() - o[1].detach()).abs().mean().item() < 1e-4
        assert (y_pred[3].detach().cpu() - o[3].detach()).abs().max().item() < 5e-5           # alignments
        gmax = max(float(v.grad.norm()) for v in leaves.values() if v.grad is not None)
        checked = 0
        for name, p in eng.model.named_parameters():
            ref = leaves[name].grad if name in leaves else None
            if ref is None or p.grad is None:
                continue
            scale = max(float(ref.norm()), 1e-4 * gmax)
            assert float((p.grad.cpu() - ref).norm()) < 3e-3 * scale, name
            checked += 1
        assert checked >= 90
    finally:
        M.drop_rate = old


def test_graph_replay_equals_eager_training():
    """hparams.graph_step: the iteration captured into a HIP graph (replayed with the device-side step record) follows
    the eager engine bit for bit — losses, gradient norms and the weights after six optimiser steps, dropout ON (the
    masks are functions of the device-side epoch, not of frozen kernel arguments)."""
    import sys
    import hparams as HP
    import train as TR
    import t2v_hip
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from bench import synthetic_batch
    batches = [synthetic_batch(3, 30, 48, 5, lens_in=[30, 22, 17], lens_out=[48, 40, 31]),
               synthetic_batch(3, 30, 48, 6, lens_in=[30, 25, 11], lens_out=[48, 37, 20])]
    runs = {}
    for mode in ('eager', 'graph'):
        hp = HP.create_hparams("batch_size=3,anneal_function=logistic,graph_step=%s" % (mode == 'graph'))
        torch.manual_seed(hp.seed)
        torch.cuda.manual_seed(hp.seed)
        eng = TR.TrainEngine(hp)
        eng.model.overlap_branches = False
        eng.model.vae_gst.eps_override = torch.full((3, 32), 0.25, device='cuda')
        vals = []
        for it in range(6):
            loss, recon, kl, w, gn = eng.step(batches[it % 2], 1000 * it)      # iteration drives KL weight + dropout epoch
            vals.append((float(loss), float(kl), float(w), float(gn)))
        torch.cuda.synchronize()
        t2v_hip.check_async_errors()
        assert (mode == 'graph') == (len(eng._graphs) == 1)
        runs[mode] = (vals, eng.optimizer.params.clone(), eng.optimizer.step_count)
    assert runs['eager'][2] == runs['graph'][2] == 6
    assert runs['eager'][0] == runs['graph'][0], (runs['eager'][0], runs['graph'][0])
    assert torch.equal(runs['eager'][1], runs['graph'][1])
    ws = [v[2] for v in runs['graph'][0]]
    assert ws[0] < ws[3] < ws[5]                 # logistic KL weight reached the replays through the device record
    ls = [v[0] for v in runs['graph'][0]]
    assert len(set(ls)) == 6                     # every replay saw fresh dropout masks / parameters


def _grad_sample_check(g, grads, tol_rel):
    names, strides = [str(x) for x in g['grad_sample_names']], [int(x) for x in g['grad_sample_strides']]
    for i, (k, st) in enumerate(zip(names, strides)):
        ref = torch.from_numpy(g['grad_sample_%d' % i]).double()
        got = grads[k].reshape(-1)[::st].double()
        assert got.shape == ref.shape, k
        scale = float(ref.abs().max())
        assert float((got - ref).abs().max()) < tol_rel * scale + 1e-9, (k, float((got - ref).abs().max()), scale)
        assert float((got * ref).sum()) > 0.999 * float((ref * ref).sum()), k


def test_full_gradient_pins_match_reference(run):
    """six key parameters over the WHOLE tensor (stride samples of the flattened gradient) against the reference:
    location_conv, query_layer, attention_rnn.weight_hh, postnet conv-0 (B-5 quirk), reverse encoder LSTM, ref-enc conv-0"""
    _grad_sample_check(run['g'], run['grads'], 3e-3)


def test_long_text_step_matches_reference(golden_dir):
    """T_in = 300 > 256 through the whole model against the reference's own outputs and gradients
    (fixture written by oracle/gen_golden.py; koemo has 76 training utterances beyond 256 symbols)."""
    import hparams as HP
    import model as M
    import train as TR
    g = np.load(os.path.join(golden_dir, 'train_step_long.npz'))
    hp = HP.create_hparams("anneal_function=constant,p_attention_dropout=0.0,p_decoder_dropout=0.0")
    old = M.drop_rate
    M.drop_rate = 0.0
    try:
        torch.manual_seed(hp.seed)
        eng = TR.TrainEngine(hp)
        eng.model.vae_gst.eps_override = torch.from_numpy(g['eps']).cuda()
        batch = (torch.from_numpy(g['text']), torch.from_numpy(g['input_lengths']), torch.from_numpy(g['mel']),
                 torch.from_numpy(g['gate']), torch.from_numpy(g['output_lengths']),
                 torch.zeros(2, 1, dtype=torch.long), torch.from_numpy(g['emotions']))
        eng.optimizer.zero_grad()
        x, y = eng.model.parse_batch(batch)
        y_pred = eng.model(x)
        loss = eng.criterion(y_pred, y, 0)[0]
        loss.backward()
        torch.cuda.synchronize()
        import t2v_hip
        t2v_hip.check_async_errors()
        tol = dict(out_mel=1e-4, out_post=3e-4, out_gate=1e-4, out_align=2e-5)
        for i, name in enumerate(['out_mel', 'out_post', 'out_gate', 'out_align']):
            d = (y_pred[i].detach().cpu() - torch.from_numpy(g[name])).abs().max().item()
            assert d < tol[name], (name, d)
        assert abs(float(loss) - g['scalars'][0]) < 1e-4 * abs(g['scalars'][0])
        grads = {n: p.grad.detach().cpu() for n, p in eng.model.named_parameters() if p.grad is not None}
        gmax = float(g['grad_norms'].max())
        for k, ref in zip([str(s) for s in g['grad_names']], g['grad_norms']):
            assert abs(float(grads[k].norm()) - float(ref)) < 3e-3 * max(float(ref), 1e-4 * gmax) + 1e-7, k
        _grad_sample_check(g, grads, 3e-3)
    finally:
        M.drop_rate = old
