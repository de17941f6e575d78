"""The REAL training engine with two data-parallel ranks (reference distributed.py:126-174, train.py:31-50,59-60).

Two processes share the one GPU of the test box; the process group is gloo (RCCL refuses two ranks on one device), whose
collectives on device tensors are staged through host memory by `distributed.all_reduce_sum / broadcast` — the SEMANTICS
of the exchange are what is pinned here, the transport is RCCL in production (`bench.py --gpus N`, 1-rank RCCL run in
test_train_plumbing_gpu.py).  Each rank runs `TrainEngine(world_size=2)` on its own 3-utterance shard, in both engines:

  eager  : bucketed all-reduce issued from the backward hooks (OverlappedArenaAllReduce), then fused clip + Adam
  graph  : forward + backward replay as one HIP graph, ONE all-reduce of the arena, fused clip + Adam

and checks what data parallelism promises:
  * `load_model` broadcast: all 142 state tensors equal rank 0's although the ranks were seeded differently;
  * the gradient arena after the exchange == g_rank0 + g_rank1 of two single-process backward passes (per-rank
    BatchNorm statistics and a per-rank KL *sum*), to fp32 round-off;
  * the update is Adam on the MEAN gradient (1/world folded into the fused clip + Adam kernel);
  * weights stay bit-identical on the two ranks over several steps, dead parameters are never touched;
  * ranks that see DIFFERENT batch shapes (one replays its captured graph, the other runs eagerly) issue the same
    collectives and stay in lock-step (mismatched collective patterns);
  * bf16_run exchanges bf16 gradients (half the bytes) and stays in lock-step.
"""
import os
import socket
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

T_IN, T_OUT = 20, 36


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, out_dir):
    for p in (os.path.join(ROOT, 'tacotron2-vae_amd'), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    # T2V_TRAIN_PERSISTENT=0: the two ranks of this test share ONE GPU; the persistent decoder forward needs all 256 CUs
    # to itself (its workgroups spin on each other), so two of them launched by two processes can starve each other
    # until the bounded spins give up.  One process per GPU — the production layout — has no such neighbour.
    # T2V_GRAPH_WATCHDOG=0: for the same reason step times here are noise (50 .. 60 ms with the neighbour's kernels in between), and a
    # replay that happens to lose to one eager step by 10 % must not make the engine under test drop its graph
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK='0', HSA_ENABLE_IPC_MODE_LEGACY='0', T2V_TRAIN_PERSISTENT='0', T2V_GRAPH_WATCHDOG='0')
    import torch.distributed as dist
    import distributed as D
    import hparams as HP
    import model as M
    import t2v_hip
    import train as TR
    from bench import synthetic_batch
    torch.cuda.set_device(0)
    D.init_distributed(backend='gloo', timeout_s=300)
    graph = mode in ('graph', 'ragged', 'poison_graph')
    bf16 = mode == 'bf16'
    extra = ',bf16_run=True,fp32_allreduce=False' if bf16 else ''      # (the bf16 wire format is opt-in since round 4)
    hp_ref = HP.create_hparams("batch_size=3,anneal_function=constant" + extra)
    hp = HP.create_hparams("batch_size=3,anneal_function=constant,distributed_run=True" + extra)

    # a single-process engine of the same weights gives this rank's own gradient (no exchange)
    torch.manual_seed(hp.seed)
    ref = TR.TrainEngine(hp_ref, world_size=1, graph=False)
    init_state = {k: v.detach().clone() for k, v in ref.model.state_dict().items()}

    torch.manual_seed(hp.seed + 17 * rank)          # ranks start from DIFFERENT weights: the broadcast must fix that
    eng = TR.TrainEngine(hp, world_size=world, graph=graph)
    res = {'rank': rank}
    sd = eng.model.state_dict()
    res['n_state'] = len(sd)
    res['bcast_equal_rank0_seed'] = all(torch.equal(sd[k].cpu(), init_state[k].cpu()) for k in sd) if rank == 0 else None
    flat_state = torch.cat([v.detach().reshape(-1).double() for v in sd.values()])
    g0 = flat_state.cpu().clone()
    D.broadcast(g0, 0)
    res['bcast_same_as_rank0'] = bool(torch.equal(g0, flat_state.cpu()))
    ref.model.load_state_dict({k: v.clone() for k, v in sd.items()})
    # the reparameterisation noise is the one torch-RNG draw of a step: fixed, so that the single-process engine and the
    # DP engine of this rank compute the same local gradient (dropout masks are counter-based: equal by construction)
    eps = torch.randn(3, 32, generator=torch.Generator().manual_seed(5 + rank)).cuda()
    ref.model.vae_gst.eps_override = eps
    eng.model.vae_gst.eps_override = eps

    def batch_for(step):
        t_out = T_OUT
        if mode == 'ragged' and rank == 1:
            t_out = T_OUT - 4 * (step % 3)          # rank 1 never sees a shape three times in a row: it stays eager
        lens_in = [T_IN, T_IN - 3, T_IN - 7]
        lens_out = [t_out, t_out - 5, t_out - 9]
        return synthetic_batch(3, T_IN, t_out, 100 + 10 * rank + step, lens_in=lens_in, lens_out=lens_out)

    dead = {n: p.detach().clone() for n, p in eng.model.named_parameters()
            if n.startswith(('speaker_embedding.', 'emotion_embedding.')) or n.startswith('vae_gst.ref_encoder.convs.0.weight')
            or n.startswith('vae_gst.ref_encoder.convs.0.bias')}

    # ---- step 0: own gradient (single-process engine), then the exchanged arena of the DP engine
    b0 = batch_for(0)
    x, y = ref.model.parse_batch(b0)
    ref._publish(0)
    ref._body_fb(x, y, 0)
    g_local = ref.optimizer.grads.detach().clone()
    p_before = eng.optimizer.params.detach().clone()
    eng.step(b0, 0)
    torch.cuda.synchronize()
    g_sum = eng.optimizer.grads.detach().clone()
    parts = []
    for r in range(world):
        t = g_local.cpu().clone()
        D.broadcast(t, r)
        parts.append(t)
    want = parts[0] + parts[1]
    scale = want.abs().max().item()
    tol = 2e-2 if bf16 else 2e-5
    res['grad_sum_err'] = float((g_sum.cpu() - want).abs().max().item() / scale)
    res['grad_sum_ok'] = res['grad_sum_err'] < tol
    # the optimiser stepped on the MEAN gradient: replay clip + Adam on a single-process optimiser fed with want / world
    ref.optimizer.params.copy_(p_before)
    ref.optimizer.exp_avg.zero_(); ref.optimizer.exp_avg_sq.zero_()
    ref.optimizer.step_count = 0
    ref.optimizer.grads.copy_((g_sum / world))
    ref.optimizer._no_grad = []
    ref.optimizer.mark_gathered()
    t2v_hip.activate_step_params(ref.step_params)
    ref.optimizer.step()
    torch.cuda.synchronize()
    res['adam_mean_err'] = float((ref.optimizer.params - eng.optimizer.params).abs().max().item())
    res['grad_norm_world'] = float(eng.optimizer.grad_norm.item())
    res['grad_norm_mean'] = float(ref.optimizer.grad_norm.item())

    # ---- more steps (graph mode captures on the third identical shape): ranks stay in lock-step
    n_steps = 6 if graph else 3
    losses = []
    for it in range(1, n_steps):
        out = eng.step(batch_for(it), it)
        losses.append(float(out[0].item()))
    torch.cuda.synchronize()
    t2v_hip.check_async_errors()
    if mode.startswith('poison'):
        # ---- a persistent-kernel time-out on rank 1 ONLY: the flag must reach rank 0
        # through the poison slot, BOTH ranks must skip the update on the device, both must re-run the same iteration with
        # the same collectives, and the weights must stay in lock-step.  (Two ranks share this GPU, so the persistent kernels
        # are off: the injected word is filed under their label, which is what recover() keys on.)
        PLABEL = 'decoder forward (persistent kernel hand-off)'
        res['n_graphs_before'] = len(eng._graphs)
        p_pre = eng.optimizer.params.detach().clone()
        count_pre = eng.optimizer.step_count
        it = n_steps
        if rank == 1:
            if graph:       # the step is a REPLAY: no host code runs that could be intercepted — set the graph's ledger word
                def after_replay(span):
                    _, pool = t2v_hip._err_pool()
                    t2v_hip.err_words(span)[:1].fill_(1)
                    pool.labels[span[0]] = PLABEL
                    eng.__dict__.pop('_test_after_replay', None)
                eng._test_after_replay = after_replay
            else:
                t2v_hip._ERR_INJECT[0] = ('decoder forward', PLABEL)
        out = eng.step_checked(batch_for(it), it)
        res['poison_loss'] = float(out[0].item())
        res['recoveries'] = eng.recoveries
        res['steps_applied'] = eng.optimizer.step_count - count_pre          # skipped + re-run = ONE optimiser step
        res['moved_by_rerun'] = float((eng.optimizer.params - p_pre).abs().max().item())
        res['n_graphs_after'] = len(eng._graphs)
        res['slot_after'] = float(eng.optimizer.poison_slot().abs().max().item())
        # the next steps run clean: nothing stays poisoned, nobody recovers again
        for it in range(n_steps + 1, n_steps + 3):
            out = eng.step_checked(batch_for(it), it)
            losses.append(float(out[0].item()))
        res['recoveries_end'] = eng.recoveries
        res['steps_total'] = eng.optimizer.step_count
        res['slot_end'] = float(eng.optimizer.poison_slot().abs().max().item())
    res['losses'] = losses
    res['n_graphs'] = len(eng._graphs)
    mine = eng.optimizer.params.detach().cpu()
    other = mine.clone()
    D.broadcast(other, 0)
    res['weights_equal_across_ranks'] = bool(torch.equal(mine, other))
    res['moved'] = float((mine - p_before.cpu()).abs().max().item())
    cur = dict(eng.model.named_parameters())
    res['dead_untouched'] = all(torch.equal(cur[n].detach(), v) for n, v in dead.items()) and len(dead) >= 4
    res['wire_bytes'] = eng.allreduce.wire_bytes()
    res['arena_bytes'] = eng.optimizer.grads_for_allreduce().numel() * 4      # gradients + the 16-byte poison slot
    res['bucket_log'] = list(eng.allreduce.launch_log)
    res['params_digest'] = float(mine.double().sum().item())
    torch.save(res, os.path.join(out_dir, 'r%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def _run(mode, tmp_path):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, mode, str(tmp_path)), nprocs=world, join=True)
    return [torch.load(os.path.join(str(tmp_path), 'r%d.pt' % r), weights_only=False) for r in range(world)]


def _common_checks(rs, bf16=False):
    for r in rs:
        assert r['n_state'] == 142
        assert r['bcast_same_as_rank0'], "load_model() must broadcast rank 0's state (distributed.py:132-135)"
        assert r['grad_sum_ok'], r['grad_sum_err']
        assert r['adam_mean_err'] < (1e-4 if bf16 else 2e-6), r['adam_mean_err']
        assert r['weights_equal_across_ranks']
        assert r['dead_untouched']
        assert r['moved'] > 1e-4
        assert all(l == l and abs(l) < 1e4 for l in r['losses'])
    assert rs[0]['bcast_equal_rank0_seed']
    assert rs[0]['params_digest'] == rs[1]['params_digest']


@pytest.mark.gpu
def test_two_ranks_eager_bucketed_allreduce(tmp_path):
    rs = _run('eager', tmp_path)
    _common_checks(rs)
    for r in rs:
        assert r['n_graphs'] == 0
        assert len(r['bucket_log']) == 4 and any(h for _, h in r['bucket_log'])      # buckets issued from backward hooks
        assert r['wire_bytes'] == r['arena_bytes'] > 100e6


@pytest.mark.gpu
def test_two_ranks_graph_engine_single_allreduce(tmp_path):
    rs = _run('graph', tmp_path)
    _common_checks(rs)
    for r in rs:
        assert r['n_graphs'] == 1            # the shape was captured (third time it was seen) and replayed
        assert r['bucket_log'] == []         # the graph engine never uses the hook-issued buckets, warm-up steps included


@pytest.mark.gpu
def test_two_ranks_with_different_shapes_do_not_mismatch_collectives(tmp_path):
    rs = _run('ragged', tmp_path)
    _common_checks(rs)
    assert rs[0]['n_graphs'] == 1 and rs[1]['n_graphs'] == 0      # rank 0 replays while rank 1 runs eagerly


@pytest.mark.gpu
def test_two_ranks_bf16_run_exchanges_bf16_gradients(tmp_path):
    rs = _run('bf16', tmp_path)
    _common_checks(rs, bf16=True)
    for r in rs:
        assert r['wire_bytes'] * 2 == r['arena_bytes']


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['poison_eager', 'poison_graph'])
def test_timeout_on_one_rank_skips_and_reruns_the_step_on_both(tmp_path, mode):
    """ADVICE r4 (high / medium), VERDICT r4 item 7c.  Eager engine: the poison slot is reduced after backward (it used to leave
    with a hook-issued bucket, stale).  Graph engine: the flag is read from the replayed graph's OWN ledger block (a replay
    leaves no eager notes, the slot used to be zeroed every step and the bad update was applied on every rank)."""
    rs = _run(mode, tmp_path)
    _common_checks(rs)
    for r in rs:
        assert r['recoveries'] == 1 and r['recoveries_end'] == 1, (r['recoveries'], r['recoveries_end'])
        assert r['steps_applied'] == 1           # the skipped update was never applied, the re-run once
        assert r['moved_by_rerun'] > 1e-5
        assert r['poison_loss'] == r['poison_loss']
        assert r['slot_end'] == 0.0
    if mode == 'poison_graph':
        for r in rs:
            assert r['n_graphs_before'] == 1 and r['n_graphs_after'] == 0      # recover() drops the captured graphs
    assert rs[0]['steps_total'] == rs[1]['steps_total']


@pytest.mark.gpu
@pytest.mark.parametrize('graph', [True, False], ids=['graph_engine', 'eager_engine'])
def test_bench_entry_with_two_ranks_prints_the_multi_gpu_line(graph):
    """VERDICT r5 next 7 (reference distributed.py:137-162, train.py:31-50, multiproc.py:1-23): `python bench.py --gpus 2` — the
    command the driver runs for the scaling curve — self-launches two ranks under torch.distributed.run, and rank 0 prints ONE JSON
    line whose multi-rank keys parse.  The two ranks share the one GPU of the test box (T2V_BENCH_SHARE_GPU=1) over gloo
    (T2V_BENCH_BACKEND=gloo: RCCL refuses two ranks on one device) on the launch-per-step kernels, like the tests above; what is
    pinned is the PLUMBING of the N > 1 line — rendezvous, barrier + max over ranks, weak-scaling frame count, per-rank step
    times / step modes, the exposed all-reduce time — so that the day a node exists the SCALE line does not fail on it."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.update(HSA_ENABLE_IPC_MODE_LEGACY='0', T2V_BENCH_SHARE_GPU='1', T2V_BENCH_BACKEND='gloo', T2V_TRAIN_PERSISTENT='0',
               T2V_GRAPH_WATCHDOG='0')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--settle', '1',
           '--no-cpu-baseline', '--no-decode', '--no-secondary', '--eager-steps', '0'] + ([] if graph else ['--no-graph'])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, "rank 0 alone prints the line"
    d = json.loads(lines[0])
    assert list(d)[:7] == ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step"]
    assert d['n_gpus'] == 2 and d['rccl_ranks'] == 2 and d['steps'] == 3 and d['warmup'] == 1
    assert d['scaling'] == 'weak' and d['config']['parallelism'] == 'dp2' and d['config']['global_batch'] == 12
    assert d['config']['frames_per_step'] == 2 * 6 * 400
    assert abs(d['value'] - d['config']['frames_per_step'] / (d['ms_per_step'] * 1e-3)) < 1e-3 * d['value']
    per = d['ms_per_step_ranks']['per_rank']
    assert len(per) == 2 and d['ms_per_step_ranks']['max'] == max(per) and abs(d['ms_per_step'] - max(per)) < 1e-2
    assert d['allreduce_exposed_ms'] >= 0.0
    assert sum(b[1] for b in d['allreduce_buckets_bytes']) > 100e6
    ranks = d['ranks']
    assert [r['rank'] for r in ranks] == [0, 1] and d['ranks_disagree'] is False
    for r in ranks:
        assert r['decoder_forward'] == 'launch-per-step' and r['decoder_backward'] == 'launch-per-step' and r['recoveries'] == 0
        assert r['step_mode'].startswith('hip-graph replay of forward + backward' if graph else 'eager launches')
    assert d['final_loss'] == d['final_loss']        # finite
