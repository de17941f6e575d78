"""world_size-2 data-parallel plumbing on CPU (gloo): state broadcast + arena all-reduce, i.e. the
N>1 path of bench.py/train.py minus the kernels."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'tacotron2-vae_amd'))
    import distributed as D
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    D.init_distributed(backend='gloo', timeout_s=60)
    torch.manual_seed(100 + rank)                       # different weights per rank before the broadcast
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.BatchNorm1d(5))
    D.apply_gradient_allreduce(net)
    w_after = torch.cat([t.reshape(-1).double() for t in net.state_dict().values()])
    flat = torch.arange(5000, dtype=torch.float32) * (rank + 1)      # "gradient arena"
    D.ArenaAllReduce(flat, n_chunks=3)()
    loss = D.reduce_tensor(torch.tensor([float(rank)]), world)
    if rank == 0:
        torch.save(dict(w=w_after, flat=flat, loss=loss), out)
    gathered = [torch.zeros_like(w_after) for _ in range(world)]
    dist.all_gather(gathered, w_after)
    assert all(torch.equal(g, gathered[0]) for g in gathered)
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_arena_allreduce(tmp_path):
    world, port = 2, _free_port()
    out = str(tmp_path / 'r0.pt')
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    r = torch.load(out)
    assert torch.equal(r['flat'], torch.arange(5000, dtype=torch.float32) * 3)      # 1x + 2x summed
    assert float(r['loss']) == 0.5
    torch.manual_seed(100)
    ref = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.BatchNorm1d(5))
    assert torch.equal(r['w'], torch.cat([t.reshape(-1).double() for t in ref.state_dict().values()]))


def test_arena_chunking_covers_everything():
    import distributed as D
    for n in (1, 1023, 1024, 28874625):
        ar = D.ArenaAllReduce(torch.empty(n), n_chunks=4)
        assert ar.bounds[0][0] == 0 and ar.bounds[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(ar.bounds, ar.bounds[1:]))


def _overlap_worker(rank, world, port, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'tacotron2-vae_amd'))
    import distributed as D
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    D.init_distributed(backend='gloo', timeout_s=60)
    torch.manual_seed(0)
    # stand-in with Tacotron2's top-level layout: encoder -> decoder -> postnet in forward, so the postnet
    # gradients are complete first in backward
    net = torch.nn.ModuleDict(dict(emb=torch.nn.Linear(4, 6), encoder=torch.nn.Linear(6, 300),
                                   decoder=torch.nn.Linear(300, 300), postnet=torch.nn.Linear(300, 300)))
    named = list(net.named_parameters())
    offs, total = [], 0
    for _, p in named:
        offs.append(total)
        total += (p.numel() + 3) & ~3
    flat = torch.zeros(total)
    for (_, p), o in zip(named, offs):
        p.grad = flat[o:o + p.numel()].view_as(p)
    ar = D.OverlappedArenaAllReduce(named, offs, flat, min_bucket=64)
    x = torch.randn(5, 4, generator=torch.Generator().manual_seed(10 + rank))
    def run():
        flat.zero_()
        ar.begin()
        y = net['postnet'](torch.tanh(net['decoder'](torch.tanh(net['encoder'](net['emb'](x))))))
        y.pow(2).sum().backward()
        ar.finish()
    run()
    first = flat.clone()
    run()                                               # hooks re-arm every step
    assert torch.equal(first, flat)
    if rank == 0:
        torch.save(dict(flat=flat.clone(), buckets=ar.buckets, log=ar.launch_log, x0=x), out)
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_bucket_allreduce(tmp_path):
    world, port = 2, _free_port()
    out = str(tmp_path / 'r0.pt')
    mp.spawn(_overlap_worker, args=(world, port, out), nprocs=world, join=True)
    r = torch.load(out, weights_only=False)
    names = [b[0] for b in r['buckets']]
    assert names == ['emb+encoder', 'decoder', 'postnet']                    # the tiny embedding rides with its neighbour
    assert r['buckets'][0][1] == 0 and r['buckets'][-1][2] == r['flat'].numel()
    assert all(a[2] == b[1] for a, b in zip(r['buckets'], r['buckets'][1:]))
    # every bucket was issued from a hook (i.e. during backward), postnet first
    assert [bi for bi, _ in r['log']] == [2, 1, 0] and all(h for _, h in r['log'])
    # reduced gradient == sum of the two ranks' single-process gradients
    torch.manual_seed(0)
    net = torch.nn.ModuleDict(dict(emb=torch.nn.Linear(4, 6), encoder=torch.nn.Linear(6, 300),
                                   decoder=torch.nn.Linear(300, 300), postnet=torch.nn.Linear(300, 300)))
    want = None
    for rank in range(world):
        net.zero_grad()
        x = torch.randn(5, 4, generator=torch.Generator().manual_seed(10 + rank))
        y = net['postnet'](torch.tanh(net['decoder'](torch.tanh(net['encoder'](net['emb'](x))))))
        y.pow(2).sum().backward()
        g = torch.cat([torch.nn.functional.pad(p.grad.reshape(-1), (0, (-p.numel()) % 4)) for p in net.parameters()])
        want = g.clone() if want is None else want + g
    assert torch.allclose(r['flat'], want, rtol=1e-5, atol=1e-6)


def _wire_worker(rank, world, port, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'tacotron2-vae_amd'))
    import distributed as D
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    D.init_distributed(backend='gloo', timeout_s=60)
    net = torch.nn.ModuleDict(dict(encoder=torch.nn.Linear(6, 300), decoder=torch.nn.Linear(300, 300)))
    named = list(net.named_parameters())
    offs, total = [], 0
    for _, p in named:
        offs.append(total)
        total += (p.numel() + 3) & ~3
    flat = torch.zeros(total)
    ar = D.OverlappedArenaAllReduce(named, offs, flat, min_bucket=64, wire_dtype=torch.bfloat16)
    g = torch.Generator().manual_seed(3 + rank)
    # (a) the graph engine's exchange: ONE collective over the whole arena, in bf16 on the wire
    flat.copy_(torch.randn(total, generator=g))
    mine = flat.clone()
    ar.reduce_all()
    whole = flat.clone()
    # (b) the eager engine's exchange: buckets (here issued by finish(): no backward ran), bf16 on the wire
    flat.copy_(mine)
    ar.begin()
    ar.finish()
    if rank == 0:
        torch.save(dict(whole=whole, bucketed=flat.clone(), mine=mine, wire=ar.wire_bytes(), total=total), out)
    dist.barrier()
    dist.destroy_process_group()


def test_bf16_wire_exchange_whole_arena_and_buckets(tmp_path):
    """bf16_run's gradient exchange (57.7 MB instead of 115.5 MB): values are rounded to bf16, summed over
    the ranks in bf16 and widened back; the single whole-arena collective of the graph engine and the bucketed one of
    the eager engine give the same sums."""
    world, port = 2, _free_port()
    out = str(tmp_path / 'r0.pt')
    mp.spawn(_wire_worker, args=(world, port, out), nprocs=world, join=True)
    r = torch.load(out, weights_only=False)
    assert r['wire'] == 2 * r['total']
    other = torch.randn(r['total'], generator=torch.Generator().manual_seed(4))
    want = (r['mine'].bfloat16() + other.bfloat16()).float()
    assert torch.equal(r['whole'], want)
    assert torch.equal(r['bucketed'], want)


def _tail_worker(rank, world, port, out, wire):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'tacotron2-vae_amd'))
    import distributed as D
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    D.init_distributed(backend='gloo', timeout_s=60)
    torch.manual_seed(0)
    net = torch.nn.ModuleDict(dict(encoder=torch.nn.Linear(6, 300), decoder=torch.nn.Linear(300, 300),
                                   postnet=torch.nn.Linear(300, 300)))
    named = list(net.named_parameters())
    offs, total = [], 0
    for _, p in named:
        offs.append(total)
        total += (p.numel() + 3) & ~3
    full = torch.zeros(total + 4)                       # gradients + the engine's 4-float poison slot
    for (_, p), o in zip(named, offs):
        p.grad = full[o:o + p.numel()].view_as(p)
    ar = D.OverlappedArenaAllReduce(named, offs, full, min_bucket=64, tail=4,
                                    wire_dtype=torch.bfloat16 if wire else None)
    x = torch.randn(5, 6, generator=torch.Generator().manual_seed(10 + rank))
    seen_by_hooked_bucket = []
    rec = {}
    for step, poisoned_rank in enumerate((None, 1, None)):
        full.zero_()
        full[total:] = 123.0                            # a stale value from the previous step: must never be summed
        ar.begin()
        y = net['postnet'](torch.tanh(net['decoder'](torch.tanh(net['encoder'](x)))))
        y.pow(2).sum().backward()
        # every bucket left from a hook, i.e. BEFORE the engine knows this step's error words
        seen_by_hooked_bucket.append(len(ar.launch_log))
        full[total:] = 1.0 if poisoned_rank == rank else 0.0        # TrainEngine._poison()
        ar.finish()
        rec[step] = full[total:].clone()
    # the graph engine's exchange carries the slot inside its ONE whole-arena collective
    full[total:] = 1.0 if rank == 0 else 0.0
    ar.reduce_all()
    rec['whole'] = full[total:].clone()
    if rank == 0:
        torch.save(dict(rec=rec, buckets=ar.buckets, total=total, hooked=seen_by_hooked_bucket), out)
    dist.barrier()
    dist.destroy_process_group()


def test_poison_slot_is_reduced_after_backward_not_with_a_hook_issued_bucket(tmp_path):
    """the 16-byte poison slot behind the gradients ("one of my persistent kernels timed out") used to
    ride in the last bucket, which a hook issues DURING backward — before the step's error words exist.  It now belongs to
    no bucket; finish() sums it over the ranks by itself: a flag raised on ONE rank after backward is seen by BOTH, a stale
    value is never re-summed, and the next step starts clean.  fp32 and bf16 wire formats."""
    for wire in (False, True):
        world, port = 2, _free_port()
        out = str(tmp_path / ('r0_%d.pt' % wire))
        mp.spawn(_tail_worker, args=(world, port, out, wire), nprocs=world, join=True)
        r = torch.load(out, weights_only=False)
        assert r['buckets'][-1][2] == r['total']                       # no bucket reaches into the slot
        assert r['hooked'] == [3, 3, 3]                                # all three buckets left from hooks each step
        assert torch.equal(r['rec'][0], torch.zeros(4))                # nobody poisoned: stays zero (the stale 123 is gone)
        assert torch.equal(r['rec'][1], torch.ones(4))                 # rank 1 raised it, rank 0 reads it
        assert torch.equal(r['rec'][2], torch.zeros(4))                # and it does not leak into the next step
        assert torch.equal(r['rec']['whole'], torch.ones(4))
