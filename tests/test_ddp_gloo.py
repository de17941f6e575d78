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
