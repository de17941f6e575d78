"""Data-parallel plumbing over RCCL/xGMI (torch.distributed backend "nccl" IS RCCL on ROCm).

Replaces the reference's hand-rolled flatten-all-grads hook (distributed.py:126-174) and its
launcher (multiproc.py): one process per GPU (torchrun-style env: RANK/LOCAL_RANK/WORLD_SIZE/
MASTER_*), parameters broadcast once from rank 0, and ONE logical all-reduce(SUM) per step
executed directly on the optimiser's flat gradient arena (zero copies, no flatten/unflatten).
xGMI is point-to-point, so the 115.5 MB fp32 message is split into a few large chunks that
are all in flight together rather than many small buckets.  The 1/world averaging
(reference distributed.py:162) is folded into the fused clip+Adam kernel.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))


def init_distributed(hparams=None, n_gpus=None, rank=None, group_name=None, backend=None, timeout_s=600):
    """reference train.py:38-50.  Rendezvous comes from the torchrun env when present, else from
    hparams.dist_url like the reference."""
    import datetime
    if dist.is_initialized():
        return
    world, env_rank, local = env_world()
    backend = backend or (hparams.dist_backend if hparams is not None else 'nccl')
    if backend == 'nccl':
        if not torch.cuda.is_available():
            raise RuntimeError("Distributed mode requires a GPU (backend nccl = RCCL).")
        torch.cuda.set_device((local if 'LOCAL_RANK' in os.environ else (rank or 0)) % torch.cuda.device_count())
    kw = dict(backend=backend, timeout=datetime.timedelta(seconds=timeout_s))
    if 'MASTER_ADDR' in os.environ and 'RANK' in os.environ:
        dist.init_process_group(init_method='env://', world_size=world, rank=env_rank, **kw)
    else:
        dist.init_process_group(init_method=hparams.dist_url, world_size=n_gpus, rank=rank, **kw)


class _Done(object):
    def wait(self):
        return True


def _stage_through_host(t, group):
    """gloo worlds (CPU tests; two ranks sharing ONE GPU in the -m gpu suite, where RCCL refuses duplicate devices)
    reduce / broadcast device tensors through host memory in fp32 — the semantics of the collective, not its speed."""
    return t.is_cuda and dist.get_backend(group) == 'gloo'


def all_reduce_sum(t, group=None, async_op=False):
    """all-reduce(SUM) of `t` in place; returns a work handle with .wait() when async_op"""
    if _stage_through_host(t, group):
        c = t.detach().float().cpu()
        dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
        t.copy_(c)
        return _Done() if async_op else None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def broadcast(t, src=0, group=None):
    if _stage_through_host(t, group):
        c = t.detach().cpu()
        dist.broadcast(c, src, group=group)
        t.copy_(c)
        return
    dist.broadcast(t, src, group=group)


def broadcast_state(module, src=0):
    """rank-0 weights and buffers to everybody (reference distributed.py:132-135), coalesced per
    dtype instead of 142 separate broadcasts."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    by_dtype = {}
    for t in module.state_dict().values():
        if torch.is_tensor(t):
            by_dtype.setdefault(t.dtype, []).append(t)
    for dtype, tensors in by_dtype.items():
        flat = torch.cat([t.reshape(-1) for t in tensors])
        broadcast(flat, src)
        o = 0
        for t in tensors:
            t.copy_(flat[o:o + t.numel()].view_as(t))
            o += t.numel()


class ArenaAllReduce(object):
    """all-reduce(SUM) of a flat gradient tensor in `n_chunks` concurrent pieces."""

    def __init__(self, flat, n_chunks=4, group=None):
        self.flat, self.group = flat, group
        n = flat.numel()
        step = -(-n // max(1, n_chunks))
        step = (step + 1023) & ~1023
        self.bounds = [(a, min(n, a + step)) for a in range(0, n, step)]

    def __call__(self):
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return
        works = [all_reduce_sum(self.flat[a:b], group=self.group, async_op=True) for a, b in self.bounds]
        for w in works:
            w.wait()


class OverlappedArenaAllReduce(object):
    """Gradient all-reduce in a few buckets that start as soon as their gradients exist.

    The arena is cut at top-level-module boundaries (parameter order = arena order), so a bucket is one
    contiguous slice.  Each live parameter carries a post-accumulate-grad hook; when the last one of a bucket
    has fired, that slice's all-reduce is issued asynchronously — for Tacotron2 the Postnet slice (first to
    finish in backward) travels over xGMI while the decoder BPTT is still running, the decoder slice while the
    encoder / reference-encoder backward runs, and only the small remainder is exposed.  `finish()` (after
    backward) issues whatever has not gone yet — tensors that got no gradient this step never fire — and waits.
    One collective per bucket on a point-to-point fabric: few, large messages (xGMI rings are per-link bound)."""

    def __init__(self, named_params, offsets, flat, group=None, min_bucket=1 << 16, force=False, side_streams=None,
                 gather=None, wire_dtype=None, tail=0):
        """named_params: [(name, param)] in arena order; offsets: start of each param inside `flat`.
        wire_dtype=torch.bfloat16 (bf16_run, SURVEY 8(e): 57.7 MB instead of 115.5 MB per step): every slice is rounded
        into a bf16 wire buffer, summed over the ranks in bf16, and widened back into the fp32 arena.
        tail: the last `tail` elements of `flat` are no gradients (the engine's poison slot: "a kernel of mine timed out").
        They belong to NO hook-issued bucket — a bucket leaves while backward is still running, before the step's error
        words exist — and are reduced by finish() in a small collective of their own, in fp32 on any wire format;
        reduce_all() carries them with the arena."""
        self.flat, self.group = flat, group
        self.tail = int(tail)
        self.wire = None
        if wire_dtype is not None and wire_dtype != flat.dtype:
            self.wire = torch.empty(flat.numel(), dtype=wire_dtype, device=flat.device)
        self.force = force      # run the hooks and collectives even in a 1-rank group (tests)
        # callable -> streams other than the current one on which gradients of the model may be produced (the model
        # runs its reference-encoder branch on a side stream): a bucket waits for them before it goes out
        self.side_streams = side_streams
        # callable(list of params): bring those parameters' gradients into `flat` (FlatAdam.gather_grads) — the
        # arena is filled bucket by bucket right before each bucket's collective
        self.gather = gather
        groups = []         # [top-level module, lo, hi, [params]] in arena order
        for (name, p), off in zip(named_params, offsets):
            top = name.split('.', 1)[0]
            end = off + ((p.numel() + 3) & ~3)
            if groups and groups[-1][0] == top:
                groups[-1][2] = end
                groups[-1][3].append(p)
            else:
                groups.append([top, off, end, [p]])
        buckets = []
        for g in groups:    # a tiny group (e.g. the symbol embedding) rides with its arena neighbour
            if buckets and buckets[-1][2] - buckets[-1][1] < min_bucket:
                buckets[-1][0] += '+' + g[0]
                buckets[-1][2] = g[2]
                buckets[-1][3] += g[3]
            else:
                buckets.append(g)
        if len(buckets) > 1 and buckets[-1][2] - buckets[-1][1] < min_bucket:
            last = buckets.pop()
            buckets[-1][0] += '+' + last[0]
            buckets[-1][2] = last[2]
            buckets[-1][3] += last[3]
        if buckets:
            buckets[-1][2] = flat.numel() - self.tail
        self.buckets = [(b[0], b[1], b[2], len(b[3])) for b in buckets]
        self._bucket_params = [list(b[3]) for b in buckets]
        self._pending = [0] * len(buckets)
        self._works = [None] * len(buckets)
        self._active = False
        self.launch_log = []          # (bucket index, launched from a hook i.e. while backward was running)
        self._exposed = []
        for bi, b in enumerate(buckets):
            for p in b[3]:
                p.register_post_accumulate_grad_hook(self._make_hook(bi))

    def _make_hook(self, bi):
        def hook(_param):
            if not self._active:
                return
            self._pending[bi] -= 1
            if self._pending[bi] == 0:
                self._launch(bi, True)
        return hook

    def _launch(self, bi, from_hook):
        if self._works[bi] is not None:
            return
        _, lo, hi, _ = self.buckets[bi]
        if self.flat.is_cuda and self.side_streams is not None:
            cur = torch.cuda.current_stream()
            for st in self.side_streams():
                cur.wait_stream(st)
        if self.gather is not None:
            self.gather(self._bucket_params[bi])
        self._works[bi] = self._issue(lo, hi)
        self.launch_log.append((bi, from_hook))

    def _issue(self, lo, hi):
        if self.wire is None:
            return all_reduce_sum(self.flat[lo:hi], group=self.group, async_op=True)
        w = self.wire[lo:hi]
        w.copy_(self.flat[lo:hi])
        return all_reduce_sum(w, group=self.group, async_op=True)

    def wire_bytes(self):
        t = self.flat if self.wire is None else self.wire
        return t.numel() * t.element_size()

    def reduce_all(self):
        """ONE collective over the whole arena (the graph engine's exchange, and its eager warm-up / fallback steps: the
        collective pattern of a rank must not depend on whether that rank replays a graph or runs eagerly this
        iteration — ranks see different batch shapes).  Timed like finish()."""
        timed = self.flat.is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        self._issue(0, self.flat.numel()).wait()
        if self.wire is not None:
            self.flat.copy_(self.wire)
        if timed:
            e1.record()
            self._exposed.append((e0, e1))
            del self._exposed[:-64]

    def begin(self):
        """call before backward()"""
        self._active = dist.is_initialized() and (self.force or dist.get_world_size(self.group) > 1)
        self._pending = [b[3] for b in self.buckets]
        self._works = [None] * len(self.buckets)
        self.launch_log = []

    def finish(self):
        """call after backward(): everything is reduced (summed) when this returns"""
        if not self._active:
            return
        for bi in range(len(self.buckets)):
            self._launch(bi, False)
        tail_work = None
        if self.tail:       # written by the caller after backward, on the current stream: the collective is ordered behind it
            tail_work = all_reduce_sum(self.flat[self.flat.numel() - self.tail:], group=self.group, async_op=True)
        timed = self.flat.is_cuda
        if timed:       # the compute stream stalls exactly between these two events: the EXPOSED part of the exchange
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for w in self._works:
            w.wait()
        if tail_work is not None:
            tail_work.wait()
        if self.wire is not None:           # the waits above ordered the compute stream behind the collectives
            n = self.flat.numel() - self.tail
            self.flat[:n].copy_(self.wire[:n])
        if timed:
            e1.record()
            self._exposed.append((e0, e1))
            del self._exposed[:-64]
        self._active = False

    def exposed_ms(self):
        """mean time per step the compute stream waited for collectives that had not finished when backward ended
        (events on the compute stream around the waits of the last <= 64 steps; synchronises)"""
        if not self._exposed:
            return 0.0
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self._exposed) / len(self._exposed)


def reduce_tensor(tensor, n_gpus):
    """reference train.py:31-35 (mean of a scalar over ranks, for logging)."""
    rt = tensor.clone()
    all_reduce_sum(rt)
    rt /= n_gpus
    return rt


def apply_gradient_allreduce(module):
    """Name kept from reference distributed.py:126 so `load_model()` reads the same: broadcast the
    state; the per-step reduction is attached by train.py to the optimiser's gradient arena."""
    broadcast_state(module)
    module._t2v_data_parallel = True
    return module
