"""Flat-arena optimiser state for the MI355X training loop.

All trainable tensors live in ONE fp32 arena (parameter order), with matching arenas for
gradients, exp_avg and exp_avg_sq: the gradient all-reduce is a zero-copy collective on the
arena and clip-by-global-norm + Adam is two HIP launches (csrc/optim.hip) instead of ~400
small kernels.  `state_dict()` / `load_state_dict()` speak torch.optim.Adam's format so the
checkpoint dict of reference train.py:113-119 stays interchangeable.
"""
import ctypes as C
import math

import torch

import t2v_hip

# tensors the reference constructs but never uses in forward: they get no
# gradient there, are skipped by Adam and by the all-reduce, and must not be touched here either.
DEAD_PARAM_PREFIXES = ('speaker_embedding.', 'emotion_embedding.')
DEAD_PARAM_NAMES = ('vae_gst.ref_encoder.convs.0.weight', 'vae_gst.ref_encoder.convs.0.bias')


def is_dead_param(name):
    return name.startswith(DEAD_PARAM_PREFIXES) or name in DEAD_PARAM_NAMES


class FlatAdam(object):
    """clip_grad_norm_ + Adam over a flat arena.  Mirrors the bits of torch.optim.Adam the
    reference loop touches: `.param_groups[i]['lr']`, `.step()`, `.state_dict()`,
    `.load_state_dict()`; plus `.zero_grad()` (the loop's `model.zero_grad()` also works) and
    `.grad_norm` (device scalar written by the last step)."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_clip_thresh=1.0,
                 world_size=1):
        named = list(model.named_parameters())
        self._all_names = [n for n, _ in named]
        self._live = [(i, n, p) for i, (n, p) in enumerate(named) if not is_dead_param(n)]
        if not self._live:
            raise ValueError("no trainable parameters")
        dev = self._live[0][2].device
        if dev.type != 'cuda':
            raise t2v_hip.T2VHipError("FlatAdam needs the model on a GPU (no CPU fallback)")
        t2v_hip.load_library()
        offs, total = [], 0
        for _, _, p in self._live:
            offs.append(total)
            total += (p.numel() + 3) & ~3          # 16-byte aligned slots
        self.numel = total
        self._offsets = offs
        f32 = dict(device=dev, dtype=torch.float32)
        self.params = torch.zeros(total, **f32)
        # (+4: one 16-byte slot behind the last gradient that is NOT part of the norm / the update.  A multi-rank engine writes
        # "one of my persistent kernels timed out" there before the all-reduce: after the SUM every rank sees a non-zero word,
        # every rank skips the update, every rank re-runs the iteration — train.TrainEngine.recover)
        self._grads_full = torch.zeros(total + 4, **f32)
        self.grads = self._grads_full[:total]
        self.exp_avg = torch.zeros(total, **f32)
        self.exp_avg_sq = torch.zeros(total, **f32)
        self._partials = torch.zeros(1024, **f32)
        self.grad_norm = torch.zeros(1, **f32)
        self._grad_views = []
        for (i, n, p), o in zip(self._live, offs):
            view = self.params[o:o + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view                                   # parameter now lives in the arena
            p.grad = None
            self._grad_views.append(self.grads[o:o + p.numel()].view_as(p))
        self._view_of = {id(p): v for (_, _, p), v in zip(self._live, self._grad_views)}
        # backward passes that know these slots write weight gradients straight into the arena (t2v_hip.grad_slot)
        t2v_hip.register_grad_slots({p.data_ptr(): (self.grads, o, tuple(p.shape))
                                     for (_, _, p), o in zip(self._live, offs)})
        self._gathered = False
        self._no_grad = []
        self._slot_of = {id(p): (o, p.numel()) for (_, _, p), o in zip(self._live, offs)}
        self.param_groups = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay,
                                  amsgrad=False, maximize=False)]
        self.grad_clip_thresh = grad_clip_thresh
        self.world_size = world_size
        self.step_count = 0
        self.step_params = None     # the owning engine's device-side record (train.TrainEngine), else the active one
        self.guard = None           # int32 device view: error words of this step's persistent kernels (TrainEngine sets it)

    SKIPPED_NORM_BITS = 0x7fc0beef      # grad_norm after a step the device skipped (t2v_clip_adam_step_guarded)

    def poison_slot(self):
        """the 4-float tail of the gradient arena (travels with the all-reduce when the collective covers grads_for_allreduce())"""
        return self._grads_full[self.numel:]

    def grads_for_allreduce(self):
        return self._grads_full

    def live_params(self):
        return [p for _, _, p in self._live]

    def arena_layout(self):
        """([(name, param)], [offset]) of the live parameters in arena order (for the bucketed all-reduce)"""
        return [(n, p) for _, n, p in self._live], list(self._offsets)

    # -- loop API
    def zero_grad(self, set_to_none=True):
        """Gradients are NOT accumulated into the arena by autograd (that costs one add launch per tensor, ~110 per
        step): autograd keeps its own `.grad` tensors and `gather_grads()` copies them into the arena with one
        multi-tensor launch before the all-reduce / the fused clip+Adam."""
        for _, _, p in self._live:
            p.grad = None
        self._gathered = False
        self._no_grad = []

    def gather_grads(self, params=None):
        """copy `.grad` of `params` (default: all live parameters) into their arena slots; a parameter without a
        gradient gets zeros.  DP buckets call this per bucket from their hooks."""
        plist = [p for _, _, p in self._live] if params is None else params
        views, srcs = [], []
        for p in plist:
            v = self._view_of[id(p)]
            if p.grad is None:
                v.zero_()
                self._no_grad.append(p)      # torch.optim.Adam skips such a parameter entirely: see step()
            elif p.grad.data_ptr() != v.data_ptr():
                views.append(v)
                srcs.append(p.grad)
        if views:
            torch._foreach_copy_(views, srcs)
        if params is None:
            self._gathered = True

    def mark_gathered(self):
        self._gathered = True

    def bias_corrections(self, step_count):
        """(1 - beta1^t, 1 - beta2^t) in double precision like torch.optim.Adam (fp32 powf loses ~1e-4 at small t)"""
        b1, b2 = self.param_groups[0]['betas']
        return 1.0 - float(b1) ** step_count, 1.0 - float(b2) ** step_count

    def publish_step_params(self, sp, step_count=None):
        """write lr / bias corrections of optimiser step `step_count` (default: the next one) into the device record"""
        bc1, bc2 = self.bias_corrections(self.step_count + 1 if step_count is None else step_count)
        sp.set(lr=self.param_groups[0]['lr'], bc1=bc1, bc2s=math.sqrt(bc2))

    def step(self):
        if not self._gathered:
            self.gather_grads()
        self._gathered = False
        g = self.param_groups[0]
        # a live parameter without a gradient is skipped by torch.optim.Adam (no weight decay, no moment decay, no
        # step); the fused kernel updates the whole arena, so such slots are put back afterwards (rare: partial freezing)
        skipped = [(self._slot_of[id(p)],) for p in self._no_grad]
        saved = [(o, n, self.params[o:o + n].clone(), self.exp_avg[o:o + n].clone(), self.exp_avg_sq[o:o + n].clone())
                 for ((o, n),) in skipped]
        self._no_grad = []
        self.step_count += 1
        bc1, bc2 = self.bias_corrections(self.step_count)
        sp = self.step_params if self.step_params is not None else t2v_hip.step_params(create=False)
        if sp is not None:       # the kernel reads lr / bias corrections from the device record once one is installed
            self.publish_step_params(sp, self.step_count)
            sp.upload()
        lib = t2v_hip.load_library()
        guard, self.guard = self.guard, None
        rc = lib.t2v_clip_adam_step_guarded(
            C.c_void_p(self.params.data_ptr()), C.c_void_p(self.grads.data_ptr()),
            C.c_void_p(self.exp_avg.data_ptr()), C.c_void_p(self.exp_avg_sq.data_ptr()),
            C.c_uint64(self.numel), C.c_float(g['lr']), C.c_float(g['betas'][0]), C.c_float(g['betas'][1]),
            C.c_float(g['eps']), C.c_float(g['weight_decay']),
            C.c_float(self.grad_clip_thresh if self.grad_clip_thresh else 0.0),
            C.c_float(1.0 / self.world_size), C.c_float(bc1), C.c_float(bc2),
            C.c_void_p(self._partials.data_ptr()), C.c_void_p(self.grad_norm.data_ptr()),
            None if guard is None else C.c_void_p(guard.data_ptr()), 0 if guard is None else int(guard.numel()),
            C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise t2v_hip.T2VHipError("t2v_clip_adam_step rc=%d" % rc)
        for o, n, p0, m0, v0 in saved:
            self.params[o:o + n].copy_(p0)
            self.exp_avg[o:o + n].copy_(m0)
            self.exp_avg_sq[o:o + n].copy_(v0)
        return self.grad_norm

    # -- checkpoint interchange with torch.optim.Adam (reference train.py:100-119)
    def state_dict(self):
        state = {}
        if self.step_count > 0:
            for (i, n, p), o in zip(self._live, self._offsets):
                k = p.numel()
                state[i] = dict(step=torch.tensor(float(self.step_count)),
                                exp_avg=self.exp_avg[o:o + k].view_as(p).clone(),
                                exp_avg_sq=self.exp_avg_sq[o:o + k].view_as(p).clone())
        group = dict(self.param_groups[0])
        group.update(foreach=None, capturable=False, differentiable=False, fused=None,
                     params=list(range(len(self._all_names))))
        return dict(state=state, param_groups=[group])

    def load_state_dict(self, sd):
        grp = sd['param_groups'][0]
        for k in ('lr', 'betas', 'eps', 'weight_decay'):
            if k in grp:
                self.param_groups[0][k] = tuple(grp[k]) if k == 'betas' else grp[k]
        steps = set()
        for (i, n, p), o in zip(self._live, self._offsets):
            st = sd['state'].get(i, sd['state'].get(str(i)))
            if st is None:
                continue
            k = p.numel()
            self.exp_avg[o:o + k].view_as(p).copy_(st['exp_avg'])
            self.exp_avg_sq[o:o + k].view_as(p).copy_(st['exp_avg_sq'])
            steps.add(int(float(st['step'])))
        if len(steps) > 1:
            raise ValueError("per-parameter Adam step counts differ: %s" % sorted(steps))
        self.step_count = steps.pop() if steps else 0

