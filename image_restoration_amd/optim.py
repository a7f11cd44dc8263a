"""Flat-arena Adam: the optimiser / gradient-exchange side of the MI355X training path.

All parameters of a network live in ONE contiguous fp32 arena and their gradients in another
(``param.data`` / ``param.grad`` are views).  Consequences:

* the generator's backward writes weight gradients straight into the gradient arena
  (``net._grad_sink``, sr_rrdbnet_backward_f32 accumulate=1) — no per-parameter tensors;
* data-parallel exchange is ONE RCCL all-reduce of the arena (66.8 MB for the 23-block generator)
  instead of DDP's per-bucket traffic — sized for point-to-point xGMI links, where few large
  ring steps beat many small ones;
* the update is one fused kernel (sr_adam_step_f32) with torch.optim.Adam semantics
  (reference: base_model.py:78-83, lr 1e-4, betas (0.9, 0.99) from train_ESRGAN_x4.yml:64-73),
  the 1/world_size of DDP's gradient mean folded in as ``grad_scale``.

state_dict()/load_state_dict() use torch.optim.Adam's format, so ``*.state`` files written by the
reference (base_model.py:279-311) resume here and vice versa.
"""
import ctypes as C

import torch

from . import _lib

_ALIGN = 64  # floats


class _GradSink:
    def __init__(self, ptrs):
        self.grad_ptrs = ptrs


class FlatAdam:

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, modules=()):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError('optimizer got an empty parameter list')
        dev = self.params[0].device
        for p in self.params:
            if p.device != dev or p.dtype != torch.float32:
                raise ValueError('FlatAdam needs fp32 parameters on one device')
        self.device = dev
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = off
        self._epoch = [0]
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.flat_p[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)
                p._sr_epoch = self._epoch   # hip_ops.cached_pack: this optimiser rewrites the parameter behind torch's version counter
        self.param_groups = [dict(params=self.params, lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay,
                                  amsgrad=False)]
        self.step_count = 0
        # steps the fused kernel refused on the device because a launch behind their gradients had timed out (sr_abort_latch):
        # parameters and moments are untouched by those, take_back_skipped() takes them out of step_count again
        self.skipped = torch.zeros(1, dtype=torch.int32, device=dev) if dev.type == 'cuda' else None
        self.modules = list(modules)
        for m in self.modules:
            # generator: let the fused backward accumulate straight into the arena
            if hasattr(m, '_param_list') and [id(p) for p in m._param_list()] == [id(p) for p in self.params]:
                m._grad_sink = _GradSink([self.flat_g.data_ptr() + 4 * o for o in self.offsets])
        self._invalidate()

    # ------------------------------------------------------------------ torch.optim-like surface
    def zero_grad(self, set_to_none=False):
        self.flat_g.zero_()
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)

    def all_reduce_grads(self, group=None):
        """SUM all-reduce of the whole gradient arena (RCCL on GPUs, gloo in CPU tests).  Returns the factor the
        update must scale gradients by (DDP averages: 1/world_size)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return 1.0
        world = dist.get_world_size(group)
        if world == 1:
            return 1.0
        dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM, group=group)
        return 1.0 / world

    def step(self, grad_scale=1.0):
        if not self.flat_p.is_cuda:
            raise _lib.SrHipError('FlatAdam.step runs only on a HIP device (no CPU fallback)')
        lib = _lib.load()
        g = self.param_groups[0]
        self.step_count += 1
        with torch.cuda.device(self.device):
            _lib.check(lib.sr_adam_step_f32(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.exp_avg.data_ptr(),
                                            self.exp_avg_sq.data_ptr(), self.numel, self.step_count, g['lr'], g['betas'][0],
                                            g['betas'][1], g['eps'], g['weight_decay'], grad_scale, lib.sr_abort_latch(),
                                            self.skipped.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream),
                       'sr_adam_step_f32')
        self._invalidate()

    def take_back_skipped(self):
        """After a time-out was reported: how many of the steps issued since were refused on the device (their bias-correction
        count is taken back, so the next real step continues the sequence).  Synchronises."""
        if self.skipped is None:
            return 0
        n = int(self.skipped.item())
        if n:
            self.step_count -= n
            self.skipped.zero_()
        return n

    def _invalidate(self):
        self._epoch[0] += 1
        for m in self.modules:
            if hasattr(m, 'invalidate_packed'):
                m.invalidate_packed()

    # ------------------------------------------------------------------ checkpoint format of torch.optim.Adam
    def state_dict(self):
        state = {}
        if self.step_count > 0:
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                state[i] = dict(step=torch.tensor(float(self.step_count)),
                                exp_avg=self.exp_avg[o:o + p.numel()].view(p.shape).clone(),
                                exp_avg_sq=self.exp_avg_sq[o:o + p.numel()].view(p.shape).clone())
        group = {k: v for k, v in self.param_groups[0].items() if k != 'params'}
        group['params'] = list(range(len(self.params)))
        return dict(state=state, param_groups=[group])

    def load_state_dict(self, sd):
        group = sd['param_groups'][0]
        for k, v in group.items():
            if k != 'params':
                self.param_groups[0][k] = tuple(v) if k == 'betas' else v
        steps = set()
        with torch.no_grad():
            for i, st in sd['state'].items():
                i = int(i)
                p, o = self.params[i], self.offsets[i]
                self.exp_avg[o:o + p.numel()].view(p.shape).copy_(st['exp_avg'])
                self.exp_avg_sq[o:o + p.numel()].view(p.shape).copy_(st['exp_avg_sq'])
                steps.add(int(float(st['step'])))
        assert len(steps) <= 1, 'per-parameter step counts differ'
        self.step_count = steps.pop() if steps else 0


def ema_update(ema_opt_or_flat, src_flat, decay, modules=()):
    """flat_ema = decay*flat_ema + (1-decay)*flat_src  (model_ema, base_model.py:50-57) as one HIP launch."""
    lib = _lib.load()
    dst = ema_opt_or_flat
    assert dst.numel() == src_flat.numel()
    with torch.cuda.device(dst.device):
        _lib.check(lib.sr_axpby_f32(dst.data_ptr(), src_flat.data_ptr(), float(decay), float(1.0 - decay), dst.numel(),
                                    lib.sr_abort_latch(), torch.cuda.current_stream(dst.device).cuda_stream), 'sr_axpby_f32')
    for m in modules:
        if hasattr(m, 'invalidate_packed'):
            m.invalidate_packed()


def flatten_parameters(net):
    """Moves a network's parameters into one arena (no optimiser) and returns it — used for the EMA copy."""
    params = [p for p in net.parameters()]
    dev = params[0].device
    offsets, off = [], 0
    for p in params:
        offsets.append(off)
        off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
    flat = torch.zeros(off, dtype=torch.float32, device=dev)
    with torch.no_grad():
        for p, o in zip(params, offsets):
            view = flat[o:o + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
    if hasattr(net, 'invalidate_packed'):
        net.invalidate_packed()
    return flat
