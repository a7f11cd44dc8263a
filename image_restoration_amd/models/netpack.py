"""NetPack: one network of a training job the way the MI355X path holds it.

The reference wraps every network in DistributedDataParallel (basicsr/models/base_model.py:62-76) and hands its
``parameters()`` to ``torch.optim.Adam`` (:78-83).  Here a network is a *pack*: the module, the fp32 parameter /
gradient / moment arenas of ``optim.FlatAdam`` it was moved into, and optionally an EMA shadow network living in an
arena of the same layout (reference ``model_ema``, base_model.py:50-57).  The pack is the unit of

* the optimiser step: ONE all-reduce of the gradient arena (RCCL) + ONE fused Adam launch;
* the EMA blend: one axpby launch over the two arenas;
* replica synchronisation: what DDP's constructor does with ``_sync_module_states`` (parameters AND buffers of
  rank 0 go to every rank) is ``NetPack.align_replicas`` here.  Without it the ranks of a job seeded ``seed + rank``
  (utils/options.py, reference options.py:148) would train different networks on averaged gradients.
"""
import torch

from .. import optim


def first_rank_broadcast(tensors, group=None):
    """Every tensor of ``tensors`` becomes rank 0's copy, in place.  Large tensors (the arenas) travel as they are;
    the many small ones (BatchNorm statistics, spectral-norm vectors) are packed per dtype into one message."""
    import torch.distributed as dist
    small = {}
    for t in tensors:
        if t.numel() >= (1 << 16) and t.is_contiguous():
            dist.broadcast(t, 0, group=group)
        else:
            small.setdefault((t.dtype, t.device), []).append(t)
    for (_, _), bunch in small.items():
        wire = torch.cat([t.detach().reshape(-1) for t in bunch])
        dist.broadcast(wire, 0, group=group)
        at = 0
        with torch.no_grad():
            for t in bunch:
                t.copy_(wire[at:at + t.numel()].view(t.shape))
                at += t.numel()


class NetPack:

    def __init__(self, label, net):
        self.label = label
        self.net = net
        self.adam = None
        self.shadow = None
        self.shadow_arena = None

    # -------------------------------------------------------------- construction
    def attach_adam(self, settings, log=None):
        """``settings`` is the yml block ``train.optim_<label>`` minus its ``type`` (lr, weight_decay, betas, ...).
        Parameters with requires_grad False stay outside the arena, with the reference's warning (sr_model.py:78-83)."""
        chosen = []
        for name, p in self.net.named_parameters():
            if p.requires_grad:
                chosen.append(p)
            elif log is not None:
                log.warning(f'Params {name} will not be optimized.')
        self.adam = optim.FlatAdam(chosen, modules=[self.net], **settings)
        self.adam.all_params = list(self.net.parameters())   # freeze() toggles these every step: one walk of the module tree, here
        return self.adam

    def attach_shadow(self, shadow_net):
        self.shadow = shadow_net.eval()
        self.shadow_arena = optim.flatten_parameters(shadow_net)

    # -------------------------------------------------------------- one optimiser step
    def freeze(self, frozen=True):
        params = self.adam.all_params if self.adam is not None and getattr(self.adam, 'all_params', None) else list(self.net.parameters())
        for p in params:
            p.requires_grad = not frozen

    def clear_grads(self):
        self.join_lane()   # (a deferred backward nobody consumed: its lane still adds into the arena)
        self.adam.zero_grad()

    def defer_weight_gradients(self, on=True):
        """The network's backward may return while its weight gradients still run on the library's second lane
        (sr_set_backward_wgrad_deferred); ``update`` joins before it touches the gradient arena.  Only for whole-network
        backward drivers that write into this pack's arena."""
        import os
        self.net._defer_wgrad = bool(on)
        self.net._defer_mode = int(os.environ.get('SR_DEFER_MODE', '1'))   # development: 2 = dense-block weight gradients behind the dgrad chain
        if not hasattr(self.net, '_lane_holds'):
            self.net._lane_holds = []

    def join_lane(self):
        """The current stream waits for the weight gradients a deferred backward left running; what they read is released."""
        holds = getattr(self.net, '_lane_holds', None)
        if holds:
            from .. import _lib
            dev = self.adam.flat_g.device
            with torch.cuda.device(dev):
                _lib.check(_lib.load().sr_backward_lane_join(torch.cuda.current_stream(dev).cuda_stream), 'sr_backward_lane_join')
            holds.clear()

    def update(self, distributed):
        """Gradient exchange (SUM over ranks, the 1/world of DDP's mean folded into the Adam kernel) + Adam."""
        self.join_lane()
        factor = self.adam.all_reduce_grads() if distributed else 1.0
        self.adam.step(grad_scale=factor)

    def blend_shadow(self, decay):
        """shadow = decay*shadow + (1-decay)*live ; decay 0 copies."""
        live = self.adam.flat_p
        if decay == 0:
            with torch.no_grad():
                self.shadow_arena.copy_(live)
            self.shadow.invalidate_packed()
        else:
            if not live.is_cuda:
                raise RuntimeError('EMA update runs only on a HIP device')
            optim.ema_update(self.shadow_arena, live, decay, modules=[self.shadow])

    # -------------------------------------------------------------- replicas
    def state_tensors(self, buffers_only=False):
        """Tensors that define this replica: parameter arena (or loose parameters before the optimiser exists),
        module buffers, and the shadow's arena + buffers."""
        out = []
        if not buffers_only:
            if self.adam is not None:
                out.append(self.adam.flat_p)
                inside = {id(p) for p in self.adam.params}
                out += [p.data for p in self.net.parameters() if id(p) not in inside]
            else:
                out += [p.data for p in self.net.parameters()]
            if self.shadow_arena is not None:
                out.append(self.shadow_arena)
        out += list(self.net.buffers())
        if self.shadow is not None and not buffers_only:
            out += list(self.shadow.buffers())
        return out

    def align_replicas(self, buffers_only=False):
        first_rank_broadcast(self.state_tensors(buffers_only))
        if not buffers_only:
            from .. import hip_ops
            hip_ops.invalidate_packs()   # (a collective does not bump torch's version counters: cached weight images are stale)
        if not buffers_only:
            for m in (self.net, self.shadow):
                if m is not None and hasattr(m, 'invalidate_packed'):
                    m.invalidate_packed()
