"""SRGANModel: generator + discriminator, one adversarial step driver for both GAN formulations of the path.

Behavioural counterpart of basicsr/models/srgan_model.py:15-143 (plain GAN loss) and, through the ``relativistic``
switch that ESRGANModel turns on, of basicsr/models/esrgan_model.py:12-83 (relativistic average GAN).  What is kept
is what a user of the reference can observe: option keys (``network_d``, ``path.pretrain_network_d / strict_load_d``,
``train.{optim_d, gan_opt, net_d_iters, net_d_init_iters}``), the order and number of discriminator forwards (they
move BatchNorm statistics), which logits carry a graph, the log keys and the checkpoint files.

A step has two phases:

``_generator_phase``  D frozen; G forward; when it is G's turn: content terms + adversarial term -> backward -> Adam.
``_critic_phase``     D live; real and fake are back-propagated by two separate backward calls that accumulate in the
                      gradient arena, which is exchanged ONCE before D's Adam (the reference's DDP exchanges twice;
                      sums are the same).

Repeated discriminator forwards.  Within one step the reference calls ``net_d`` on the same two tensors several times while
D's weights do not change (esrgan_model.py:38-39 in the generator phase, :65-66,70 in the critic phase: ``net_d(gt)`` twice,
``net_d(output)`` three times; srgan_model.py:104,122: ``net_d(output)`` twice).  For a network whose train-mode forward is a
pure function of (weights, input) — the BatchNorm VGG discriminators: batch statistics only, deterministic launches — the
repeats are bit-identical, so ``_critic`` runs each distinct forward ONCE, keeps its activations for every backward that needs
them, and replays only what a repeat changes: the BatchNorm running statistics and ``num_batches_tracked``, in the reference's
call order.  This does NOT hold for UNetDiscriminatorSN: spectral normalisation does one power iteration per train-mode forward,
which moves ``weight_u / weight_v`` and with them the effective weights, so its five forwards differ and all run.
"""
import torch

from .. import hip_autograd as A
from ..archs import build_network
from ..losses import build_loss
from ..utils.registry import MODEL_REGISTRY
from .sr_model import LossBook, SRModel


@MODEL_REGISTRY.register()
class SRGANModel(SRModel):

    log_prefix = 'l_g_'
    relativistic = False

    # ------------------------------------------------------------------ set-up
    def init_training_settings(self):
        cfg = self.opt['train']
        self.critic = self.adopt('d', build_network(self.opt['network_d']))
        self.net_d = self.critic.net
        self.net_g.train()
        self.net_d.train()
        self._build_content_losses(cfg)
        if cfg.get('gan_opt'):
            self.cri_gan = build_loss(cfg['gan_opt']).to(self.device)
        # option key beyond the reference's (default on): run a repeated forward of a repeatable discriminator once per step
        self.reuse_d_forwards = bool(cfg.get('reuse_d_forwards', True))
        self._d_kept, self.d_forwards_run = {}, 0
        # (same switch family) net_d(gt) of the generator phase on a second stream beside G's forward
        self.prefetch_d_real = self.reuse_d_forwards and bool(cfg.get('prefetch_d_real', True)) and self.device.type == 'cuda'
        self._d_side, self._d_real_ready = None, None
        # option key beyond the reference's (default on for a bf16 generator): G's weight gradients overlap the critic phase
        self.overlap_g_wgrad = bool(cfg.get('overlap_g_wgrad', True)) and getattr(self.net_g, 'compute_dtype', 'fp32') == 'bf16' \
            and self.device.type == 'cuda'
        self.net_d_iters = cfg.get('net_d_iters', 1)
        self.net_d_init_iters = cfg.get('net_d_init_iters', 0)
        self.setup_optimizers()
        self.setup_schedulers()
        self._build_shadow(cfg)
        if self.overlap_g_wgrad:
            self.gen.defer_weight_gradients(True)

    def setup_optimizers(self):
        self.optimizer_g = self.make_adam(self.gen, self.opt['train']['optim_g'])
        self.optimizer_d = self.make_adam(self.critic, self.opt['train']['optim_d'])

    # ------------------------------------------------------------------ discriminator calls of one step
    def _critic(self, x, tag):
        """``self.net_d(x)``; ``tag`` names the tensor ('gt' / 'out').  With a repeatable network the first call of a tag in
        this step runs and keeps the forward, later calls reuse it (module docstring).  The cache lives for one
        optimize_parameters and is dropped before D's weights move."""
        net = self.net_d
        if not (getattr(net, 'repeatable_forward', False) and net.training and self.reuse_d_forwards):
            self.d_forwards_run += 1
            return net(x)
        slot = []
        out = net(x, kept=self._d_kept.get(tag), slot=slot)
        if self._d_kept.get(tag) is not slot[0]:
            self.d_forwards_run += 1
        self._d_kept[tag] = slot[0]
        return out

    def _prefetch_real_logits(self, current_iter):
        """``net_d(gt)`` of the generator phase (esrgan_model.py:38: no graph, constants for G) depends on nothing G computes, and
        at the recipe's patch size neither it nor G's forward fills the chip: it is issued on a second stream BEFORE G's forward and
        joined where its logits are used.  Only with a repeatable discriminator (the forward is kept and serves the critic phase,
        its BatchNorm statistics land before the next call's by the event order) — same kernels, same order per buffer, same bits."""
        self._d_real_ready = None
        net = self.net_d
        if not (self.prefetch_d_real and self.relativistic and self.generator_turn(current_iter) and net.training
                and getattr(net, 'repeatable_forward', False) and self.gt.is_cuda):
            return
        cur = torch.cuda.current_stream()
        if self._d_side is None:
            self._d_side = torch.cuda.Stream()
        side = self._d_side
        side.wait_stream(cur)   # gt, D's weights (last step's Adam) and the refreshed buffers are final on the caller's stream
        with torch.cuda.stream(side), torch.no_grad():
            logits = self._critic(self.gt, 'gt')
        kept = self._d_kept['gt']
        for t in (kept.saved, kept.logits, logits):
            t.record_stream(cur)   # allocated under the side stream, read by the caller's stream until the step ends
        self._d_real_ready = (side.record_event(), logits)

    def generator_turn(self, current_iter):
        """G trains every ``net_d_iters`` iterations once ``net_d_init_iters`` have passed."""
        return current_iter > self.net_d_init_iters and current_iter % self.net_d_iters == 0

    # ------------------------------------------------------------------ adversarial terms
    def _fool_critic(self):
        """Generator-side GAN term on ``self.output`` (graph into G)."""
        if not self.relativistic:
            return self.cri_gan(self._critic(self.output, 'out'), True, is_disc=False)
        if self._d_real_ready is not None:         # issued beside G's forward (_prefetch_real_logits)
            landed, on_real = self._d_real_ready
            self._d_real_ready = None
            torch.cuda.current_stream().wait_event(landed)
        else:
            with torch.no_grad():                  # esrgan_model.py:38 - real logits are constants for G
                on_real = self._critic(self.gt, 'gt')
        on_fake = self._critic(self.output, 'out')
        gan = self.cri_gan.relativistic
        return (gan(on_real, on_fake, False, is_disc=False) + gan(on_fake, on_real, True, is_disc=False)) / 2

    def _critic_phase(self, book):
        fake_img = self.output.detach()
        if not self.relativistic:
            on_real = self._critic(self.gt, 'gt')
            loss_real = self.cri_gan(on_real, True, is_disc=True)
            book.log('l_d_real', loss_real)
            book.log('out_d_real', A.mean(on_real))
            loss_real.backward()
            on_fake = self._critic(fake_img, 'out')
            loss_fake = self.cri_gan(on_fake, False, is_disc=True)
            book.log('l_d_fake', loss_fake)
            book.log('out_d_fake', A.mean(on_fake))
            loss_fake.backward()
            return
        gan = self.cri_gan.relativistic
        with torch.no_grad():   # the reference detaches this forward's result (:65); BatchNorm statistics still move
            fake_const = self._critic(fake_img, 'out')
        on_real = self._critic(self.gt, 'gt')
        loss_real = gan(on_real, fake_const, True, is_disc=True) * 0.5
        loss_real.backward()
        on_fake = self._critic(fake_img, 'out')
        loss_fake = gan(on_fake, on_real.detach(), False, is_disc=True) * 0.5
        loss_fake.backward()
        book.log('l_d_real', loss_real)
        book.log('l_d_fake', loss_fake)
        book.log('out_d_real', A.mean(on_real))
        book.log('out_d_fake', A.mean(on_fake))

    # ------------------------------------------------------------------ one step
    def _generator_phase(self, book, current_iter):
        self.critic.freeze(True)
        self.gen.clear_grads()
        self._prefetch_real_logits(current_iter)
        self.output = self.net_g(self.lq)
        if not self.generator_turn(current_iter):
            return
        self.content_terms(book)
        book.charge('l_g_gan', self._fool_critic())
        book.objective.backward()
        if self.overlap_g_wgrad:
            return True   # G's optimiser step follows the critic phase: its weight gradients are still running beside it
        self.gen.update(self.distributed)

    def optimize_parameters(self, current_iter):
        self.refresh_buffers()
        book = LossBook()
        self._d_kept = {}
        self.d_forwards_run = 0   # distinct discriminator forwards this step issued (5 calls per ESRGAN step)
        g_step_due = self._generator_phase(book, current_iter)
        self.critic.freeze(False)
        self.critic.clear_grads()
        self._critic_phase(book)
        self._d_kept = {}         # D's weights move next: nothing kept survives them
        if g_step_due:
            # The reference steps optimizer_g before the critic phase (esrgan_model.py:48); that phase reads self.output (computed
            # before the step) and D's weights only, so stepping G here gives the same bits — and lets G's weight gradients, which
            # nothing but this step waits for, run under the critic phase on the second lane.
            self.gen.update(self.distributed)
        self.critic.update(self.distributed)
        self.finish_step(book)

    # ------------------------------------------------------------------ files
    def save(self, epoch, current_iter):
        if self.device.type == 'cuda':
            from .. import watchdog
            watchdog.verify('save')   # before ANY file of this checkpoint: net_d_<iter>.pth must not appear behind a timed-out step
        self.save_network(self.net_d, 'net_d', current_iter)
        super().save(epoch, current_iter)
