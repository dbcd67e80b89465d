"""Non-saturating logistic GAN loss with R1 and path-length regularisation.

Same class, constructor arguments and phase semantics as the reference's
``training/loss.py`` (StyleGAN2Loss, :25-139); its aio sibling in ``custom/loss_aio.py``
subclasses this one.  Phases: Gmain / Greg (path length) / Dmain / Dreg (R1), or the
``*both`` fusions.  On the CPU random draws happen in the same order and with the same shapes
as upstream, so a CPU run from the same seed reproduces the reference's numbers (the golden
tests rely on it).  On the GPU two draws are batched -- the style-mixing partner is mapped
together with z, and the synthesis network draws the noise of all its layers at once
(training/networks.py) --: same distributions, other positions of the generator's sequence.

MI355X notes: the style-mixing splice is a masked select instead of ``ws[:, cutoff:] = ...``
(upstream indexes with a device scalar, which is a host sync per generator pass), and the
regularisers run under ``conv2d_gradfix.no_weight_gradients()`` which this build's gradfix
honours again.
"""
import numpy as np
import torch

from ..torch_utils import misc
from ..torch_utils import training_stats
from ..torch_utils.ops import conv2d_gradfix


class Loss:
    def accumulate_gradients(self, phase, real_img, gen_z, sync, gain, real_c=None, gen_c=None):
        raise NotImplementedError()


def _splice_after(ws, other, cutoff):
    """ws with ws[:, cutoff:] replaced by other[:, cutoff:], cutoff a 0-dim device tensor; no host sync."""
    keep = torch.arange(ws.shape[1], device=ws.device) < cutoff
    return torch.where(keep.reshape(1, -1, 1), ws, other)


class StyleGAN2Loss(Loss):
    stat_prefix = ''        # the aio variant prefixes statistics with '<layer_name>/'

    def __init__(self, device, G_mapping, G_synthesis, D, augment_pipe=None, style_mixing_prob=0.9, r1_gamma=10,
                 pl_batch_shrink=2, pl_decay=0.01, pl_weight=2):
        super().__init__()
        self.device = device
        self.G_mapping, self.G_synthesis, self.D = G_mapping, G_synthesis, D
        self.augment_pipe = augment_pipe
        self.style_mixing_prob, self.r1_gamma = style_mixing_prob, r1_gamma
        self.pl_batch_shrink, self.pl_decay, self.pl_weight = pl_batch_shrink, pl_decay, pl_weight
        self.pl_mean = torch.zeros([], device=device)

    # -- pieces -----------------------------------------------------------------------------
    def _map(self, z, c, **kw):
        return self.G_mapping(z, c, **kw)

    def run_G(self, z, c, sync):
        with misc.ddp_sync(self.G_mapping, sync):
            if self.style_mixing_prob > 0:
                with torch.autograd.profiler.record_function('style_mixing'):
                    # upstream maps z, then draws cutoff, the mixing coin and the partner latent, then maps the partner
                    # (:66-72).  The mapping network draws nothing, so the same draws in the same order can come first
                    # and both latents go through the network as ONE batch of 2N independent rows: half the launches.
                    n, num_ws = z.shape[0], self.G_mapping.num_ws
                    cutoff = torch.empty([], dtype=torch.int64, device=z.device).random_(1, num_ws)
                    cutoff = torch.where(torch.rand([], device=z.device) < self.style_mixing_prob, cutoff,
                                         torch.full_like(cutoff, num_ws))
                    both = self._map(torch.cat([z, torch.randn_like(z)]), None if c is None else torch.cat([c, c]), w_avg_rows=n)
                    ws = _splice_after(both[:n], both[n:], cutoff)
            else:
                ws = self._map(z, c)
        with misc.ddp_sync(self.G_synthesis, sync):
            ws = ws[:, :self.G_synthesis.num_ws]
            img = self.G_synthesis(ws)
        return img, ws

    def run_D(self, img, c, sync):
        if self.augment_pipe is not None:
            img = self.augment_pipe(img)
        with misc.ddp_sync(self.D, sync):
            return self.D(img, c)

    def _report(self, name, value):
        return training_stats.report(self.stat_prefix + name, value)

    # -- phases -----------------------------------------------------------------------------
    def accumulate_gradients(self, phase, real_img, gen_z, sync, gain, real_c=None, gen_c=None):
        assert phase in ['Gmain', 'Greg', 'Gboth', 'Dmain', 'Dreg', 'Dboth']
        do_Gmain = phase in ['Gmain', 'Gboth']
        do_Dmain = phase in ['Dmain', 'Dboth']
        do_Gpl = phase in ['Greg', 'Gboth'] and self.pl_weight != 0
        do_Dr1 = phase in ['Dreg', 'Dboth'] and self.r1_gamma != 0
        rec = torch.autograd.profiler.record_function

        if do_Gmain:    # maximise logits of generated images
            with rec('Gmain_forward'):
                gen_img, _ = self.run_G(gen_z, gen_c, sync=(sync and not do_Gpl))
                gen_logits = self.run_D(gen_img, gen_c, sync=False)
                self._report('Loss/scores/fake', gen_logits)
                self._report('Loss/signs/fake', gen_logits.sign())
                loss_Gmain = torch.nn.functional.softplus(-gen_logits)
                self._report('Loss/G/loss', loss_Gmain)
            with rec('Gmain_backward'):
                loss_Gmain.mean().mul(gain).backward()

        if do_Gpl:      # path-length regularisation on a shrunken batch
            with rec('Gpl_forward'):
                n = gen_z.shape[0] // self.pl_batch_shrink
                gen_img, gen_ws = self.run_G(gen_z[:n], gen_c[:n] if gen_c is not None else None, sync=sync)
                pl_noise = torch.randn_like(gen_img) / np.sqrt(gen_img.shape[2] * gen_img.shape[3])
                with rec('pl_grads'), conv2d_gradfix.no_weight_gradients():
                    pl_grads, = torch.autograd.grad(outputs=[(gen_img * pl_noise).sum()], inputs=[gen_ws],
                                                    create_graph=True, only_inputs=True)
                pl_lengths = pl_grads.square().sum(2).mean(1).sqrt()
                pl_mean = self.pl_mean.lerp(pl_lengths.mean(), self.pl_decay)
                self.pl_mean.copy_(pl_mean.detach())
                pl_penalty = (pl_lengths - pl_mean).square()
                self._report('Loss/pl_penalty', pl_penalty)
                loss_Gpl = pl_penalty * self.pl_weight
                self._report('Loss/G/reg', loss_Gpl)
            with rec('Gpl_backward'):
                (gen_img[:, 0, 0, 0] * 0 + loss_Gpl).mean().mul(gain).backward()

        loss_Dgen = 0
        if do_Dmain:    # minimise logits of generated images
            with rec('Dgen_forward'):
                gen_img, _ = self.run_G(gen_z, gen_c, sync=False)
                gen_logits = self.run_D(gen_img, gen_c, sync=False)      # synchronised by the real pass below
                self._report('Loss/scores/fake', gen_logits)
                self._report('Loss/signs/fake', gen_logits.sign())
                loss_Dgen = torch.nn.functional.softplus(gen_logits)
            with rec('Dgen_backward'):
                loss_Dgen.mean().mul(gain).backward()

        if do_Dmain or do_Dr1:      # maximise logits of real images; R1 penalty on them
            name = 'Dreal_Dr1' if do_Dmain and do_Dr1 else 'Dreal' if do_Dmain else 'Dr1'
            with rec(name + '_forward'):
                real_img_tmp = real_img.detach().requires_grad_(do_Dr1)
                real_logits = self.run_D(real_img_tmp, real_c, sync=sync)
                self._report('Loss/scores/real', real_logits)
                self._report('Loss/signs/real', real_logits.sign())
                loss_Dreal = 0
                if do_Dmain:
                    loss_Dreal = torch.nn.functional.softplus(-real_logits)
                    self._report('Loss/D/loss', loss_Dgen + loss_Dreal)
                loss_Dr1 = 0
                if do_Dr1:
                    with rec('r1_grads'), conv2d_gradfix.no_weight_gradients():
                        r1_grads, = torch.autograd.grad(outputs=[real_logits.sum()], inputs=[real_img_tmp],
                                                        create_graph=True, only_inputs=True)
                    r1_penalty = r1_grads.square().sum([1, 2, 3])
                    loss_Dr1 = r1_penalty * (self.r1_gamma / 2)
                    self._report('Loss/r1_penalty', r1_penalty)
                    self._report('Loss/D/reg', loss_Dr1)
            with rec(name + '_backward'):
                (real_logits * 0 + loss_Dreal + loss_Dr1).mean().mul(gain).backward()
