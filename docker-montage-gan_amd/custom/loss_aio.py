"""Losses of the all-in-one MontageGAN step: L local StyleGAN2 losses + the global GAN + the
renderer phase.  Same classes, constructor arguments and phase names as the reference's
``custom/loss_aio.py`` (StyleGAN2Loss :18-179, theta_constrain_loss :182-188, MontageGANLoss
:191-372):

    local_{Gmain,Greg,Gboth,Dmain,Dreg,Dboth}_<layer_name>
    global_{Gmain,Greg,Gboth,Dmain,Dreg,Dboth}
    Renderer

Quirks preserved on purpose: ``global_Greg`` does nothing (B-5); the global R1
penalty sums dims [1,2,3] of the 5-D gradient and reports under un-prefixed names (B-6).
On the MI355X path the global D input comes from the one-kernel ``alpha_composite`` and the
layers are placed by ``translate_sample`` inside the STN.
"""
from typing import List, Optional

import torch

from ..custom_utils.image_utils import (alpha_composite_pytorch, convert_translate_to_2x3,
                                        make_batch_for_pos_estimator, normalize_minus11, normalize_zero1)
from ..torch_utils import misc
from ..torch_utils import training_stats
from ..torch_utils.ops import conv2d_gradfix
from ..training import loss as base_loss


class Loss(base_loss.Loss):
    pass


class StyleGAN2Loss(base_loss.StyleGAN2Loss):
    """Local GAN `layer_index` of the stack.  All local generators read the same z through a shared
    mapping network; with a GlobalMappingNetwork ([B,L,num_ws,w]) each takes its own slice."""

    def __init__(self, layer_name, layer_index, device, mapping_network, local_G, local_D, augment_pipe=None,
                 style_mixing_prob=0.9, r1_gamma=10, pl_batch_shrink=2, pl_decay=0.01, pl_weight=2):
        super().__init__(device=device, G_mapping=mapping_network, G_synthesis=local_G, D=local_D,
                         augment_pipe=augment_pipe, style_mixing_prob=style_mixing_prob, r1_gamma=r1_gamma,
                         pl_batch_shrink=pl_batch_shrink, pl_decay=pl_decay, pl_weight=pl_weight)
        self.layer_name, self.layer_index = layer_name, layer_index
        self.stat_prefix = f'{layer_name}/'

    # aliases with the reference's attribute names
    @property
    def mapping_network(self):
        return self.G_mapping

    @property
    def local_G(self):
        return self.G_synthesis

    @property
    def local_D(self):
        return self.D

    def _map(self, z, c, **kw):
        ws = self.G_mapping(z, c, **kw)
        return ws[:, self.layer_index] if ws.ndim == 4 else ws


def theta_constrain_loss(theta: torch.Tensor, device):
    """|| theta - clamp(theta, +-bound) ||_2 with bound = translation +-1 (reference :182-188).
    Only the translation column can leave the box; the linear part is the constant identity."""
    img_layers = theta.shape[-3]
    upper = convert_translate_to_2x3(torch.ones([img_layers, 2], device=device))
    lower = convert_translate_to_2x3(-torch.ones([img_layers, 2], device=device))
    return torch.norm(theta - torch.max(torch.min(theta, upper), lower), 2)


class MontageGANLoss:
    def __init__(self, device, global_d_real_use_renderer, renderer_retrain_use_real,
                 mapping_network: torch.nn.Module, layer_names: List[str], local_G_list: List[torch.nn.Module],
                 augment_pipe_list: List[Optional[torch.nn.Module]], local_D_list: List[torch.nn.Module],
                 pos_estimator: Optional[torch.nn.Module] = None, renderer: Optional[torch.nn.Module] = None,
                 criterion_renderer: Optional[torch.nn.Module] = None,
                 global_augment_pipe: Optional[torch.nn.Module] = None, global_D: Optional[torch.nn.Module] = None,
                 global_r1_gamma=10, **sg2loss_kwargs):
        assert len(local_G_list) == len(local_D_list) == len(augment_pipe_list) == len(layer_names)
        self.device = device
        self.global_d_real_use_renderer = global_d_real_use_renderer
        self.renderer_retrain_use_real = renderer_retrain_use_real
        self.layer_names = layer_names
        self.local_GAN_loss_list = [
            StyleGAN2Loss(name, idx, device, mapping_network, g, d, aug, **sg2loss_kwargs)
            for idx, (name, g, d, aug) in enumerate(zip(layer_names, local_G_list, local_D_list, augment_pipe_list))]
        self.pos_estimator, self.renderer, self.criterion_renderer = pos_estimator, renderer, criterion_renderer
        self.global_augment_pipe, self.global_D, self.global_r1_gamma = global_augment_pipe, global_D, global_r1_gamma

    def run_global_G(self, z, c, sync):
        outs = [loss.run_G(z, c, sync)[0] for loss in self.local_GAN_loss_list]
        fake_layer = make_batch_for_pos_estimator(outs, pad_value=-1)          # [B,L,C,256,256] in [-1,1]
        with misc.ddp_sync(self.pos_estimator, sync):
            return self.pos_estimator(fake_layer)                               # (placed layers, theta)

    def run_global_D(self, blchw, c, sync, use_renderer=True):
        if self.renderer is not None and use_renderer:
            blended = self.renderer(blchw)
        else:
            blended = normalize_minus11(alpha_composite_pytorch(normalize_zero1(blchw)))
        if self.global_augment_pipe is not None:
            blended = self.global_augment_pipe(blended)
        with misc.ddp_sync(self.global_D, sync):
            return self.global_D(blended, c)

    def accumulate_gradients(self, phase, real_blchw, real_list_of_bchw, gen_z, sync, gain, real_c=None, gen_c=None):
        rec = torch.autograd.profiler.record_function
        if phase.startswith('local_'):
            _, base_phase, *rest = phase.split('_')
            layer_name = '_'.join(rest)
            assert base_phase in ['Gmain', 'Greg', 'Gboth', 'Dmain', 'Dreg', 'Dboth'] and layer_name in self.layer_names
            idx = self.layer_names.index(layer_name)
            self.local_GAN_loss_list[idx].accumulate_gradients(base_phase, real_list_of_bchw[idx].to(self.device), gen_z,
                                                               sync, gain, real_c, gen_c)
            return

        if phase.startswith('global_'):
            _, base_phase = phase.split('_')
            assert base_phase in ['Gmain', 'Greg', 'Gboth', 'Dmain', 'Dreg', 'Dboth']
            do_Gmain = base_phase in ['Gmain', 'Gboth']
            do_Dmain = base_phase in ['Dmain', 'Dboth']
            do_Dr1 = base_phase in ['Dreg', 'Dboth'] and self.global_r1_gamma != 0

            if do_Gmain:
                with rec('Gmain_forward_global'):
                    placed, theta = self.run_global_G(gen_z, gen_c, sync=sync)
                    gen_logits = self.run_global_D(placed, gen_c, sync=False)
                    training_stats.report('global/Loss/scores/fake', gen_logits)
                    training_stats.report('global/Loss/signs/fake', gen_logits.sign())
                    loss_Gmain = torch.nn.functional.softplus(-gen_logits)
                    training_stats.report('global/Loss/G/loss', loss_Gmain)
                    loss_theta = theta_constrain_loss(theta, self.device)
                    training_stats.report('global/Loss/STN/theta_constrain', loss_theta)
                with rec('Gmain_backward_global'):
                    # two backward calls upstream (retain_graph); gradients add, so one call on the sum is identical
                    (loss_Gmain.mean().mul(gain) + loss_theta.mul(gain)).backward()

            loss_Dgen = 0
            if do_Dmain:
                with rec('Dgen_forward_global'):
                    placed, _ = self.run_global_G(gen_z, gen_c, sync=False)
                    gen_logits = self.run_global_D(placed, gen_c, sync=False)
                    training_stats.report('global/Loss/scores/fake', gen_logits)
                    training_stats.report('global/Loss/signs/fake', gen_logits.sign())
                    loss_Dgen = torch.nn.functional.softplus(gen_logits)
                with rec('Dgen_backward_global'):
                    loss_Dgen.mean().mul(gain).backward()

            if do_Dmain or do_Dr1:
                name = 'Dreal_Dr1' if do_Dmain and do_Dr1 else 'Dreal' if do_Dmain else 'Dr1'
                with rec(name + '_forward_global'):
                    real_tmp = real_blchw.detach().requires_grad_(do_Dr1)
                    real_logits = self.run_global_D(real_tmp, real_c, sync=sync, use_renderer=self.global_d_real_use_renderer)
                    training_stats.report('global/Loss/scores/real', real_logits)
                    training_stats.report('global/Loss/signs/real', real_logits.sign())
                    loss_Dreal = 0
                    if do_Dmain:
                        loss_Dreal = torch.nn.functional.softplus(-real_logits)
                        training_stats.report('global/Loss/D/loss', loss_Dgen + loss_Dreal)
                    loss_Dr1 = 0
                    if do_Dr1:
                        with rec('r1_grads_global'), conv2d_gradfix.no_weight_gradients():
                            r1_grads, = torch.autograd.grad(outputs=[real_logits.sum()], inputs=[real_tmp],
                                                            create_graph=True, only_inputs=True)
                        r1_penalty = r1_grads.square().sum([1, 2, 3])       # [B, W] on the 5-D gradient, as upstream
                        loss_Dr1 = r1_penalty * (self.global_r1_gamma / 2)
                        training_stats.report('Loss/r1_penalty', r1_penalty)
                        training_stats.report('Loss/D/reg', loss_Dr1)
                with rec(name + '_backward_global'):
                    (real_logits * 0 + loss_Dreal + loss_Dr1).mean().mul(gain).backward()
            return

        assert phase == 'Renderer'
        assert self.renderer is not None and self.criterion_renderer is not None
        with rec('renderer_forward'):
            x, _ = self.run_global_G(gen_z, gen_c, sync=False)
            out = self.renderer(x)
            # upstream's target is the PIL (uint8) composite computed on the host per sample
            # (image_utils.py:74-96); the device composite is the same function without the 8-bit
            # quantisation (<= 1.4e-2 apart) and keeps the phase on the GPU.
            target = alpha_composite_pytorch(normalize_zero1(x.detach()))
            loss_gen = self.criterion_renderer(normalize_zero1(out), target)
            training_stats.report('Renderer/loss_gen', loss_gen.detach())
            loss_real = None
            if self.renderer_retrain_use_real:
                out_r = self.renderer(real_blchw)
                target_r = alpha_composite_pytorch(normalize_zero1(real_blchw.detach()))
                loss_real = self.criterion_renderer(normalize_zero1(out_r), target_r)
                training_stats.report('Renderer/loss_real', loss_real.detach())
        with rec('renderer_backward'):
            loss_gen.backward()
            if loss_real is not None:
                loss_real.backward()
