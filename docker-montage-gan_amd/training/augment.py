"""Adaptive discriminator augmentation pipeline (ADA).

Same class, constructor arguments, buffers (``p``, ``Hz_geom``, ``Hz_fbank``) and -- deliberately -- the same
ORDER and SHAPES of random draws as the reference's ``training/augment.py`` (AugmentPipe :156-482, with the
RGBA branch of the colour stage :403-406), so that a seeded CPU run reproduces its output and
``debug_percentile`` gives the same deterministic images.

It is the main consumer of the separable resampler: the geometric stage is
``pad(reflect) -> upfirdn2d.upsample2d(sym6, x2) -> affine_warp (= affine_grid + grid_sample) -> upfirdn2d.downsample2d(sym6, /2)``
i.e. four launches of csrc/upfirdn2d.hip (two separable passes each) around the native resampler of csrc/augment.hip;
the colour matrix is one native pass (channel_mix) instead of a 3-row batched GEMM.
Host-sync note: upstream reads the data-dependent reflect-padding margin back as python ints (:313), one
device->host synchronisation per discriminator call -- on a GPU-bound step that serialises host and device time.
Two module switches remove that wait (both off by default = upstream behaviour):
  * ``side_stream_margin = True``: the transform parameters and the margin are computed on a separate HIP stream
    that does not queue behind the training step; the host waits only for those few tiny kernels, reads the four
    ints from pinned memory, and the main stream picks the matrices up through an event.  Same random draws, same
    margins, same images -- only the wait shrinks from "the whole backlog" to microseconds.
  * ``fixed_margin = True``: pad by the clamp bound (width-1, height-1) with no read-back at all; same images up to
    float rounding of the grid coordinates, but the 4-channel tensors of the geometric stage get 9x larger.
"""
import numpy as np
import scipy.signal
import torch

from ..torch_utils import misc
from ..torch_utils.ops import conv2d_gradfix
from ..torch_utils.ops import ada_params
from ..torch_utils.ops import affine_warp
from ..torch_utils.ops import channel_mix
from ..torch_utils.ops import grid_sample_gradfix  # noqa: F401  (reference import; affine_warp falls back to it)
from ..torch_utils.ops import upfirdn2d

fixed_margin = False          # True: never read the padding margin back to the host (see module docstring)
device_window = True          # with fixed_margin: the pad + upsample pass computes only the region the device-side margin asks for
side_stream_margin = False    # True: compute transform parameters + margin on a side stream, wait only for that
fused_params = True            # GPU, no debug_percentile: transform parameters from ONE native launch (ops/ada_params.py)
_side_streams = dict()


class _Draws:
    """Where the pipe's random numbers come from: fresh draws in the reference's order and shapes (tables None), or the
    columns of two pre-drawn tables U [n,nu] (uniform) / Z [n,nz] (normal), consumed left to right -- the layout
    ``ops.ada_params`` documents."""

    def __init__(self, device, U=None, Z=None):
        self.device, self.U, self.Z, self.iu, self.iz = device, U, Z, 0, 0

    def _take(self, table, pos, shape):
        k = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        return table[:, pos:pos + k].reshape(shape), pos + k

    def rand(self, *shape):
        if self.U is None:
            return torch.rand(list(shape), device=self.device)
        out, self.iu = self._take(self.U, self.iu, list(shape))
        return out

    def randn(self, *shape):
        if self.Z is None:
            return torch.randn(list(shape), device=self.device)
        out, self.iz = self._take(self.Z, self.iz, list(shape))
        return out

# Orthogonal wavelet low-pass filters (Daubechies / symlets), as tabulated in PyWavelets.
wavelets = {
    'haar': [0.7071067811865476, 0.7071067811865476],
    'db1': [0.7071067811865476, 0.7071067811865476],
    'db2': [-0.12940952255092145, 0.22414386804185735, 0.836516303737469, 0.48296291314469025],
    'db3': [0.035226291882100656, -0.08544127388224149, -0.13501102001039084, 0.4598775021193313, 0.8068915093133388, 0.3326705529509569],
    'db4': [-0.010597401784997278, 0.032883011666982945, 0.030841381835986965, -0.18703481171888114, -0.02798376941698385,
            0.6308807679295904, 0.7148465705525415, 0.23037781330885523],
    'sym2': [-0.12940952255092145, 0.22414386804185735, 0.836516303737469, 0.48296291314469025],
    'sym3': [0.035226291882100656, -0.08544127388224149, -0.13501102001039084, 0.4598775021193313, 0.8068915093133388, 0.3326705529509569],
    'sym4': [-0.07576571478927333, -0.02963552764599851, 0.49761866763201545, 0.8037387518059161, 0.29785779560527736,
             -0.09921954357684722, -0.012603967262037833, 0.0322231006040427],
    'sym5': [0.027333068345077982, 0.029519490925774643, -0.039134249302383094, 0.1993975339773936, 0.7234076904024206,
             0.6339789634582119, 0.01660210576452232, -0.17532808990845047, -0.021101834024758855, 0.019538882735286728],
    'sym6': [0.015404109327027373, 0.0034907120842174702, -0.11799011114819057, -0.048311742585633, 0.4910559419267466,
             0.787641141030194, 0.3379294217276218, -0.07263752278646252, -0.021060292512300564, 0.04472490177066578,
             0.0017677118642428036, -0.007800708325034148],
}


def matrix(*rows, device=None):
    """Rows of python scalars and/or batched tensors -> [..., R, C] (constants broadcast to the batch shape)."""
    assert all(len(row) == len(rows[0]) for row in rows)
    elems = [x for row in rows for x in row]
    ref = [x for x in elems if isinstance(x, torch.Tensor)]
    if not ref:
        return misc.constant(np.asarray(rows), device=device)
    assert device is None or device == ref[0].device
    elems = [x if isinstance(x, torch.Tensor) else misc.constant(x, shape=ref[0].shape, device=ref[0].device) for x in elems]
    return torch.stack(elems, dim=-1).reshape(ref[0].shape + (len(rows), -1))


def translate2d(tx, ty, **kw):
    return matrix([1, 0, tx], [0, 1, ty], [0, 0, 1], **kw)


def translate3d(tx, ty, tz, **kw):
    return matrix([1, 0, 0, tx], [0, 1, 0, ty], [0, 0, 1, tz], [0, 0, 0, 1], **kw)


def scale2d(sx, sy, **kw):
    return matrix([sx, 0, 0], [0, sy, 0], [0, 0, 1], **kw)


def scale3d(sx, sy, sz, **kw):
    return matrix([sx, 0, 0, 0], [0, sy, 0, 0], [0, 0, sz, 0], [0, 0, 0, 1], **kw)


def rotate2d(theta, **kw):
    return matrix([torch.cos(theta), torch.sin(-theta), 0], [torch.sin(theta), torch.cos(theta), 0], [0, 0, 1], **kw)


def rotate3d(v, theta, **kw):
    vx, vy, vz = v[..., 0], v[..., 1], v[..., 2]
    s, c = torch.sin(theta), torch.cos(theta)
    cc = 1 - c
    return matrix([vx * vx * cc + c, vx * vy * cc - vz * s, vx * vz * cc + vy * s, 0],
                  [vy * vx * cc + vz * s, vy * vy * cc + c, vy * vz * cc - vx * s, 0],
                  [vz * vx * cc - vy * s, vz * vy * cc + vx * s, vz * vz * cc + c, 0],
                  [0, 0, 0, 1], **kw)


def translate2d_inv(tx, ty, **kw):
    return translate2d(-tx, -ty, **kw)


def scale2d_inv(sx, sy, **kw):
    return scale2d(1 / sx, 1 / sy, **kw)


def rotate2d_inv(theta, **kw):
    return rotate2d(-theta, **kw)


class AugmentPipe(torch.nn.Module):
    def __init__(self, xflip=0, rotate90=0, xint=0, xint_max=0.125, scale=0, rotate=0, aniso=0, xfrac=0, scale_std=0.2,
                 rotate_max=1, aniso_std=0.2, xfrac_std=0.125, brightness=0, contrast=0, lumaflip=0, hue=0, saturation=0,
                 brightness_std=0.2, contrast_std=0.5, hue_max=1, saturation_std=1, imgfilter=0, imgfilter_bands=[1, 1, 1, 1],
                 imgfilter_std=1, noise=0, cutout=0, noise_std=0.1, cutout_size=0.5):
        super().__init__()
        self.register_buffer('p', torch.ones([]))       # overall multiplier of every probability below
        for k, v in dict(xflip=xflip, rotate90=rotate90, xint=xint, xint_max=xint_max, scale=scale, rotate=rotate, aniso=aniso,
                         xfrac=xfrac, scale_std=scale_std, rotate_max=rotate_max, aniso_std=aniso_std, xfrac_std=xfrac_std,
                         brightness=brightness, contrast=contrast, lumaflip=lumaflip, hue=hue, saturation=saturation,
                         brightness_std=brightness_std, contrast_std=contrast_std, hue_max=hue_max,
                         saturation_std=saturation_std, imgfilter=imgfilter, imgfilter_std=imgfilter_std, noise=noise,
                         cutout=cutout, noise_std=noise_std, cutout_size=cutout_size).items():
            setattr(self, k, float(v))
        self.imgfilter_bands = list(imgfilter_bands)
        self.register_buffer('Hz_geom', upfirdn2d.setup_filter(wavelets['sym6']))       # 12 taps -> separable
        # 4-band filter bank from sym2 for the image-space filter stage (:197-207)
        lo = np.asarray(wavelets['sym2'])
        hi = lo * ((-1) ** np.arange(lo.size))
        lo2, hi2 = np.convolve(lo, lo[::-1]) / 2, np.convolve(hi, hi[::-1]) / 2
        fbank = np.eye(4, 1)
        for i in range(1, fbank.shape[0]):
            fbank = np.dstack([fbank, np.zeros_like(fbank)]).reshape(fbank.shape[0], -1)[:, :-1]
            fbank = scipy.signal.convolve(fbank, [lo2])
            fbank[i, (fbank.shape[1] - hi2.size) // 2:(fbank.shape[1] + hi2.size) // 2] += hi2
        self.register_buffer('Hz_fbank', torch.as_tensor(fbank, dtype=torch.float32))

    # one "maybe apply" draw: value where rand < prob else neutral; debug_percentile overrides both draws
    @staticmethod
    def _gate(value, prob, neutral, shape, draws):
        return torch.where(draws.rand(*shape) < prob, value, neutral)

    def _geometric_params(self, n, height, width, dev, dp, draws=None):
        """Random draws of the blit/geometric stage -> (G_inv [n,3,3] or the identity object, I_3, margin [4] or None)."""
        draws = draws if draws is not None else _Draws(dev)
        rand, randn = draws.rand, draws.randn
        # ---- pixel blitting + general geometric transforms, accumulated as one inverse matrix ----
        I_3 = torch.eye(3, device=dev)
        G_inv = I_3
        if self.xflip > 0:
            i = torch.floor(rand(n) * 2)
            i = self._gate(i, self.xflip * self.p, torch.zeros_like(i), [n], draws)
            if dp is not None:
                i = torch.full_like(i, torch.floor(dp * 2))
            G_inv = G_inv @ scale2d_inv(1 - 2 * i, 1)
        if self.rotate90 > 0:
            i = torch.floor(rand(n) * 4)
            i = self._gate(i, self.rotate90 * self.p, torch.zeros_like(i), [n], draws)
            if dp is not None:
                i = torch.full_like(i, torch.floor(dp * 4))
            G_inv = G_inv @ rotate2d_inv(-np.pi / 2 * i)
        if self.xint > 0:
            t = (rand(n, 2) * 2 - 1) * self.xint_max
            t = self._gate(t, self.xint * self.p, torch.zeros_like(t), [n, 1], draws)
            if dp is not None:
                t = torch.full_like(t, (dp * 2 - 1) * self.xint_max)
            G_inv = G_inv @ translate2d_inv(torch.round(t[:, 0] * width), torch.round(t[:, 1] * height))
        if self.scale > 0:
            s = torch.exp2(randn(n) * self.scale_std)
            s = self._gate(s, self.scale * self.p, torch.ones_like(s), [n], draws)
            if dp is not None:
                s = torch.full_like(s, torch.exp2(torch.erfinv(dp * 2 - 1) * self.scale_std))
            G_inv = G_inv @ scale2d_inv(s, s)
        p_rot = 1 - torch.sqrt((1 - self.rotate * self.p).clamp(0, 1))      # P(pre OR post rotation) = p
        if self.rotate > 0:
            theta = (rand(n) * 2 - 1) * np.pi * self.rotate_max
            theta = self._gate(theta, p_rot, torch.zeros_like(theta), [n], draws)
            if dp is not None:
                theta = torch.full_like(theta, (dp * 2 - 1) * np.pi * self.rotate_max)
            G_inv = G_inv @ rotate2d_inv(-theta)
        if self.aniso > 0:
            s = torch.exp2(randn(n) * self.aniso_std)
            s = self._gate(s, self.aniso * self.p, torch.ones_like(s), [n], draws)
            if dp is not None:
                s = torch.full_like(s, torch.exp2(torch.erfinv(dp * 2 - 1) * self.aniso_std))
            G_inv = G_inv @ scale2d_inv(s, 1 / s)
        if self.rotate > 0:
            theta = (rand(n) * 2 - 1) * np.pi * self.rotate_max
            theta = self._gate(theta, p_rot, torch.zeros_like(theta), [n], draws)
            if dp is not None:
                theta = torch.zeros_like(theta)
            G_inv = G_inv @ rotate2d_inv(-theta)
        if self.xfrac > 0:
            t = randn(n, 2) * self.xfrac_std
            t = self._gate(t, self.xfrac * self.p, torch.zeros_like(t), [n, 1], draws)
            if dp is not None:
                t = torch.full_like(t, torch.erfinv(dp * 2 - 1) * self.xfrac_std)
            G_inv = G_inv @ translate2d_inv(t[:, 0] * width, t[:, 1] * height)

        margin = None
        if G_inv is not I_3:
            # how far do the transformed image corners reach?  -> reflect padding needed
            cx, cy = (width - 1) / 2, (height - 1) / 2
            corners = matrix([-cx, -cy, 1], [cx, -cy, 1], [cx, cy, 1], [-cx, cy, 1], device=dev)
            corners = G_inv @ corners.t()                                       # [batch, xyz, corner]
            hz_pad = self.Hz_geom.shape[0] // 4
            margin = corners[:, :2, :].permute(1, 0, 2).flatten(1)              # [xy, batch*corner]
            margin = torch.cat([-margin, margin]).max(dim=1).values             # [x0, y0, x1, y1]
            margin = margin + misc.constant([hz_pad * 2 - cx, hz_pad * 2 - cy] * 2, device=dev)
            margin = margin.max(misc.constant([0, 0] * 2, device=dev)).min(misc.constant([width - 1, height - 1] * 2, device=dev))
        return G_inv, I_3, margin

    def _color_matrix(self, n, ch, dev, dp, I_4, draws=None):
        """Random draws of the colour stage -> C [n,4,4] acting on (r, g, b, 1), or the identity object I_4."""
        draws = draws if draws is not None else _Draws(dev)
        rand, randn = draws.rand, draws.randn
        C = I_4
        if self.brightness > 0:
            b = randn(n) * self.brightness_std
            b = self._gate(b, self.brightness * self.p, torch.zeros_like(b), [n], draws)
            if dp is not None:
                b = torch.full_like(b, torch.erfinv(dp * 2 - 1) * self.brightness_std)
            C = translate3d(b, b, b) @ C
        if self.contrast > 0:
            c = torch.exp2(randn(n) * self.contrast_std)
            c = self._gate(c, self.contrast * self.p, torch.ones_like(c), [n], draws)
            if dp is not None:
                c = torch.full_like(c, torch.exp2(torch.erfinv(dp * 2 - 1) * self.contrast_std))
            C = scale3d(c, c, c) @ C
        v = misc.constant(np.asarray([1, 1, 1, 0]) / np.sqrt(3), device=dev)       # luma axis
        vv = torch.outer(v, v)
        if self.lumaflip > 0:
            i = torch.floor(rand(n, 1, 1) * 2)
            i = self._gate(i, self.lumaflip * self.p, torch.zeros_like(i), [n, 1, 1], draws)
            if dp is not None:
                i = torch.full_like(i, torch.floor(dp * 2))
            C = (I_4 - 2 * vv * i) @ C                                             # Householder reflection
        if self.hue > 0 and ch > 1:
            theta = (rand(n) * 2 - 1) * np.pi * self.hue_max
            theta = self._gate(theta, self.hue * self.p, torch.zeros_like(theta), [n], draws)
            if dp is not None:
                theta = torch.full_like(theta, (dp * 2 - 1) * np.pi * self.hue_max)
            C = rotate3d(v, theta) @ C
        if self.saturation > 0 and ch > 1:
            s = torch.exp2(randn(n, 1, 1) * self.saturation_std)
            s = self._gate(s, self.saturation * self.p, torch.ones_like(s), [n, 1, 1], draws)
            if dp is not None:
                s = torch.full_like(s, torch.exp2(torch.erfinv(dp * 2 - 1) * self.saturation_std))
            C = (vv + (I_4 - vv) * s) @ C
        return C

    def _sampling_factors(self, n, ch, height, width, margins):
        """The host-known factors around G_inv (upstream chains six small matrix products here, :365-375: shift by the
        padding, x2 scale, half-pixel shift, normalisation to [-1,1]) multiplied on the host in the same order (float32)
        into ONE left and ONE right factor; also the padded and the sampled sizes."""
        mx0, my0, mx1, my1 = margins
        hz_pad = self.Hz_geom.shape[0] // 4
        pad_h, pad_w = (height + my0 + my1) * 2, (width + mx0 + mx1) * 2          # size after reflect padding and x2 upsampling
        shape = [n, ch, (height + hz_pad * 2) * 2, (width + hz_pad * 2) * 2]
        f32 = np.float32

        def T(tx, ty):
            return np.asarray([[1, 0, tx], [0, 1, ty], [0, 0, 1]], dtype=f32)

        def S(sx, sy):
            return np.asarray([[sx, 0, 0], [0, sy, 0], [0, 0, 1]], dtype=f32)
        left = S(2 / pad_w, 2 / pad_h) @ (T(-0.5, -0.5) @ (S(2, 2) @ T((mx0 - mx1) / 2, (my0 - my1) / 2)))
        right = (S(1 / 2, 1 / 2) @ T(0.5, 0.5)) @ S(1 / (2 / shape[3]), 1 / (2 / shape[2]))
        return left, right, shape

    def params_from_tables(self, U, Z, n, ch, height, width):
        """The python formulation of the transform parameters driven by pre-drawn tables (what ``ops.ada_params`` computes in
        one native launch): -> (G_inv or None, margin or None, C or None)."""
        dev = U.device
        draws = _Draws(dev, U, Z)
        G_inv, I_3, margin = self._geometric_params(n, height, width, dev, None, draws)
        I_4 = torch.eye(4, device=dev)
        C = self._color_matrix(n, ch, dev, None, I_4, draws)
        return (None if G_inv is I_3 else G_inv), margin, (None if C is I_4 else C)

    def forward(self, images, debug_percentile=None):
        assert isinstance(images, torch.Tensor) and images.ndim == 4
        n, ch, height, width = images.shape
        dev = images.device
        dp = None if debug_percentile is None else torch.as_tensor(debug_percentile, dtype=torch.float32, device=dev)
        draws = _Draws(dev)
        rand, randn = draws.rand, draws.randn
        hz_pad = self.Hz_geom.shape[0] // 4
        # MI355X path: all ~13 random decisions of the blit / geometric / colour stages and the matrices built from them
        # in ONE native launch from two tables of pre-drawn numbers (ops/ada_params.py).  debug_percentile and CPU tensors
        # keep the reference's draw-by-draw formulation below (same order and shapes of draws as upstream).
        use_native = (fused_params and images.is_cuda and dp is None and ch in (1, 3, 4)
                      and (ada_params.has_geometric(self) or ada_params.has_color(self, ch)))
        native_out = None

        def native_params(left=None, right=None):
            U, Z = torch.rand([n, ada_params.NU], device=dev), torch.randn([n, ada_params.NZ], device=dev)
            return ada_params.ada_params(self, U, Z, width, height, ch, left, right)

        # ---- pixel blitting + general geometric transforms (one inverse matrix) and the padding they need ----
        use_side = side_stream_margin and images.is_cuda and not fixed_margin
        fixed = (width - 1, height - 1, width - 1, height - 1)                   # the clamp bound: no read-back
        theta = None
        if use_native and fixed_margin:
            left, right, shape = self._sampling_factors(n, ch, height, width, fixed)
            native_out = native_params(left, right)
            G_inv, I_3, margin, theta = native_out['G_inv'], None, native_out['margin'], native_out['theta']
        elif use_side:
            main = torch.cuda.current_stream(dev)
            side = _side_streams.get(dev)
            if side is None:
                side = _side_streams[dev] = torch.cuda.Stream(device=dev, priority=-1)
            if not hasattr(self, '_pinned_margin'):
                self._pinned_margin = torch.empty([4], dtype=torch.int32).pin_memory()
            with torch.cuda.stream(side):
                if use_native:
                    native_out = native_params()
                    G_inv, I_3, margin = native_out['G_inv'], None, native_out['margin']
                    for v in native_out.values():
                        if v is not None:
                            v.record_stream(main)
                else:
                    G_inv, I_3, margin = self._geometric_params(n, height, width, dev, dp, draws)
                if margin is not None:
                    self._pinned_margin.copy_(margin.ceil().to(torch.int32), non_blocking=True)
                done = torch.cuda.Event()
                done.record(side)
            done.synchronize()                  # the host waits for the side stream only
            main.wait_event(done)
            if G_inv is not I_3 and not use_native:
                G_inv.record_stream(main)
        elif use_native:
            native_out = native_params()
            G_inv, I_3, margin = native_out['G_inv'], None, native_out['margin']
        else:
            G_inv, I_3, margin = self._geometric_params(n, height, width, dev, dp, draws)
        if G_inv is not I_3:
            if fixed_margin:
                margins = fixed
            elif use_side:
                margins = tuple(int(v) for v in self._pinned_margin.tolist())             # already on the host
            else:
                margins = tuple(int(v) for v in margin.ceil().to(torch.int32).tolist())   # the one host sync
            mx0, my0, mx1, my1 = margins
            # reflect-pad + x2 upsample with the 12-tap symlet (:315-317): one native launch, the padded image is never written
            # fixed margin: the padding is the clamp bound, the WORK the margin computed on the device (`margin` never leaves it)
            images = upfirdn2d.reflect_pad_upsample2d(images, self.Hz_geom, [mx0, mx1, my0, my1],
                                                      window=margin if (fixed_margin and device_window and margin is not None and images.is_cuda) else None)
            if theta is None:
                # sample on the output lattice: left @ G_inv @ right
                left, right, shape = self._sampling_factors(n, ch, height, width, margins)
                if fixed_margin:        # constants of the CONFIGURATION: cached on the device (keeps the call recordable into a HIP graph)
                    lr = misc.constant(np.stack([left, right]), device=dev)
                else:                   # depend on this call's margin: uploaded without a blocking copy
                    lr = misc.upload_small(np.stack([left, right]), dev)
                theta = (lr[0] @ G_inv @ lr[1])[:, :2, :]
            images = affine_warp.affine_warp(images, theta, shape[2], shape[3])     # affine_grid + grid_sample, one pass
            # back down, cropping the filter margin
            images = upfirdn2d.downsample2d(x=images, f=self.Hz_geom, down=2, padding=-hz_pad * 2, flip_filter=True)

        # ---- colour transforms, accumulated as one 4x4 matrix on (r, g, b, 1) ----
        I_4 = None
        if native_out is not None:
            C = native_out['C']
        else:
            I_4 = torch.eye(4, device=dev)
            C = self._color_matrix(n, ch, dev, dp, I_4, draws)
        if C is not I_4 and C is not None:
            flat = images.reshape([n, ch, height * width])
            if ch == 4 and native_out is not None:
                flat = channel_mix.channel_mix(flat, native_out['A4'], native_out['t4'])
            elif ch == 4:       # RGBA: colour matrix on RGB, alpha untouched (reference :403-406)
                A4 = I_4.repeat([n, 1, 1])
                A4[:, :3, :3] = C[:, :3, :3]
                flat = channel_mix.channel_mix(flat, A4, torch.nn.functional.pad(C[:, :3, 3], [0, 1]))
            elif ch == 3:
                flat = channel_mix.channel_mix(flat, C[:, :3, :3], C[:, :3, 3])
            elif ch == 1:
                Cm = C[:, :3, :].mean(dim=1, keepdims=True)
                flat = flat * Cm[:, :, :3].sum(dim=2, keepdims=True) + Cm[:, :, 3:]
            else:
                raise ValueError('Image must be RGBA (4 channels), RGB (3 channels) or L (1 channel)')
            images = flat.reshape([n, ch, height, width])

        # ---- image-space filtering ----
        if self.imgfilter > 0:
            bands = self.Hz_fbank.shape[0]
            assert len(self.imgfilter_bands) == bands
            expected_power = misc.constant(np.array([10, 1, 1, 1]) / 13, device=dev)       # 1/f spectrum
            g = torch.ones([n, bands], device=dev)
            for i, strength in enumerate(self.imgfilter_bands):
                t_i = torch.exp2(randn(n) * self.imgfilter_std)
                t_i = self._gate(t_i, self.imgfilter * self.p * strength, torch.ones_like(t_i), [n], draws)
                if dp is not None:
                    t_i = torch.full_like(t_i, torch.exp2(torch.erfinv(dp * 2 - 1) * self.imgfilter_std)) if strength > 0 else torch.ones_like(t_i)
                t = torch.ones([n, bands], device=dev)
                t[:, i] = t_i
                t = t / (expected_power * t.square()).sum(dim=-1, keepdims=True).sqrt()
                g = g * t
            taps = (g @ self.Hz_fbank).unsqueeze(1).repeat([1, ch, 1]).reshape([n * ch, 1, -1])
            pad = self.Hz_fbank.shape[1] // 2
            images = images.reshape([1, n * ch, height, width])
            images = torch.nn.functional.pad(input=images, pad=[pad, pad, pad, pad], mode='reflect')
            images = conv2d_gradfix.conv2d(input=images, weight=taps.unsqueeze(2), groups=n * ch)
            images = conv2d_gradfix.conv2d(input=images, weight=taps.unsqueeze(3), groups=n * ch)
            images = images.reshape([n, ch, height, width])

        # ---- image-space corruptions ----
        if self.noise > 0:
            sigma = randn(n, 1, 1, 1).abs() * self.noise_std
            sigma = self._gate(sigma, self.noise * self.p, torch.zeros_like(sigma), [n, 1, 1, 1], draws)
            if dp is not None:
                sigma = torch.full_like(sigma, torch.erfinv(dp) * self.noise_std)
            images = images + randn(n, ch, height, width) * sigma
        if self.cutout > 0:
            size = torch.full([n, 2, 1, 1, 1], self.cutout_size, device=dev)
            size = self._gate(size, self.cutout * self.p, torch.zeros_like(size), [n, 1, 1, 1, 1], draws)
            center = rand(n, 2, 1, 1, 1)
            if dp is not None:
                size = torch.full_like(size, self.cutout_size)
                center = torch.full_like(center, dp)
            cx_ = torch.arange(width, device=dev).reshape([1, 1, 1, -1])
            cy_ = torch.arange(height, device=dev).reshape([1, 1, -1, 1])
            mask_x = ((cx_ + 0.5) / width - center[:, 0]).abs() >= size[:, 0] / 2
            mask_y = ((cy_ + 0.5) / height - center[:, 1]).abs() >= size[:, 1] / 2
            images = images * torch.logical_or(mask_x, mask_y).to(torch.float32)
        return images


augpipe_specs = {       # train.py:245-259 / train_aio.py augpipe table
    'blit': dict(xflip=1, rotate90=1, xint=1),
    'geom': dict(scale=1, rotate=1, aniso=1, xfrac=1),
    'color': dict(brightness=1, contrast=1, lumaflip=1, hue=1, saturation=1),
    'filter': dict(imgfilter=1),
    'noise': dict(noise=1),
    'cutout': dict(cutout=1),
    'bg': dict(xflip=1, rotate90=1, xint=1, scale=1, rotate=1, aniso=1, xfrac=1),
    'bgc': dict(xflip=1, rotate90=1, xint=1, scale=1, rotate=1, aniso=1, xfrac=1, brightness=1, contrast=1, lumaflip=1, hue=1,
                saturation=1),
}
