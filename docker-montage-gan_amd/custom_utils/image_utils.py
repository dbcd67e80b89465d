"""Tensor-layout helpers of the MontageGAN step, restated to run on the device without per-sample
python loops or host synchronisation.

Same function names and argument meaning as the reference's ``custom_utils/image_utils.py``
(the PIL / torchvision helpers for saving and viewing images are out of scope):

    alpha_composite_pytorch   :112-163  -> one HIP kernel (torch_utils/ops/alpha_composite.py)
    normalize_minus11 / zero1 :181-195
    bounding_box, crop_to_content, pad_256  :198-226
    make_batch_for_pos_estimator :229-243
    make_batch_for_local_d, generate_pseudo_fake :246-291
    convert_translate_to_2x3  :316-335
    stack/unstack_layer_to_channel, calc_psnr, assert_range_zero1

Naming (as upstream): blchw = [B,L,C,H,W], lchw = [L,C,H,W], bchw = [B,C,H,W], chw = [C,H,W].
"""
import math

import torch
import torch.nn.functional as F

from ..torch_utils.ops import alpha_composite as _ac
from ..torch_utils.ops import translate_sample as _ts


def alpha_composite_pytorch(blchw_lchw, use_premultiplied=False):
    """Differentiable bottom-up "over" composite, range [0,1] (reference :112-163, default branch;
    the premultiplied branch discards its result upstream -- SURVEY App. B-2 -- and is not offered)."""
    assert not use_premultiplied, 'the premultiplied branch is dead code upstream'
    return _ac.alpha_composite(blchw_lchw)


def normalize_minus11(tensor):
    """[0,1] -> [-1,1]."""
    return tensor * 2. - 1.


def normalize_zero1(tensor):
    """[-1,1] -> [0,1]  (the reference's docstring says the opposite)."""
    return (tensor + 1.) / 2.


def assert_range_zero1(tensor):
    _min, _max = torch.min(tensor), torch.max(tensor)
    assert _min >= 0 and _max <= 1


def bounding_box(mask):
    """mask: [K,2] (y,x) indices of non-zero pixels -> (x_min, y_min, x_max, y_max)  (reference :198-207)."""
    y_min, x_min = mask.min(dim=0)[0]
    y_max, x_max = mask.max(dim=0)[0]
    return x_min, y_min, x_max, y_max


def content_boxes(alpha):
    """Vectorised bounding boxes: alpha [M,H,W] -> int64 (x1, y1, x2, y2) each [M], all zero for an
    empty layer -- what ``bounding_box(torch.nonzero(alpha))`` gives per image upstream."""
    m, h, w = alpha.shape
    nz = alpha != 0
    cols, rows = nz.any(dim=1), nz.any(dim=2)                           # [M,W], [M,H]
    xs = torch.arange(w, device=alpha.device).expand(m, w)
    ys = torch.arange(h, device=alpha.device).expand(m, h)
    big = max(h, w)
    x1 = torch.where(cols, xs, torch.full_like(xs, big)).min(dim=1)[0]
    x2 = torch.where(cols, xs, torch.full_like(xs, -1)).max(dim=1)[0]
    y1 = torch.where(rows, ys, torch.full_like(ys, big)).min(dim=1)[0]
    y2 = torch.where(rows, ys, torch.full_like(ys, -1)).max(dim=1)[0]
    empty = ~cols.any(dim=1)
    zero = torch.zeros_like(x1)
    return (torch.where(empty, zero, x1), torch.where(empty, zero, y1), torch.where(empty, zero, x2),
            torch.where(empty, zero, y2))


def crop_to_content(img_tensor):
    """chw RGBA -> the crop [:, y1:y2, x1:x2] of the alpha bounding box (reference :210-214; the max
    row/column is excluded, as upstream's slice does).  Host-synchronising; the batched path below is not."""
    x1, y1, x2, y2 = (int(v[0]) for v in content_boxes(img_tensor[3][None]))
    return img_tensor[:, y1:y2, x1:x2]


def pad_256(img_tensor, pad_value=0):
    """Centre chw in a 256x256 canvas filled with pad_value (reference :216-226)."""
    _, h, w = img_tensor.shape
    pad_x, pad_y = 256 - w, 256 - h
    px1, py1 = pad_x // 2, pad_y // 2
    return F.pad(img_tensor, (px1, pad_x - px1, py1, pad_y - py1), mode='constant', value=pad_value)


def make_batch_for_pos_estimator(list_of_bchw, pad_value=0):
    """list (one per layer) of [B,C,h,w] -> [B,L,C,256,256]; smaller layers are centred on a
    pad_value canvas (reference :229-243).  One F.pad per layer instead of one per sample."""
    out = []
    for bchw in list_of_bchw:
        b, c, h, w = bchw.shape
        if h != 256 or w != 256:
            pad_x, pad_y = 256 - w, 256 - h
            px1, py1 = pad_x // 2, pad_y // 2
            bchw = F.pad(bchw, (px1, pad_x - px1, py1, pad_y - py1), mode='constant', value=pad_value)
        out.append(bchw)
    return torch.stack(out, dim=1).contiguous()


def generate_pseudo_fake(blchw):
    """Move the content of every layer to the centre of its canvas: crop_to_content then pad_256,
    for all B*L images at once (reference :281-291 loops over them with nonzero()/slicing).
    Input range [0,1]; canvas must be 256x256 as upstream (pad_256)."""
    b, l, c, h, w = blchw.shape
    assert h == 256 and w == 256
    x = blchw.reshape(b * l, c, h, w)
    x1, y1, x2, y2 = content_boxes(x[:, 3])
    cw, chh = x2 - x1, y2 - y1                                  # crop size (max row/col excluded)
    px1, py1 = (256 - cw) // 2, (256 - chh) // 2                # pad_256's left / top padding
    ys = torch.arange(h, device=x.device).reshape(1, h)
    xs = torch.arange(w, device=x.device).reshape(1, w)
    src_y = ys - py1.reshape(-1, 1) + y1.reshape(-1, 1)         # [M,H] source row of each output row
    src_x = xs - px1.reshape(-1, 1) + x1.reshape(-1, 1)
    ok_y = (ys >= py1.reshape(-1, 1)) & (ys < (py1 + chh).reshape(-1, 1))
    ok_x = (xs >= px1.reshape(-1, 1)) & (xs < (px1 + cw).reshape(-1, 1))
    src_y, src_x = src_y.clamp(0, h - 1), src_x.clamp(0, w - 1)
    rows = torch.gather(x, 2, src_y.reshape(-1, 1, h, 1).expand(-1, c, h, w))
    out = torch.gather(rows, 3, src_x.reshape(-1, 1, 1, w).expand(-1, c, h, w))
    out = out * (ok_y.reshape(-1, 1, h, 1) & ok_x.reshape(-1, 1, 1, w)).to(out.dtype)
    return out.reshape(b, l, c, h, w)


def make_batch_for_local_d(blchw, layer_size_list, to_minus11=False):
    """[B,L,C,H,W] in [0,1] -> list (one per layer) of centre crops [B,C,h_l,w_l] of the centred
    content, the real batches of the local discriminators (reference :246-278)."""
    b, l, c, h, w = blchw.shape
    centered = generate_pseudo_fake(blchw)
    out = []
    for li, (bh, bw) in enumerate(layer_size_list):
        x1, y1 = (w - bw) // 2, (h - bh) // 2
        x2, y2 = (w + bw) // 2, (h + bh) // 2
        crop = centered[:, li, :, y1:y2, x1:x2]
        out.append(normalize_minus11(crop) if to_minus11 else crop)
    return out


def convert_translate_to_2x3(translation_bl2_l2):
    """[..., 2] translations -> [..., 2, 3] affine matrices [[1,0,dx],[0,1,dy]] (reference :316-335)."""
    return _ts.translation_to_theta(translation_bl2_l2)


def random_position(blchw):
    """Move each layer to a random position; input range [0,1] (reference :294-308)."""
    b, l, c, h, w = blchw.shape
    t = torch.empty((b * l, 2), device=blchw.device).uniform_(-1., 1.)
    return _ts.translate_sample(blchw.reshape(-1, c, h, w), t, shift=0.0).reshape(b, l, c, h, w)


def stack_layer_to_channel(tensor_in):
    b, l, c, h, w = tensor_in.shape
    return torch.reshape(tensor_in, (b, l * c, h, w))


def unstack_layer_to_channel(tensor_in, num_channels=4):
    b, lc, h, w = tensor_in.shape
    return torch.reshape(tensor_in, (b, lc // num_channels, num_channels, h, w))


def calc_psnr(x, y, data_range=1):
    mse = torch.nn.functional.mse_loss(x, y)
    return 10 * math.log10(data_range ** 2 / mse.item())
