"""Refinement networks -- MI355X-native stand-ins for the reference's model/refinement.py (same names, constructor
arguments, ``state_dict`` keys; reference lines in each docstring).  Inference only; forwards run HIP kernels."""
import math

import torch
from torch import nn

from model.unet import UNet3D, DecoderNoJoining, _group_elements
from rfuse import ops


def _unet(width, out_channels, order, levels, f_maps=None, trim=0):
    """The one U-Net flavour every refinement network uses: 1 input channel, GroupNorm groups = width / 2, no final conv / activation head;
    ``trim`` = remove_n_final_layers (decoder stages dropped at the fine end)."""
    return UNet3D(1, out_channels, f_maps=width if f_maps is None else f_maps, num_groups=width // 2, num_levels=levels, layer_order=order,
                  remove_n_final_layers=trim, final_sigmoid=False, final_conv=False, is_segmentation=False)


def _up_stage(width, cin, cout, order):
    """x2 nearest upsample + DoubleConv without a skip connection"""
    return DecoderNoJoining(cin, cout, conv_layer_order=order, num_groups=width // 2)


class _Chain(nn.Module):
    """``self.network`` = ModuleList applied in order (the list indices are part of the state_dict key contract: network.0, network.1, ...)"""

    def __init__(self, stages):
        super().__init__()
        self.network = nn.ModuleList(stages)

    def forward(self, x):
        for stage in self.network:
            x = stage(x)
        return x


class Superresolution08UNetBackbone(_Chain):
    """1 x 8^3 -> nf x 32^3 (reference model/refinement.py:6-19): U-Net to 2 nf channels at 8^3, then two up stages (2 nf @16^3, nf @32^3)."""

    def __init__(self, nf, num_levels, layer_order):
        super().__init__([_unet(nf, 2 * nf, layer_order, num_levels), _up_stage(nf, 2 * nf, 2 * nf, layer_order), _up_stage(nf, 2 * nf, nf, layer_order)])


class Superresolution16UNetBackbone(_Chain):
    """1 x 16^3 -> nf x 32^3 (reference model/refinement.py:22-34): the same with a single up stage."""

    def __init__(self, nf, num_levels, layer_order):
        super().__init__([_unet(nf, 2 * nf, layer_order, num_levels), _up_stage(nf, 2 * nf, nf, layer_order)])


class SurfaceReconstructionUNetBackbone(nn.Module):
    """1 x 128^3 occupancy grid -> nf x 32^3 (reference model/refinement.py:37-45): U-Net whose two finest decoder stages are dropped."""

    def __init__(self, nf, num_levels, layer_order):
        super().__init__()
        self.network = _unet(nf, nf, layer_order, num_levels, trim=2)

    def forward(self, x):
        return self.network(x)


class PointwiseConvParams(nn.Module):
    """Conv3d(nf, 1, 1) parameter holder: keys ``weight`` [1,nf,1,1,1] / ``bias`` [1], nn.Conv3d default init."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, 1, 1, 1))
        self.bias = nn.Parameter(torch.empty(out_channels))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_channels)
        nn.init.uniform_(self.bias, -bound, bound)


class TanhMarker(nn.Module):
    """Place-holder keeping ``network.2`` in the module list like the reference's nn.Tanh (no parameters)."""


class Superresolution08FinalDecoder(nn.Module):
    """nf x 32^3 -> 1 x 64^3 in (-1,1) (reference model/refinement.py:48-61): DecoderNoJoining, 1x1x1 conv, tanh.

    ``forward_df(x, trunc)`` additionally applies network_pred_to_df (trainer/train_refinement.py:242-243) inside the
    same kernel epilogue."""

    def __init__(self, nf, layer_order):
        super().__init__()
        self.network = nn.ModuleList([_up_stage(nf, nf, nf, layer_order), PointwiseConvParams(nf, 1), TanhMarker()])

    def forward(self, x):
        if not ops.needs_grad(x, *self.parameters()):
            return self._head(x, 0.0, 1.0)
        # grad mode (training slice, rfuse/autograd.py): the 16 -> 1 pointwise conv + tanh is 0.1 % of the work and torch differentiates it --
        # written as a weighted channel sum, not as F.conv3d: that was the one MIOpen call of the path, and its first use runs MIOpen's solver
        # search (naive_conv_ab_nonpacked_wrw, a batched-GEMM bwd_weight, ...: 2.4 s of kernels before the first step finishes)
        x = self.network[0](x)
        w, b = self.network[1].weight, self.network[1].bias
        return torch.tanh((x.unsqueeze(1) * w.view(1, w.shape[0], w.shape[1], 1, 1, 1)).sum(dim=2) + b.view(1, -1, 1, 1, 1))

    def _head(self, x, post_add, post_mul):
        """the up stage's second conv + the pointwise head (+ network_pred_to_df).  Where the split box kernel takes the conv, the head runs in its epilogue:
        the nf-channel 64^3 tensor (0.5 GB per 32 chunks) is neither written nor read back."""
        dc = self.network[0].basic_module
        c1, c2, pw = dc.SingleConv1, dc.SingleConv2, self.network[1]
        g1, g2 = c1.groupnorm, c2.groupnorm
        cmid, cout2, edge2 = c1.conv.out_channels, c2.conv.out_channels, 2 * x.shape[2]
        if (ops.conv_up_split_ch8_supported(x, cmid, cout2)
                and ops.split_range_ok(c1.conv.weight, g1.weight, g1.bias, _group_elements(g1, edge2))
                and ops.split_range_ok(c2.conv.weight, g2.weight, g2.bias, _group_elements(g2, edge2))):
            # both convs on 8^3 boxes of the 64^3 volume: the first writes its output channel-interleaved (8 channels of a voxel together), which is how the
            # second stages it -- two 16-byte loads per voxel instead of eight 4-byte gathers; the second GroupNorm comes from the first's per-box sums
            aff1 = ops.gn_affine(None, x, g1.weight, g1.bias, g1.num_groups, g1.eps)
            y1c, stats, tiles = ops.conv3d_up_split_gn_relu_ch8(x, aff1, c1.conv.packed_up_split(0), cmid)
            aff2 = ops.gn_affine_from_stats(stats, tiles, x.shape[0], cmid, edge2, g2.weight, g2.bias, g2.num_groups, g2.eps)
            return ops.conv3d_split_pointwise_tanh_ch8(y1c, aff2, c2.conv.packed_split(), cout2, pw.weight, pw.bias, post_add, post_mul)
        y1 = c1(None, x)
        cout, edge = c2.conv.out_channels, y1.shape[2]
        if (ops.conv_split_pointwise_supported(y1, cout)
                and ops.split_range_ok(c2.conv.weight, g2.weight, g2.bias, _group_elements(g2, edge))):
            aff = ops.gn_affine(y1, None, g2.weight, g2.bias, g2.num_groups, g2.eps)
            return ops.conv3d_split_pointwise_tanh(y1, aff, c2.conv.packed_split(), cout, pw.weight, pw.bias, post_add, post_mul)
        return ops.conv1x1_tanh(c2(y1), pw.weight, pw.bias, post_add=post_add, post_mul=post_mul)

    def forward_df(self, x, target_trunc):
        if ops.needs_grad(x, *self.parameters()):
            return (self.forward(x) + 1.0) * (float(target_trunc) / 2)
        return self._head(x, 1.0, float(target_trunc) / 2)


class RetrievalUNetBackbone(nn.Module):
    """1x16^3 patches -> nf x 8^3 features (reference model/refinement.py:64-73)."""

    def __init__(self, f_maps, nf, num_levels, layer_order):
        super().__init__()
        self.nf = nf
        self.network = _unet(nf, nf, layer_order, num_levels, f_maps=f_maps, trim=1)

    def forward(self, x, after_encoders=None):
        return self.network(x, after_encoders=after_encoders) if after_encoders is not None else self.network(x)
