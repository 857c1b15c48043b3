"""Refinement networks -- MI355X-native stand-ins for the reference's model/refinement.py (same names, constructor
arguments, ``state_dict`` keys; reference lines in each docstring).  Inference only; forwards run HIP kernels."""
import math

import torch
from torch import nn

from model.unet import UNet3D, DecoderNoJoining
from rfuse import ops


class Superresolution08UNetBackbone(nn.Module):
    """1x8^3 -> nf x 32^3 (reference model/refinement.py:6-19): UNet3D + two DecoderNoJoining."""

    def __init__(self, nf, num_levels, layer_order):
        super().__init__()
        self.network = nn.ModuleList([
            UNet3D(in_channels=1, out_channels=2 * nf, final_sigmoid=False, final_conv=False, f_maps=nf, num_groups=nf // 2,
                   layer_order=layer_order, num_levels=num_levels, is_segmentation=False),
            DecoderNoJoining(2 * nf, 2 * nf, conv_layer_order=layer_order, num_groups=nf // 2),
            DecoderNoJoining(2 * nf, nf, conv_layer_order=layer_order, num_groups=nf // 2),
        ])

    def forward(self, x):
        for net in self.network:
            x = net(x)
        return x


class Superresolution16UNetBackbone(nn.Module):
    """1x16^3 -> nf x 32^3 (reference model/refinement.py:22-34)."""

    def __init__(self, nf, num_levels, layer_order):
        super().__init__()
        self.network = nn.ModuleList([
            UNet3D(in_channels=1, out_channels=2 * nf, final_sigmoid=False, final_conv=False, f_maps=nf, num_groups=nf // 2,
                   layer_order=layer_order, num_levels=num_levels, is_segmentation=False),
            DecoderNoJoining(2 * nf, nf, conv_layer_order=layer_order, num_groups=nf // 2),
        ])

    def forward(self, x):
        for net in self.network:
            x = net(x)
        return x


class SurfaceReconstructionUNetBackbone(nn.Module):
    """1x128^3 occupancy grid -> nf x 32^3 (reference model/refinement.py:37-45)."""

    def __init__(self, nf, num_levels, layer_order):
        super().__init__()
        self.network = UNet3D(in_channels=1, out_channels=nf, final_sigmoid=False, final_conv=False, remove_n_final_layers=2, f_maps=nf,
                              layer_order=layer_order, num_groups=nf // 2, num_levels=num_levels, is_segmentation=False)

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
        self.network = nn.ModuleList([
            DecoderNoJoining(nf, nf, conv_layer_order=layer_order, num_groups=nf // 2),
            PointwiseConvParams(nf, 1),
            TanhMarker(),
        ])

    def forward(self, x):
        x = self.network[0](x)
        if ops.needs_grad(x, self.network[1].weight):
            # grad mode (training slice, rfuse/autograd.py): the 16 -> 1 pointwise conv + tanh is 0.1 % of the work, torch differentiates it
            return torch.tanh(torch.nn.functional.conv3d(x, self.network[1].weight, self.network[1].bias))
        return ops.conv1x1_tanh(x, self.network[1].weight, self.network[1].bias)

    def forward_df(self, x, target_trunc):
        if ops.needs_grad(x, self.network[1].weight):
            return (self.forward(x) + 1.0) * (float(target_trunc) / 2)
        x = self.network[0](x)
        return ops.conv1x1_tanh(x, self.network[1].weight, self.network[1].bias, post_add=1.0, post_mul=float(target_trunc) / 2)


class RetrievalUNetBackbone(nn.Module):
    """1x16^3 patches -> nf x 8^3 features (reference model/refinement.py:64-73)."""

    def __init__(self, f_maps, nf, num_levels, layer_order):
        super().__init__()
        self.nf = nf
        self.network = UNet3D(in_channels=1, out_channels=nf, num_groups=nf // 2, final_sigmoid=False, final_conv=False,
                              remove_n_final_layers=1, f_maps=f_maps, layer_order=layer_order, num_levels=num_levels, is_segmentation=False)

    def forward(self, x):
        return self.network(x)
