"""3D U-Net building blocks -- MI355X-native stand-ins for the reference's model/unet.py.

Same class names, constructor arguments and ``state_dict`` keys as the reference (model/unet.py:79-100 SingleConv,
:103-144 DoubleConv, :147-159 StepDownDoubleConv, :210-253 Encoder, :256-322 Decoder/DecoderNoJoining, :392-537
Abstract3DUNet/UNet3D), so reference checkpoints load unchanged.  The modules only HOLD parameters; ``forward`` runs
the gfx950 kernels of librfuse_hip.so through ``rfuse.ops``:

  SingleConv('gcr')  = rf_gn_stats / rf_gn_from_stats  +  rf_conv3d_k3_gn_relu  (GroupNorm apply, upsample+concat read, ReLU fused in)
  Encoder pooling    = rf_maxpool3d_2
  Decoder upsample + concat: never materialised -- the conv reads (skip, low-res) as two sources.

Only layer order 'gcr' with DoubleConv is implemented: it is the only path any shipped config reaches (SURVEY.md 2,
row 1).  Inference only (no autograd); CPU tensors raise.
"""
import math

import torch
from torch import nn

from rfuse import ops


class GroupNormParams(nn.Module):
    """weight/bias holder with nn.GroupNorm's names, init (ones / zeros) and the <groups collapse of model/unet.py:62-63."""

    def __init__(self, num_groups, num_channels, eps=1e-5):
        super().__init__()
        if num_channels < num_groups:
            num_groups = 1
        assert num_channels % num_groups == 0, \
            f'Expected number of channels in input to be divisible by num_groups. num_channels={num_channels}, num_groups={num_groups}'
        self.num_groups, self.num_channels, self.eps = num_groups, num_channels, eps
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))

    def extra_repr(self):
        return f'{self.num_groups}, {self.num_channels}, eps={self.eps}'


class Conv3dParams(nn.Module):
    """weight (and optional bias) holder with nn.Conv3d's names, shapes (OIDHW) and default initialisation."""

    def __init__(self, in_channels, out_channels, kernel_size, bias, stride=1, padding=0):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()
        self._packed = ops.PackedWeight('conv3' if (kernel_size == 3 and padding == 1) else 'convv')
        self._packed_up = ops.PackedWeight('conv3up')
        self._packed_up_split = ops.PackedWeight('conv3ups')
        self._packed_split = ops.PackedWeight('conv3s')
        self._packed_e2 = ops.PackedWeight('conv3e2')
        self._packed_lds = ops.PackedWeight('convvl')
        self._packed_valu = ops.PackedWeight('convvv')
        self._packed_vsplit = ops.PackedWeight('convvs')
        self._packed_vpg = ops.PackedWeight('convvpg')

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * self.kernel_size ** 3
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)

    def packed(self):
        return self._packed.get(self.weight)

    def packed_lds(self):
        """operand image of the LDS-staged valid-conv form (patch encoders' large layers)"""
        return self._packed_lds.get(self.weight)

    def packed_valu(self):
        """[cin, k,k,k, cout] image of the VALU valid-conv form (patch encoders' first layers)"""
        return self._packed_valu.get(self.weight)

    def packed_valid_split(self, s):
        """f16 fragment image of the split-operand valid-conv form for input edge s (csrc/conv_valid_split.hip)"""
        return self._packed_vsplit.get(self.weight, s, self.stride)

    def packed_valid_split_pg(self, s):
        """weight image of the persistent grid form of the split-operand valid conv for input edge s (csrc/conv_valid_split_pg.hip)"""
        return self._packed_vpg.get(self.weight, s, self.stride)

    def packed_up(self, c0):
        """operand image of the decoder form (first c0 input channels = skip source, rest = upsampled source)"""
        return self._packed_up.get(self.weight, c0)

    def packed_split(self):
        """f16 fragment image of the split-operand box conv (csrc/conv3d_split.hip)"""
        return self._packed_split.get(self.weight)

    def packed_e2_split(self, edge):
        """f16 fragment image of the dense GEMM form on whole 2^3 / 1^3 volumes (csrc/conv3d_e2_split.hip)"""
        return self._packed_e2.get(self.weight, edge)

    def packed_up_split(self, c0):
        """f16 fragment image of the split-operand decoder form (csrc/conv3d_up_split.hip)"""
        return self._packed_up_split.get(self.weight, c0)

    def extra_repr(self):
        return f'{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, padding={self.padding}'


def _groups_of(gn):
    """the group count GroupNormParams really normalises with (reference model/unet.py:62-63: one group when there are fewer channels than groups)"""
    return 1 if gn.num_channels < gn.num_groups else gn.num_groups


def _group_elements(gn, edge):
    """elements one GroupNorm group holds on an edge^3 volume (the bound on |GroupNorm output| that ops.split_range_ok checks is gamma * sqrt(that) + |beta|)"""
    return (gn.num_channels // _groups_of(gn)) * edge ** 3


class SingleConv(nn.Module):
    """GroupNorm -> Conv3d(k3, p1, no bias) -> ReLU, order 'gcr' (reference model/unet.py:19-76,79-100)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, order='gcr', num_groups=8, padding=1):
        super().__init__()
        if order != 'gcr' or kernel_size != 3 or padding != 1:
            raise NotImplementedError(
                "only layer order 'gcr' with 3x3x3 kernels and padding 1 is built (the only order the shipped configs use); got "
                f"order={order!r} kernel_size={kernel_size} padding={padding}")
        self.groupnorm = GroupNormParams(num_groups, in_channels)
        self.conv = Conv3dParams(in_channels, out_channels, 3, bias=False, padding=1)

    def forward(self, x, upsampled=None, _direct=False, pool=None):
        """x: full-resolution source [N,C0,S,S,S] or None; ``upsampled``: low-resolution source [N,C1,S/2,S/2,S/2] that the
        reference would nearest-upsample and concatenate after x (model/unet.py:297-308).

        ``pool``: None -> returns the output; 'also' / 'only' -> returns (output, MaxPool3d(2)(output)) with the pooling
        fused into the conv epilogue where the tiling allows ('only': the caller never reads the full-resolution output, which
        is then not written at all and returned as None)."""
        gn = self.groupnorm
        cout = self.conv.out_channels
        if ops.needs_grad(x, upsampled, self.conv.weight, gn.weight, gn.bias):
            # training slice: the same kernels behind torch.autograd.Function, see rfuse/autograd.py
            from rfuse import autograd as rf_autograd
            out = rf_autograd.conv_gn_relu(x, upsampled, gn.weight, gn.bias, self.conv.weight, gn.num_groups, gn.eps)
            return out if pool is None else (out, rf_autograd.max_pool2(out))
        aff = ops.gn_affine(x, upsampled, gn.weight, gn.bias, gn.num_groups, gn.eps)
        edge = x.shape[2] if x is not None else 2 * upsampled.shape[2]
        # the split-operand (F16 matrix core) forms only where they cannot saturate: weights and GroupNorm outputs inside the f16 pair's range
        # (ops.split_range_ok, decided from the parameters once per version); otherwise the fp32 kernels, like the reference's fp32 path
        split_ok = ops.CONV_ARITH == 'split' and ops.split_range_ok(self.conv.weight, gn.weight, gn.bias, _group_elements(gn, edge))
        if not split_ok:
            return self._forward_fp32(x, upsampled, aff, cout, edge, _direct, pool)
        if upsampled is None and not _direct and ops.conv_split_supported(x, None, cout):
            return ops.conv3d_split_gn_relu(x, aff, self.conv.packed_split(), cout, pool=pool)
        if upsampled is None and not _direct and edge <= 2 and ops.conv_e2_split_supported(x, cout):
            out = ops.conv3d_e2_split_gn_relu(x, aff, self.conv.packed_e2_split(edge), cout)
            return out if pool is None else (out, ops.maxpool2(out))
        if upsampled is not None and edge == 2 and not _direct and pool is None:
            # decoder stage on 2^3 volumes (skip @2^3 + a 1^3 source): the eight copies of the low-resolution voxel written out (a [n][c0 + c1][8] tensor, a
            # few KB per sample) and the layer run as the dense GEMM -- the affine table is per channel of the concatenation either way
            up = upsampled.reshape(upsampled.shape[0], upsampled.shape[1], 1, 1, 1).expand(-1, -1, 2, 2, 2)
            xc = torch.cat((x, up), dim=1) if x is not None else up.contiguous()
            if ops.conv_e2_split_supported(xc, cout):
                return ops.conv3d_e2_split_gn_relu(xc, aff, self.conv.packed_e2_split(2), cout)
        if pool is not None and upsampled is None and not _direct and edge >= 4 and ops.conv_pool_supported(x, None, cout):
            return ops.conv3d_gn_relu_pool(x, None, aff, self.conv.packed(), cout, keep_full=(pool == 'also'))
        if _direct or edge == 1:
            out = ops.conv3d_gn_relu(x, upsampled, aff, None, cout, direct_weight=self.conv.weight)
        elif ops.conv_up_split_supported(x, upsampled, cout):
            c0 = x.shape[1] if x is not None else 0
            out = ops.conv3d_up_split_gn_relu(x, upsampled, aff, self.conv.packed_up_split(c0), cout)
        elif upsampled is not None and edge >= 8 and ops.conv_split_supported_shape(upsampled.shape[0], (x.shape[1] if x is not None else 0) + upsampled.shape[1], edge, cout):
            # a decoder stage outside the decoder forms of the F16 cores (C5's 96 + 192 -> 96 @16^3 at 16 chunks: 192 upsampled channels, 128 boxes): the
            # concatenation written out once (as for the 2^3 stage above; 75 MB there) and the layer run by the split-operand box kernel with its cout blocks
            # on grid.y -- 0.53 ms as the fp32-MFMA decoder form.  The affine table is per channel of the concatenation either way.
            up = upsampled.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3).repeat_interleave(2, dim=4)
            xc = torch.cat((x, up), dim=1) if x is not None else up
            out = ops.conv3d_split_gn_relu(xc, aff, self.conv.packed_split(), cout)
        elif ops.conv_up_supported(x, upsampled, cout):
            c0 = x.shape[1] if x is not None else 0
            out = ops.conv3d_up_gn_relu(x, upsampled, aff, self.conv.packed_up(c0), cout)
        else:
            out = ops.conv3d_gn_relu(x, upsampled, aff, self.conv.packed(), cout)
        return out if pool is None else (out, ops.maxpool2(out))

    def _forward_fp32(self, x, upsampled, aff, cout, edge, _direct, pool):
        """the same layer on the fp32-MFMA kernels only (a parameter outside the split forms' range)"""
        if pool is not None and upsampled is None and not _direct and edge >= 4 and ops.conv_pool_supported(x, None, cout):
            return ops.conv3d_gn_relu_pool(x, None, aff, self.conv.packed(), cout, keep_full=(pool == 'also'))
        if _direct or edge == 1:
            out = ops.conv3d_gn_relu(x, upsampled, aff, None, cout, direct_weight=self.conv.weight)
        elif ops.conv_up_supported(x, upsampled, cout):
            out = ops.conv3d_up_gn_relu(x, upsampled, aff, self.conv.packed_up(x.shape[1] if x is not None else 0), cout)
        else:
            out = ops.conv3d_gn_relu(x, upsampled, aff, self.conv.packed(), cout)
        return out if pool is None else (out, ops.maxpool2(out))




def _decoder_pair_presplit_ok(c1, c2, x, upsampled):
    """A decoder's conv pair on whole 8^3 samples: the first conv (decoder form, split operands) sees the whole sample, so it can apply the SECOND
    conv's GroupNorm to its own output and hand it over pre-split: no fp32 intermediate, no rf_gn_from_stats, the second conv stages copies."""
    g1, g2 = c1.groupnorm, c2.groupnorm
    if ops.needs_grad(x, upsampled, c1.conv.weight, c2.conv.weight, g1.weight, g2.weight):
        return False
    cmid, n, edge = c1.conv.out_channels, upsampled.shape[0], 2 * upsampled.shape[2]
    if not ops.conv_up_split_presplit_supported(x, upsampled, cmid, _groups_of(g2)):
        return False
    if not bool(ops._lib.load().rf_conv3d_split_pre_supported(cmid, n, edge, c2.conv.out_channels)):
        return False
    return (ops.split_range_ok(c1.conv.weight, g1.weight, g1.bias, _group_elements(g1, edge))
            and ops.split_range_ok(c2.conv.weight, g2.weight, g2.bias, _group_elements(g2, edge)))


def _decoder_pair_presplit(c1, c2, x, upsampled):
    g1, g2 = c1.groupnorm, c2.groupnorm
    aff = ops.gn_affine(x, upsampled, g1.weight, g1.bias, g1.num_groups, g1.eps)
    c0 = x.shape[1] if x is not None else 0
    # parity-major hand-over where the persistent producer and the persistent consumer both take the shapes (the bench's 8192 patches: k_conv3_up_split_pp)
    pm = ops.conv_up_split_presplit_pm_supported(x, upsampled, c1.conv.out_channels, _groups_of(g2), c2.conv.out_channels)
    pre = ops.conv3d_up_split_presplit(x, upsampled, aff, c1.conv.packed_up_split(c0), c1.conv.out_channels, g2.weight, g2.bias, _groups_of(g2), g2.eps, parity_major=pm)
    return ops.conv3d_split_pre_relu(pre, c1.conv.out_channels, upsampled.shape[0], 2 * upsampled.shape[2], c2.conv.packed_split(), c2.conv.out_channels, parity_major=pm)


class DoubleConv(nn.Module):
    """Two SingleConvs; channel plan of reference model/unet.py:125-144."""

    def __init__(self, in_channels, out_channels, encoder, kernel_size=3, order='gcr', num_groups=8):
        super().__init__()
        if encoder:
            c1_in, c1_out = in_channels, max(out_channels // 2, in_channels)
            c2_in, c2_out = c1_out, out_channels
        else:
            c1_in, c1_out = in_channels, out_channels
            c2_in, c2_out = out_channels, out_channels
        self.SingleConv1 = SingleConv(c1_in, c1_out, kernel_size, order, num_groups)
        self.SingleConv2 = SingleConv(c2_in, c2_out, kernel_size, order, num_groups)

    def forward(self, x, upsampled=None, pool=None, next_block=None):
        """``next_block``: the DoubleConv that will read MaxPool3d(2) of this block's output (the next encoder level), when the caller knows it -- with
        pool='only' the pooled tensor can then be handed over pre-split (returned as an ops.PreSplit in place of the pooled tensor)."""
        c1, c2 = self.SingleConv1, self.SingleConv2
        if isinstance(x, ops.PreSplit):
            # the previous level handed its pooled output over pre-split for THIS block's first GroupNorm: both convs stage copies
            g2 = c2.groupnorm
            mid = ops.conv3d_split_pre_presplit(x, c1.conv.packed_split(), c1.conv.out_channels, g2.weight, g2.bias, _groups_of(g2), g2.eps)
            return ops.conv3d_split_pre_relu(mid.data, mid.channels, mid.n, mid.edge, c2.conv.packed_split(), c2.conv.out_channels, pool=pool)
        if upsampled is None and self._presplit_ok(x):
            # level 0 of a U-Net on 16^3 samples: the first conv hands the second its input already normalised (second GroupNorm) and split
            # into f16 pairs (ops.conv3d_cin1_presplit) -- the second conv stages it with copies
            g1, g2 = c1.groupnorm, c2.groupnorm
            n, edge, cmid, cout = x.shape[0], x.shape[2], c1.conv.out_channels, c2.conv.out_channels
            pre = ops.conv3d_cin1_presplit(x, g1.weight, g1.bias, g1.eps, c1.conv.packed(), cmid, g2.weight, g2.bias, g2.num_groups, g2.eps)
            if pool == 'only' and next_block is not None and next_block.accepts_prepooled(n, cout, edge // 2):
                ng = next_block.SingleConv1.groupnorm
                if ops.conv_split_pre_pool_presplit_supported(cmid, n, edge, cout, _groups_of(ng)):
                    _, handed = ops.conv3d_split_pre_relu_pool_presplit(pre, cmid, n, edge, c2.conv.packed_split(), cout, ng.weight, ng.bias, _groups_of(ng), ng.eps)
                    return None, handed
            return ops.conv3d_split_pre_relu(pre, cmid, n, edge, c2.conv.packed_split(), cout, pool=pool)
        if upsampled is not None and pool is None and _decoder_pair_presplit_ok(c1, c2, x, upsampled):
            return _decoder_pair_presplit(c1, c2, x, upsampled)
        if upsampled is None and x is not None and self._box_pair_presplit_ok(x):
            # an encoder level on whole 8^3 samples (16 -> 16 -> 32 of the retrieval backbone): as above, the producer is the split box kernel
            g1, g2 = c1.groupnorm, c2.groupnorm
            aff = ops.gn_affine(x, None, g1.weight, g1.bias, g1.num_groups, g1.eps)
            pre = ops.conv3d_split_presplit(x, aff, c1.conv.packed_split(), c1.conv.out_channels, g2.weight, g2.bias, _groups_of(g2), g2.eps)
            return ops.conv3d_split_pre_relu(pre, c1.conv.out_channels, x.shape[0], x.shape[2], c2.conv.packed_split(), c2.conv.out_channels, pool=pool)
        return c2(c1(x, upsampled), pool=pool)

    def accepts_prepooled(self, n, cin, edge):
        """this block's conv pair can take its input as an ops.PreSplit of [n, cin, edge^3] (normalised for SingleConv1's GroupNorm) -- decided from shapes and
        parameters only, so that the PRODUCER can ask before it writes"""
        c1, c2 = self.SingleConv1, self.SingleConv2
        g1, g2 = c1.groupnorm, c2.groupnorm
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return False
        if c1.conv.in_channels != cin:
            return False
        cmid = c1.conv.out_channels
        if not ops.conv_split_pre_presplit_supported(cin, n, edge, cmid, _groups_of(g2)) or not bool(ops._lib.load().rf_conv3d_split_pre_supported(cmid, n, edge, c2.conv.out_channels)):
            return False
        return (ops.split_range_ok(c1.conv.weight, g1.weight, g1.bias, _group_elements(g1, edge))
                and ops.split_range_ok(c2.conv.weight, g2.weight, g2.bias, _group_elements(g2, edge)))

    def _box_pair_presplit_ok(self, x):
        c1, c2 = self.SingleConv1, self.SingleConv2
        g1, g2 = c1.groupnorm, c2.groupnorm
        if ops.needs_grad(x, c1.conv.weight, c2.conv.weight, g1.weight, g2.weight):
            return False
        cmid, n, edge = c1.conv.out_channels, x.shape[0], x.shape[2]
        if not ops.conv_split_presplit_supported(x, cmid, _groups_of(g2)) or not bool(ops._lib.load().rf_conv3d_split_pre_supported(cmid, n, edge, c2.conv.out_channels)):
            return False
        return (ops.split_range_ok(c1.conv.weight, g1.weight, g1.bias, _group_elements(g1, edge))
                and ops.split_range_ok(c2.conv.weight, g2.weight, g2.bias, _group_elements(g2, edge)))

    def _presplit_ok(self, x):
        c1, c2 = self.SingleConv1, self.SingleConv2
        g1, g2 = c1.groupnorm, c2.groupnorm
        if x is None or ops.needs_grad(x, c1.conv.weight, c2.conv.weight, g1.weight, g2.weight):
            return False
        if not ops.cin1_presplit_supported(x, c1.conv.out_channels, g2.num_groups, c2.conv.out_channels):
            return False
        edge = x.shape[2]
        return ops.split_range_ok(c2.conv.weight, g2.weight, g2.bias, _group_elements(g2, edge))


class StepDownDoubleConv(nn.Module):
    """in -> (in+out)//2 -> out (reference model/unet.py:149-159)."""

    def __init__(self, in_channels, out_channels, encoder, kernel_size=3, order='gcr', num_groups=8):
        super().__init__()
        self.encoder = encoder
        mid = (in_channels + out_channels) // 2
        self.SingleConv1 = SingleConv(in_channels, mid, kernel_size, order, num_groups)
        self.SingleConv2 = SingleConv(mid, out_channels, kernel_size, order, num_groups)

    def forward(self, x, upsampled=None):
        if upsampled is not None and _decoder_pair_presplit_ok(self.SingleConv1, self.SingleConv2, x, upsampled):
            return _decoder_pair_presplit(self.SingleConv1, self.SingleConv2, x, upsampled)
        return self.SingleConv2(self.SingleConv1(x, upsampled))


class Encoder(nn.Module):
    """optional MaxPool3d(2) then the basic module (reference model/unet.py:230-253)."""

    def __init__(self, in_channels, out_channels, conv_kernel_size=3, apply_pooling=True, pool_kernel_size=(2, 2, 2),
                 pool_type='max', basic_module=DoubleConv, conv_layer_order='gcr', num_groups=8):
        super().__init__()
        if apply_pooling and (pool_type != 'max' or tuple(pool_kernel_size) != (2, 2, 2)):
            raise NotImplementedError('only MaxPool3d(2) is built')
        self.apply_pooling = apply_pooling
        self.basic_module = basic_module(in_channels, out_channels, encoder=True, kernel_size=conv_kernel_size,
                                         order=conv_layer_order, num_groups=num_groups)

    def forward(self, x, prepooled=None, pool=None, next_block=None):
        """``prepooled``: MaxPool3d(2)(x) when the producer already emitted it (fused epilogue; an ops.PreSplit when it was handed over pre-split);
        ``pool``: see SingleConv.forward; ``next_block``: see DoubleConv.forward."""
        if self.apply_pooling:
            if prepooled is not None:
                x = prepooled
            elif ops.needs_grad(x):
                x = torch.nn.functional.max_pool3d(x, 2)            # grad mode: torch's max-pool carries the backward
            else:
                x = ops.maxpool2(x)
        if pool is None:
            return self.basic_module(x)
        if next_block is not None and isinstance(self.basic_module, DoubleConv):
            return self.basic_module(x, pool=pool, next_block=next_block)
        return self.basic_module(x, pool=pool)


class Decoder(nn.Module):
    """nearest upsample to the skip's size + concat (skip first) + basic module (reference model/unet.py:273-308)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, scale_factor=(2, 2, 2), basic_module=DoubleConv,
                 conv_layer_order='gcr', num_groups=8, mode='nearest'):
        super().__init__()
        if basic_module not in (DoubleConv, StepDownDoubleConv) or mode != 'nearest' or tuple(scale_factor) != (2, 2, 2):
            raise NotImplementedError('only nearest x2 upsampling with concat joining is built')
        self.basic_module = basic_module(in_channels, out_channels, encoder=False, kernel_size=kernel_size,
                                         order=conv_layer_order, num_groups=num_groups)

    def forward(self, encoder_features, x):
        if encoder_features.shape[2] != 2 * x.shape[2]:
            raise NotImplementedError('skip connection must be exactly twice the decoder input resolution')
        return self.basic_module(encoder_features, x)


class DecoderNoJoining(Decoder):
    """x2 nearest upsample then the basic module, no skip (reference model/unet.py:311-322)."""

    # noinspection PyMethodOverriding
    def forward(self, x):
        return self.basic_module(None, x)


def number_of_features_per_level(init_channel_number, num_levels):
    return [init_channel_number * 2 ** k for k in range(num_levels)]


class UNet3D(nn.Module):
    """Encoder/decoder wiring of reference Abstract3DUNet (model/unet.py:424-520) with DoubleConv, final_conv=False."""

    def __init__(self, in_channels, out_channels, final_sigmoid=True, f_maps=64, layer_order='gcr', num_groups=8, num_levels=4,
                 is_segmentation=True, remove_n_final_layers=0, final_conv=False, **kwargs):
        super().__init__()
        if final_conv or is_segmentation:
            raise NotImplementedError('final_conv / segmentation heads are never instantiated by the refinement path')
        if isinstance(f_maps, int):
            f_maps = number_of_features_per_level(f_maps, num_levels=num_levels)
        encoders = []
        for i, out_feature_num in enumerate(f_maps):
            encoders.append(Encoder(in_channels if i == 0 else f_maps[i - 1], out_feature_num, apply_pooling=(i > 0),
                                    basic_module=DoubleConv, conv_layer_order=layer_order, num_groups=num_groups))
        self.encoders = nn.ModuleList(encoders)

        reversed_f_maps = list(reversed(f_maps))
        if remove_n_final_layers > 0:
            reversed_f_maps = reversed_f_maps[:-remove_n_final_layers]
        outs = list(reversed_f_maps)
        outs[-1] = out_channels                                      # model/unet.py:453-455
        decoders = []
        for i in range(len(reversed_f_maps) - 1):
            in_feature_num = reversed_f_maps[i] + reversed_f_maps[i + 1]
            last_and_trimmed = i == (len(reversed_f_maps) - 2) and remove_n_final_layers > 0      # model/unet.py:465
            decoders.append(Decoder(in_feature_num, outs[i + 1], basic_module=StepDownDoubleConv if last_and_trimmed else DoubleConv,
                                    conv_layer_order=layer_order, num_groups=num_groups))
        self.decoders = nn.ModuleList(decoders)

    def forward(self, x, after_encoders=None):
        """``after_encoders``: optional callable run between the encoder and the decoder launches (the engine issues the next batch's front end there)"""
        feats = []
        n_enc, n_dec = len(self.encoders), len(self.decoders)
        pooled = None
        for i, encoder in enumerate(self.encoders):
            # Who reads this level's full-resolution output?  The next level reads MaxPool3d(2) of it; a decoder reads it as
            # a skip only for levels n_enc-1-n_dec .. n_enc-2 (model/unet.py:500-507: the list is cut and zip() truncates, so
            # with remove_n_final_layers the finest levels are never joined).  Unread outputs are not even written.
            if i == n_enc - 1:
                x, pooled = encoder(x, prepooled=pooled), None
            else:
                is_skip = i >= n_enc - 1 - n_dec
                nxt = self.encoders[i + 1].basic_module
                x, pooled = encoder(x, prepooled=pooled, pool='also' if is_skip else 'only', next_block=nxt if isinstance(nxt, DoubleConv) else None)
            feats.insert(0, x)
        feats = feats[1:]                                            # model/unet.py:500-504
        if after_encoders is not None:
            after_encoders()
        for decoder, skip in zip(self.decoders, feats):              # zip truncates, model/unet.py:507
            x = decoder(skip, x)
        return x
