"""Patch embedding networks -- MI355X-native stand-ins for the reference's model/retrieval.py.

All 14 class names of the reference exist with the same constructor ``(nf, z_dim)``, the same ``state_dict`` keys
(``layers.<even index>.{weight,bias}``, ``final_layer.{weight,bias}``) and the same output shape ``[B, z_dim, 1,1,1]``,
so ``model.get_retrieval_networks`` (reference model/__init__.py:6-38) keeps working.  Architectures are kept as data
(one table row per class, reference line numbers alongside) and executed by two generic forwards:

  MLP family   (Patch04 :64-84, Patch05 :87-107, Patch04V2 :110-132): rf_linear (fp32 MFMA) with fused ReLU
  conv family  (Patch08 :136-156, Patch12 :364-388, Patch16 :277-303, Patch24 :306-332, Patch24V2 :335-361,
                Patch32 :4-28, PCPatch32 :187-213, PCPatch48 :217-243, PCPatch64 :247-273):
                rf_conv3d_valid_leaky_valu (first layers: packed-fp32 VALU) / _lds (large layers: LDS-staged MFMA) / _mfma (small ones:
                gather-form MFMA) -- valid
                strided conv + bias + LeakyReLU 0.2 on the fp32 matrix cores -- then rf_linear for final_layer

The two BatchNorm variants (PatchNorm08 :160-184, PatchNorm32 :31-61) construct and serialise identically but their
forward is not built: no shipped config selects them.
"""
import torch
from torch import nn

from model.attention import LinearParams, ActivationMarker
from model.unet import Conv3dParams
from rfuse import ops


class _MLPPatchEncoder(nn.Module):
    WIDTHS = ()          # in units of nf, input first (absolute voxel count)

    def __init__(self, nf, z_dim):
        super().__init__()
        dims = [self.WIDTHS[0]] + [nf * m for m in self.WIDTHS[1:]] + [z_dim]
        layers = []
        for i in range(len(dims) - 1):
            layers.append(LinearParams(dims[i], dims[i + 1]))
            if i < len(dims) - 2:
                layers.append(ActivationMarker('relu'))
        self.layers = nn.ModuleList(layers)

    def forward(self, x):
        ops._no_grad_only(x, self.layers[0].weight)
        x = x.contiguous().reshape([x.shape[0], -1])
        last = len(self.layers) - 1
        for i in range(0, len(self.layers), 2):
            x = self.layers[i].apply_to(x, ops.ACT_NONE if i == last else ops.ACT_RELU)
        return x.reshape([x.shape[0], x.shape[1], 1, 1, 1])


class Patch04(_MLPPatchEncoder):
    WIDTHS = (4 ** 3, 4, 8, 16, 8)


class Patch05(_MLPPatchEncoder):
    WIDTHS = (5 ** 3, 4, 8, 16, 8)


class Patch04V2(_MLPPatchEncoder):
    WIDTHS = (4 ** 3, 4, 8, 16, 16, 8)


class BatchNormParams(nn.Module):
    """nn.BatchNorm3d state holder (weight, bias, running_mean, running_var, num_batches_tracked)."""

    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer('running_mean', torch.zeros(c))
        self.register_buffer('running_var', torch.ones(c))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))


class _ConvPatchEncoder(nn.Module):
    # rows: (cin multiple of nf [0 = single input channel], cout multiple, kernel, stride)
    SPEC = ()
    BATCHNORM = False

    def __init__(self, nf, z_dim):
        super().__init__()
        layers = []
        for cin_m, cout_m, k, stride in self.SPEC:
            layers.append(Conv3dParams(1 if cin_m == 0 else cin_m * nf, cout_m * nf, k, bias=True, stride=stride, padding=0))
            if self.BATCHNORM:
                layers.append(BatchNormParams(cout_m * nf))
            layers.append(ActivationMarker('leaky_relu', 0.2))
        self.layers = nn.ModuleList(layers)
        self.final_layer = LinearParams(self.SPEC[-1][1] * nf, z_dim)

    @staticmethod
    def _split_takes(layer, shape):
        """does `layer` run as the split-operand F16-MFMA form on an input of shape (n, cin, edge)?"""
        return (ops.CONV_ARITH == 'split' and layer.in_channels % 4 == 0 and ops.split_range_ok(layer.weight)
                and bool(ops._lib.load().rf_conv3d_valid_split_supported(max(shape[0], 1), shape[1], shape[2], layer.out_channels, layer.kernel_size, layer.stride)))

    @staticmethod
    def _valu_takes(layer, shape):
        return ops.USE_CONVV_VALU and bool(ops._lib.load().rf_conv3d_valid_valu_supported(max(shape[0], 1), shape[1], shape[2], layer.out_channels, layer.kernel_size,
                                                                                             layer.stride))

    def _conv(self, layer, x, out_split=False):
        """one conv + bias + LeakyReLU; `x` fp32 or ops.SplitActs; out_split: leave the output in split form (the caller knows the next layer reads it)"""
        shape = (x.shape[0], x.shape[1], x.shape[2])
        if isinstance(x, ops.SplitActs) and out_split and ops.conv_valid_split_pg_supported(shape, layer.out_channels, layer.kernel_size, layer.stride) \
                and ops.split_range_ok(layer.weight):
            return ops.conv3d_valid_leaky_split_pg(x, layer.packed_valid_split_pg(x.shape[2]), layer.bias, layer.out_channels, layer.kernel_size, layer.stride, 0.2)
        if isinstance(x, ops.SplitActs) or self._split_takes(layer, shape):
            return ops.conv3d_valid_leaky_split(x, layer.packed_valid_split(x.shape[2]), layer.bias, layer.out_channels, layer.kernel_size, layer.stride, 0.2,
                                                out_split=out_split)
        if self._valu_takes(layer, shape):
            return ops.conv3d_valid_leaky_valu(x, layer.packed_valu(), layer.bias, layer.stride, 0.2, out_split=out_split)
        assert not out_split
        if ops.conv_valid_lds_supported(x, layer.out_channels, layer.kernel_size, layer.stride):
            return ops.conv3d_valid_leaky_lds(x, layer.packed_lds(), layer.bias, layer.out_channels, layer.kernel_size, layer.stride, 0.2)
        return ops.conv3d_valid_leaky_mfma(x, layer.packed(), layer.bias, layer.out_channels, layer.kernel_size, layer.stride, 0.2)

    def _run(self, convs, x, cut=None):
        """The conv layers in order on x.  cut = (index, window edge, lattice step, windows per axis): behind layer `index` the windows are cut out of the
        feature grid (forward_grid).  Activations stay in SPLIT FORM between two layers when the producer can write it (the VALU and split forms;
        couts in fours) and the consumer is a split-operand layer: no GroupNorm sits between the encoders' convs, so what the consumer would make
        of every value it stages (scale, clamp, split -- with its halo, three times per value) the producer makes once."""
        for i, layer in enumerate(convs):
            shape = (x.shape[0], x.shape[1], x.shape[2])
            so = (shape[2] - layer.kernel_size) // layer.stride + 1
            cut_here = cut is not None and cut[0] == i
            next_shape = (shape[0] * cut[3] ** 3, layer.out_channels, cut[1]) if cut_here else (shape[0], layer.out_channels, so)
            writes_split = (ops.USE_SPLIT_CHAIN and i + 1 < len(convs) and layer.out_channels % 4 == 0
                            and (isinstance(x, ops.SplitActs) or self._split_takes(layer, shape) or self._valu_takes(layer, shape))
                            and self._split_takes(convs[i + 1], next_shape))
            x = self._conv(layer, x, out_split=writes_split)
            if cut_here:
                x = ops.gather_windows(x, cut[1], cut[2], cut[3])
        return x

    def _head(self, x):
        if tuple(x.shape[2:]) != (1, 1, 1):
            raise ValueError(f'{type(self).__name__}: input window does not reduce to 1^3 (got {tuple(x.shape[2:])})')
        x = self.final_layer.apply_to(x.reshape(x.shape[0], x.shape[1]))
        return x.reshape([x.shape[0], x.shape[1], 1, 1, 1])

    def forward(self, x):
        if self.BATCHNORM:
            raise NotImplementedError(f'{type(self).__name__}: BatchNorm patch encoders are not built (no shipped config selects them)')
        ops._no_grad_only(x, self.final_layer.weight)
        return self._head(self._run([layer for layer in self.layers if isinstance(layer, Conv3dParams)], x.contiguous()))

    def grid_plan(self, window, step, npatch):
        """How many leading conv layers to evaluate on the whole grid of npatch^3 windows (edge `window`, stride `step`) instead of per window:
        a layer stays on the grid while the window origins stay on its sampling lattice (step divisible by the accumulated stride) and the
        grid has no more output voxels than the windows together (a tie goes to the grid since round 6: one big volume runs on the tiled
        split-operand kernels, 4^3 windows fall to the fp32 gather form -- PCPatch48's 48 -> 96 stride-2 layer: 0.36 ms per 16 chunks).  -> (layers on the grid, window edge / lattice step after them)"""
        sw, sg, lat, on_grid = window, (npatch - 1) * step + window, step, 0
        for layer in self.layers:
            if not isinstance(layer, Conv3dParams):
                continue
            k, st = layer.kernel_size, layer.stride
            if lat % st or sw < k:
                break
            sw2, sg2 = (sw - k) // st + 1, (sg - k) // st + 1
            if sg2 ** 3 > npatch ** 3 * sw2 ** 3 or (npatch - 1) * (lat // st) + sw2 > sg2:
                break
            sw, sg, lat, on_grid = sw2, sg2, lat // st, on_grid + 1
        return on_grid, sw, lat

    def forward_grid(self, grid, window, step):
        """grid [B,1,G,G,G] (a padded, normalised chunk) -> the embeddings of its ((G - window) / step + 1)^3 windows, [B * np^3, z, 1,1,1] in
        rf_query_windows order.  Valid convolutions are translation equivariant: the leading layers run ONCE on the grid (the windows overlap:
        PCPatch48 on 48^3 windows at stride 32 computes every first-layer output 2x, Patch32 at stride 16 3.2x), the windows are cut out of
        the feature grid where that stops paying, the remaining layers run per window.  Same kernels, same per-output arithmetic."""
        if self.BATCHNORM:
            raise NotImplementedError(f'{type(self).__name__}: BatchNorm patch encoders are not built (no shipped config selects them)')
        ops._no_grad_only(grid, self.final_layer.weight)
        g = grid.shape[2]
        if (g - window) % step:
            raise ValueError(f'{type(self).__name__}.forward_grid: grid edge {g} is not window {window} + a multiple of stride {step}')
        npatch = (g - window) // step + 1
        on_grid, _, _ = self.grid_plan(window, step, npatch)
        convs = [layer for layer in self.layers if isinstance(layer, Conv3dParams)]
        # how many of the planned layers the kernels that tile a big volume efficiently really take (the others assume a window-sized input)
        sw, lat, done = window, step, 0
        shape = (grid.shape[0], 1, g)
        for layer in convs[:on_grid]:
            if not (self._split_takes(layer, shape) or self._valu_takes(layer, shape)):
                break
            sw, lat, done = (sw - layer.kernel_size) // layer.stride + 1, lat // layer.stride, done + 1
            shape = (shape[0], layer.out_channels, (shape[2] - layer.kernel_size) // layer.stride + 1)
        x = grid.contiguous()
        if done == 0:
            return self._head(self._run(convs, ops.gather_windows(x, window, step, npatch)))
        return self._head(self._run(convs, x, cut=(done - 1, sw, lat, npatch)))


class Patch32(_ConvPatchEncoder):
    SPEC = ((0, 1, 5, 1), (1, 2, 3, 1), (2, 4, 3, 2), (4, 8, 3, 1), (8, 8, 3, 2), (8, 8, 4, 1))


class PatchNorm32(_ConvPatchEncoder):
    SPEC = Patch32.SPEC
    BATCHNORM = True


class Patch08(_ConvPatchEncoder):
    SPEC = ((0, 1, 3, 1), (1, 4, 3, 1), (4, 4, 3, 1), (4, 8, 2, 1))


class PatchNorm08(_ConvPatchEncoder):
    SPEC = Patch08.SPEC
    BATCHNORM = True


class PCPatch32(_ConvPatchEncoder):
    SPEC = ((0, 1, 3, 1), (1, 2, 3, 1), (2, 4, 3, 2), (4, 4, 3, 1), (4, 8, 3, 2), (8, 8, 3, 1), (8, 8, 3, 1))


class PCPatch48(_ConvPatchEncoder):
    SPEC = ((0, 1, 5, 1), (1, 2, 3, 1), (2, 4, 3, 2), (4, 4, 3, 2), (4, 8, 3, 2), (8, 8, 3, 1), (8, 8, 2, 1))


class PCPatch64(_ConvPatchEncoder):
    SPEC = ((0, 1, 5, 1), (1, 2, 3, 1), (2, 4, 3, 2), (4, 4, 3, 2), (4, 8, 3, 2), (8, 8, 3, 1), (8, 8, 4, 1))


class Patch16(_ConvPatchEncoder):
    SPEC = ((0, 1, 3, 1), (1, 2, 3, 1), (2, 2, 3, 1), (2, 4, 3, 1), (4, 4, 3, 1), (4, 8, 3, 1), (8, 8, 4, 1))


class Patch24(_ConvPatchEncoder):
    SPEC = ((0, 1, 5, 1), (1, 2, 3, 1), (2, 2, 3, 2), (2, 4, 3, 1), (4, 8, 3, 1), (8, 8, 3, 1), (8, 8, 2, 1))


class Patch24V2(_ConvPatchEncoder):
    SPEC = ((0, 1, 3, 1), (1, 2, 3, 1), (2, 2, 3, 2), (2, 4, 3, 1), (4, 8, 3, 1), (8, 8, 3, 1), (8, 8, 3, 1))


class Patch12(_ConvPatchEncoder):
    SPEC = ((0, 1, 3, 1), (1, 2, 3, 1), (2, 4, 3, 1), (4, 4, 3, 1), (4, 8, 3, 1), (8, 8, 2, 1))
