"""Training slice: ``torch.autograd.Function``s whose forward AND backward run the HIP kernels.

The reference trains THROUGH the drop-in modules (trainer/train_refinement.py:41-43 optimisers over unet_backbone / decoder /
retrieval_backbone / attention parameters, :295-306 phase hand-over).  Built here:

  ConvGnRelu   y = ReLU(conv3(GroupNorm(x)))   one SingleConv 'gcr' layer (reference model/unet.py:19-76) -- 97 % of the FLOPs
      forward   rf_gn_stats / rf_gn_from_stats + the inference kernels, chosen like SingleConv.forward chooses them (split-operand F16 forms where
                the parameters are inside their range: one batched range check per optimiser step)
      backward  rf_relu_backward[_amax] -> d xn = conv3(dz, W^T with flipped taps), no ReLU: rf_conv3d_split_k3_gn on dz scaled by a power of two
                into the f16 pairs' range (rf_dgrad_scale_affine; the scale leaves again in the GroupNorm backward), else rf_conv3d_k3_gn (fp32 MFMA)
                rf_conv3d_k3_wgrad_split (F16 matrix cores, split operands; edge >= 8 boxes and whole 4^3 samples) / rf_conv3d_k3_wgrad (fp32 MFMA)
                rf_gn_backward (three passes, every tensor read once per pass)                  = dx, dgamma, dbeta
                2^3 / 1^3 volumes: weight gradient as a split-K MFMA GEMM (rf_linear_wgrad) on the unfolded input, float64 slice sum
  Linear       y = act(x W^T + b)               the layers of AttentionFeatureEncoder (reference model/attention.py:29-46)
      backward  dx = rf_linear(dpre, W^T-as-weight),  dW = rf_linear_wgrad(dpre, x) in row chunks summed in float64
  MaxPool3d(2) / nearest x2 upsample             rf_maxpool3d_2 + rf_maxpool3d_2_backward, rf_upsample3d_2 + rf_sumpool3d_2 (bit-equal to torch's)

torch does the bookkeeping and the light per-row work: transposes / flips of weights, the activation mask of Linear, sums over the batch of
per-sample float64 pieces, the 16 -> 1 pointwise conv + tanh of the final decoder, fold / unfold as views, and the patch attention's per-row
normalise / scores / softmax or straight-through Gumbel-hard / blend (model/attention.py:_forward_autograd) -- together < 1 % of the FLOPs.  With
that the whole training graph of the reference (trainer/train_refinement.py:108-116 forward_full, all four networks trainable = phase 3) runs
through the drop-in modules in grad mode; loss and every parameter gradient are checked against float64 autograd of the oracle in
tests/test_autograd_gpu.py.  Not built: backward of the pre-split pair routes and the fused attention MLP (grad mode takes the plain routes), the
patch encoders (trained by trainer/train_retrieval.py, outside the refinement path).
"""
import torch
import torch.nn.functional as F

from . import _lib, ops

_p, _stream = ops._p, ops._stream         # (the helpers below launch on the device of their tensors: ops._device_scoped)


def _ws(dev, nbytes):
    return ops._workspace(dev, nbytes)


@ops._device_scoped
def relu_backward(dy, y):
    out = torch.empty_like(y)
    _lib.check(_lib.load().rf_relu_backward(_p(dy), _p(y), y.numel(), _p(out), _stream()), 'rf_relu_backward')
    return out


@ops._device_scoped
def relu_backward_amax(dy, y):
    """relu_backward and max |result| (per-workgroup maxima on the device) beside it"""
    out = torch.empty_like(y)
    lib = _lib.load()
    amax = torch.empty(lib.rf_relu_backward_amax_slots(), dtype=torch.float32, device=y.device)       # per-workgroup maxima, reduced by the consumer
    _lib.check(lib.rf_relu_backward_amax(_p(dy), _p(y), y.numel(), _p(out), _p(amax), _stream()), 'rf_relu_backward_amax')
    return out, amax


@ops._device_scoped
def dz_scale(amax, n, cout):
    """from the maxima of relu_backward_amax: s = the power of two that puts max |dz| into [512, 1024) (the split forms carry values as f16 pairs: a
    gradient has to be brought into their range first; a power of two is exact), decided and applied on the device.  Returns the identity GroupNorm
    affine with scale s for [n][cout] and the device pair (s, 1 / s)."""
    ident = torch.empty((n, cout, 4), dtype=torch.float32, device=amax.device)
    scales = torch.empty(2, dtype=torch.float32, device=amax.device)
    _lib.check(_lib.load().rf_dgrad_scale_affine(_p(amax), n * cout, _p(ident), _p(scales), _stream()), 'rf_dgrad_scale_affine')
    return ident, scales


@ops._device_scoped
def dgrad_split(dz, ident, weight, cin):
    """d xn * s = conv3(dz * s, W^T with flipped taps) on the F16 matrix cores (csrc/conv3d_split.hip, split operands)"""
    n, cout, edge = dz.shape[0], dz.shape[1], dz.shape[2]
    wt = weight.flip(2, 3, 4).transpose(0, 1).contiguous()          # [cin, cout, 3,3,3]: the data-gradient conv's weight
    out = torch.empty((n, cin, edge, edge, edge), dtype=torch.float32, device=dz.device)
    _lib.check(_lib.load().rf_conv3d_split_k3_gn(_p(dz), cout, n, edge, _p(ident), _p(ops.pack_conv3_split_weight(wt)), cin, 0, _p(out), _stream()),
               'rf_conv3d_split_k3_gn')
    return out


@ops._device_scoped
def conv3d_wgrad_split(x, aff, dz, scales, cout):
    """dW on the F16 matrix cores (csrc/conv3d_wgrad_split.hip): dz scaled by scales[0] inside the kernel, the result rescaled in its reduction"""
    n, cin, edge = x.shape[0], x.shape[1], x.shape[2]
    lib = _lib.load()
    dw = torch.empty((cout, cin, 3, 3, 3), dtype=torch.float32, device=x.device)
    ws = _ws(x.device, lib.rf_conv3d_k3_wgrad_split_ws_bytes(cin, cout, n, edge))
    _lib.check(lib.rf_conv3d_k3_wgrad_split(_p(x), cin, n, edge, _p(aff), _p(dz), cout, _p(scales), _p(dw), _p(ws), ws.numel(), _stream()),
               'rf_conv3d_k3_wgrad_split')
    return dw


def dgrad_split_supported(dz, weight, cin):
    n, cout, edge = dz.shape[0], dz.shape[1], dz.shape[2]
    return (ops.CONV_ARITH == 'split' and dz.numel() % 4 == 0 and bool(_lib.load().rf_conv3d_split_k3_gn_supported(cout, n, edge, cin))
            and ops.split_range_ok(weight))


@ops._device_scoped
def conv3d_gn(x, aff, w_packed, cout, relu):
    n, c, edge = x.shape[0], x.shape[1], x.shape[2]
    out = torch.empty((n, cout, edge, edge, edge), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().rf_conv3d_k3_gn(_p(x), c, _p(None), 0, n, edge, _p(aff), _p(w_packed), cout, int(relu), _p(out), _stream()), 'rf_conv3d_k3_gn')
    return out


@ops._device_scoped
def gn_backward(x, dxn, gamma, groups, eps, inv_scale=None):
    """dx, dgamma, dbeta of GroupNorm; ``inv_scale``: a device scalar when dxn arrives multiplied by 1 / inv_scale (the scaled split data gradient)"""
    n, c, edge = x.shape[0], x.shape[1], x.shape[2]
    lib = _lib.load()
    dx = torch.empty_like(x)
    dg = torch.empty(c, dtype=torch.float32, device=x.device)
    db = torch.empty(c, dtype=torch.float32, device=x.device)
    ws = _ws(x.device, lib.rf_gn_backward_ws_bytes(n, c, edge))
    _lib.check(lib.rf_gn_backward(_p(x), _p(dxn), n, c, edge, _p(gamma), groups, eps, _p(inv_scale), _p(dx), _p(dg), _p(db), _p(ws), ws.numel(), _stream()),
               'rf_gn_backward')
    return dx, dg, db


@ops._device_scoped
def conv3d_wgrad(x, aff, dz, cout):
    n, cin, edge = x.shape[0], x.shape[1], x.shape[2]
    lib = _lib.load()
    dw = torch.empty((cout, cin, 3, 3, 3), dtype=torch.float32, device=x.device)
    ws = _ws(x.device, lib.rf_conv3d_k3_wgrad_ws_bytes(cin, cout, n, edge))
    _lib.check(lib.rf_conv3d_k3_wgrad(_p(x), cin, n, edge, _p(aff), _p(dz), cout, _p(dw), _p(ws), ws.numel(), _stream()), 'rf_conv3d_k3_wgrad')
    return dw


class ConvGnRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, weight, groups, eps, skip=None, low=None):
        """``skip`` / ``low``: for a decoder layer, the two sources x was concatenated from (full-resolution skip or None, low-resolution source):
        hints for the forward kernel only (the decoder form convolves the upsampled channels in low resolution), no gradient flows to them."""
        x = x.contiguous()
        n, cin, edge = x.shape[0], x.shape[1], x.shape[2]
        cout = weight.shape[0]
        g = 1 if cin < groups else groups
        with torch.no_grad():
            aff = ops.gn_affine(x, None, gamma, beta, g, eps)
            w = weight.contiguous()
            # the inference kernels, chosen like model/unet.py:SingleConv.forward chooses them: the split-operand forms where the parameters are
            # inside their range (one batched range check per optimiser step, ops._abs_max), the fp32 kernels otherwise
            split_ok = ops.CONV_ARITH == 'split' and ops.split_range_ok(weight, gamma, beta, (cin // g) * edge ** 3)
            if split_ok and edge <= 2 and ops.conv_e2_split_supported(x, cout):
                y = ops.conv3d_e2_split_gn_relu(x, aff, ops.pack_conv3_e2_split_weight(w, edge), cout)
            elif edge == 1:
                y = ops.conv3d_gn_relu(x, None, aff, None, cout, direct_weight=w)
            elif split_ok and ops.conv_split_supported(x, None, cout):
                y = ops.conv3d_split_gn_relu(x, aff, ops.pack_conv3_split_weight(w), cout)
            elif split_ok and low is not None and ops.conv_up_split_supported(skip, low, cout):
                y = ops.conv3d_up_split_gn_relu(skip, low, aff, ops.pack_conv3_up_split_weight(w, cin - low.shape[1]), cout)
            else:
                y = ops.conv3d_gn_relu(x, None, aff, ops.pack_conv3_weight(w), cout)
        ctx.save_for_backward(x, gamma, weight, aff, y)
        ctx.groups, ctx.eps, ctx.split_ok = g, eps, bool(split_ok) and edge > 1
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, weight, aff, y = ctx.saved_tensors
        n, cin, edge = x.shape[0], x.shape[1], x.shape[2]
        cout = weight.shape[0]
        with torch.no_grad():
            inv_s = scales = None
            use_dgrad = edge >= 4 and dgrad_split_supported(y, weight, cin)
            use_wgrad = ctx.split_ok and y.numel() % 4 == 0 and bool(_lib.load().rf_conv3d_k3_wgrad_split_supported(cin, cout, n, edge))
            if use_dgrad or use_wgrad:
                dz, amax = relu_backward_amax(dy.contiguous(), y)
                ident, scales = dz_scale(amax, n, cout)
            else:
                dz = relu_backward(dy.contiguous(), y) if y.numel() % 4 == 0 else dy * (y > 0)
            if use_dgrad:
                dxn, inv_s = dgrad_split(dz, ident, weight, cin), scales[1:]
            if inv_s is not None:
                pass
            elif edge >= 2:
                wt = weight.flip(2, 3, 4).transpose(0, 1).contiguous()          # [cin, cout, 3,3,3]: the data-gradient conv's weight
                ident = torch.zeros((n, cout, 4), dtype=torch.float32, device=x.device)
                ident[..., 1] = 1.0
                dxn = conv3d_gn(dz, ident, ops.pack_conv3_weight(wt), cin, relu=False)
            else:                                                                   # 1^3 volume: only the centre tap touches data
                dxn = ops.linear(dz.reshape(n, cout), ops.pack_linear_weight(weight[:, :, 1, 1, 1].t().contiguous()), None, cin).reshape(n, cin, 1, 1, 1)
            if use_wgrad:
                dw = conv3d_wgrad_split(x, aff, dz, scales, cout)
            elif edge >= 4:
                dw = conv3d_wgrad(x, aff, dz, cout)
            else:
                # small volumes: dW = dz^T . im2col(GN(x)) through rf_linear (the operands are re-laid by torch, the products run on rf_linear_wgrad's split-K MFMA GEMM)
                xn = torch.addcmul(aff[..., 2, None, None, None], x - aff[..., 0, None, None, None], aff[..., 1, None, None, None])
                cols = F.pad(xn, (1, 1, 1, 1, 1, 1)).unfold(2, 3, 1).unfold(3, 3, 1).unfold(4, 3, 1)       # [n, cin, e,e,e, 3,3,3]
                cols = cols.permute(0, 2, 3, 4, 1, 5, 6, 7).reshape(n * edge ** 3, cin * 27)
                dzf = dz.permute(0, 2, 3, 4, 1).reshape(n * edge ** 3, cout)
                dw = ops.linear_wgrad(dzf.contiguous(), cols.contiguous()).reshape(cout, cin, 3, 3, 3)
            # (d xn of the split data gradient came scaled by s: GroupNorm backward is linear in it and takes 1 / s out)
            dx, dgamma, dbeta = gn_backward(x, dxn.contiguous(), gamma, ctx.groups, ctx.eps, inv_s)
        return dx, dgamma, dbeta, dw, None, None, None, None


class Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act, slope):
        x = x.contiguous()
        with torch.no_grad():
            y = ops.linear(x, ops.pack_linear_weight(weight.contiguous()), bias, weight.shape[0], act, slope)
        ctx.save_for_backward(x, weight, y)
        ctx.act, ctx.slope, ctx.has_bias = act, slope, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        with torch.no_grad():
            if ctx.act == ops.ACT_NONE:
                dpre = dy.contiguous()
            else:                                                   # y > 0 <=> pre-activation > 0 for ReLU and LeakyReLU(slope > 0)
                dpre = torch.where(y > 0, dy, dy * (ctx.slope if ctx.act == ops.ACT_LEAKY else 0.0)).contiguous()
            dx = ops.linear(dpre, ops.pack_linear_weight(weight.t().contiguous()), None, weight.shape[1])
            dw = ops.linear_wgrad(dpre, x)
            db = dpre.double().sum(0).float() if ctx.has_bias else None
        return dx, dw, db, None, None


@ops._device_scoped
def upsample2(x):
    ops._req(x, 'x')
    n, c, e = x.shape[0], x.shape[1], x.shape[2]
    out = torch.empty((n, c, 2 * e, 2 * e, 2 * e), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().rf_upsample3d_2(_p(x), n, c, e, _p(out), _stream()), 'rf_upsample3d_2')
    return out


@ops._device_scoped
def sumpool2(g):
    ops._req(g, 'g')
    n, c, e = g.shape[0], g.shape[1], g.shape[2]
    out = torch.empty((n, c, e // 2, e // 2, e // 2), dtype=torch.float32, device=g.device)
    _lib.check(_lib.load().rf_sumpool3d_2(_p(g), n, c, e, _p(out), _stream()), 'rf_sumpool3d_2')
    return out


@ops._device_scoped
def maxpool2_backward(x, g):
    ops._req(x, 'x'), ops._req(g, 'g')
    n, c, e = x.shape[0], x.shape[1], x.shape[2]
    dx = torch.empty_like(x)
    _lib.check(_lib.load().rf_maxpool3d_2_backward(_p(x), _p(g), n, c, e, _p(dx), _stream()), 'rf_maxpool3d_2_backward')
    return dx


class _Upsample2(torch.autograd.Function):
    """nearest x2 upsample (rf_upsample3d_2) whose backward is the 2x2x2 sum pool (rf_sumpool3d_2): torch's upsample_nearest3d_backward took 0.48 ms
    per call on a [4,16,64^3] gradient, its avg_pool3d and upsample kernels launch 32 workgroups on the retrieval backbone's shapes"""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        if x.shape[2] < 2 or x.shape[2] % 2:
            return F.interpolate(x, scale_factor=2, mode='nearest')
        return upsample2(x)

    @staticmethod
    def backward(ctx, g):
        return sumpool2(g.contiguous())


class _MaxPool2(torch.autograd.Function):
    """MaxPool3d(2): forward rf_maxpool3d_2, backward rf_maxpool3d_2_backward (the gradient to the first maximum of a cell, like torch)"""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        with torch.no_grad():
            return ops.maxpool2(x)

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        return maxpool2_backward(x, g.contiguous())


def max_pool2(x):
    return _MaxPool2.apply(x) if x.shape[2] % 2 == 0 and x.shape[2] >= 2 else F.max_pool3d(x, 2)


def conv_gn_relu(x, upsampled, gamma, beta, weight, groups, eps):
    """grad-mode SingleConv: two-source (decoder) layers are differentiated through a materialised nearest-x2 upsample + concat"""
    skip = low = None
    if upsampled is not None:
        skip, low = (x.detach().contiguous() if x is not None else None), upsampled.detach().contiguous()
        up = _Upsample2.apply(upsampled)
        x = up if x is None else torch.cat((x, up), dim=1)
    return ConvGnRelu.apply(x, gamma, beta, weight, groups, eps, skip, low)
