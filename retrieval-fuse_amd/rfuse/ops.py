"""Tensor-level wrappers over the C ABI (include/rfuse.h).  PyTorch is used for device memory and streams only:
every function takes contiguous float32 CUDA(HIP) tensors, launches on the current stream OF THE TENSORS' DEVICE (the call
is scoped to that device, whatever the caller's current device is) and returns freshly allocated outputs.  Anything else
(CPU tensors, other dtypes, tensors on different devices, missing library) raises -- no fallback.
"""
import ctypes
import functools
import sys
import weakref

import torch

from . import _lib

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
ATTN_SOFTMAX, ATTN_GUMBEL_HARD = 0, 1


def _req(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise TypeError('%s: expected a torch.Tensor' % name)
    if not t.is_cuda:
        raise RuntimeError('%s: the refinement hot path runs on the GPU only (got a %s tensor); there is no CPU fallback' % (name, t.device))
    if t.dtype != dtype:
        raise TypeError('%s: expected %s, got %s' % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError('%s: tensor must be contiguous' % name)
    return t


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    # the current stream of the CURRENT device; every public op runs inside _device_scoped, which makes the tensors' device current
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _device_scoped(fn):
    """Run ``fn`` with the device of its tensor arguments current (HIP launches go to the current device's stream), and refuse
    tensors spread over several devices.  Costs one ``current_device()`` query when the device already matches."""
    @functools.wraps(fn)
    def scoped(*args, **kwargs):
        dev = None
        flat = []
        for a in args + tuple(kwargs.values()):
            flat.extend(a) if isinstance(a, (list, tuple)) else flat.append(a)
        for a in flat:
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if dev is None:
                    dev = a.device
                elif a.device != dev:
                    raise ValueError('%s: tensors on different devices (%s and %s)' % (fn.__name__, dev, a.device))
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return scoped


def needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _no_grad_only(*tensors):
    if needs_grad(*tensors):
        raise NotImplementedError(
            'this rfuse op implements inference only; call under torch.no_grad().  Backward exists for the SingleConv layers and '
            'the attention feature encoders (rfuse/autograd.py, SURVEY.md section 8f row N4), not for this op')


_ws_cache = {}


def _workspace(device, nbytes):
    """Persistent per-(device, stream) scratch, grown on demand."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws
    return ws


class PackedWeight:
    """Caches the MFMA operand image of a Parameter; re-packs when the parameter is modified, moved or replaced."""

    def __init__(self, kind):
        self.kind = kind        # 'conv3' | 'linear' | 'convv' | 'conv3up'
        self._key = None
        self._packed = None
        self._ready = _PackedReady()

    def get(self, w, *extra):
        key = (w.data_ptr(), w._version, tuple(w.shape), w.device) + extra
        if key != self._key:
            self._packed = {'conv3': pack_conv3_weight, 'linear': pack_linear_weight, 'convv': pack_convv_weight,
                            'convvl': pack_convv_lds_weight, 'convvv': pack_convv_valu_weight, 'conv3up': pack_conv3_up_weight, 'conv3ups': pack_conv3_up_split_weight, 'conv3s': pack_conv3_split_weight, 'conv3e2': pack_conv3_e2_split_weight, 'convvs': pack_convv_split_weight, 'convvpg': pack_convv_split_pg_weight}[self.kind](w, *extra)
            self._key = key
            self._ready.packed_on(w.device)
        else:
            self._ready.wait(w.device, self._packed)
        return self._packed


class _PackedReady:
    """Orders readers of a packed operand image after the (stream-ordered) pack kernel: the image is packed on whichever
    stream first needs it, and the engine later reads it from other streams (the backbone runs on a side stream)."""

    def __init__(self):
        self._event = None
        self._ok_streams = set()

    def packed_on(self, device):
        with torch.cuda.device(device):
            st = torch.cuda.current_stream()
            self._event = torch.cuda.Event()
            self._event.record(st)
            self._ok_streams = {st.cuda_stream}

    def wait(self, device, packed=None):
        st = torch.cuda.current_stream(device)
        if st.cuda_stream not in self._ok_streams and not torch.cuda.is_current_stream_capturing():
            st.wait_event(self._event)
            self._ok_streams.add(st.cuda_stream)
            # a reader on another stream: when the image is dropped (re-pack after an optimizer step / load_state_dict) the caching allocator
            # must not hand its block out again before this stream's kernels have read it
            for t in (packed if isinstance(packed, (tuple, list)) else (packed,)):
                if isinstance(t, torch.Tensor):
                    t.record_stream(st)


# ------------------------------------------------------------------------------------------------ U-Net primitives

def pack_conv3_weight(w):
    _req(w.detach(), 'conv weight')
    cout, cin = w.shape[0], w.shape[1]
    if tuple(w.shape[2:]) != (3, 3, 3):
        raise ValueError('pack_conv3_weight: expected an OIDHW 3x3x3 weight, got %s' % (tuple(w.shape),))
    lib = _lib.load()
    out = torch.empty(lib.rf_conv3_packed_floats(cout, cin), dtype=torch.float32, device=w.device)
    _lib.check(lib.rf_conv3_pack_weight(_p(w.detach()), cout, cin, _p(out), _stream()), 'rf_conv3_pack_weight')
    return out


def _src_dims(src0, src1):
    n = (src0 if src0 is not None else src1).shape[0]
    c0 = src0.shape[1] if src0 is not None else 0
    c1 = src1.shape[1] if src1 is not None else 0
    edge = src0.shape[2] if src0 is not None else 2 * src1.shape[2]
    for t, e in ((src0, edge), (src1, edge // 2)):
        if t is not None and (t.dim() != 5 or tuple(t.shape[2:]) != (e, e, e) or t.shape[0] != n):
            raise ValueError('expected cubic NCDHW volumes with matching batch, got %s' % (tuple(t.shape),))
    return n, c0, c1, edge


def gn_affine(src0, src1, gamma, beta, groups, eps=1e-5):
    """GroupNorm of cat(src0, up2(src1)) folded to the per-(n, c) triple the conv kernels apply: [n, C, 4] float32 =
    (center, scale, shift, 0) with y = (x - center) * scale + shift.  Either source may be None."""
    for t, nm in ((src0, 'src0'), (src1, 'src1')):
        if t is not None:
            _req(t, nm)
    _req(gamma.detach(), 'gamma'), _req(beta.detach(), 'beta')
    n, c0, c1, edge = _src_dims(src0, src1)
    c = c0 + c1
    if c < groups:
        groups = 1                                           # model/unet.py:62-63
    dev = gamma.device
    lib = _lib.load()
    aff = torch.empty((n, c, 4), dtype=torch.float32, device=dev)
    st0, st1 = _fresh_stats(src0), _fresh_stats(src1)
    if USE_FUSED_STATS and (src0 is None or st0 is not None) and (src1 is None or st1 is not None):
        # the producers (conv / max-pool epilogues) already emitted per-tile sums: no re-read of the activations
        _lib.check(lib.rf_gn_from_stats(_p(st0[0]) if st0 else _p(None), c0, st0[1] if st0 else 0,
                                        _p(st1[0]) if st1 else _p(None), c1, st1[1] if st1 else 0, n, edge,
                                        _p(gamma.detach()), _p(beta.detach()), groups, eps, _p(aff), _stream()), 'rf_gn_from_stats')
        return aff
    nbytes = lib.rf_gn_stats_ws_bytes(n, groups)
    ws = _workspace(dev, nbytes)
    _lib.check(lib.rf_gn_stats(_p(src0), c0, _p(src1), c1, n, edge, _p(gamma.detach()), _p(beta.detach()), groups, eps,
                               _p(aff), _p(ws), ws.numel(), _stream()), 'rf_gn_stats')
    return aff


def _check_affine(aff, n, c):
    _req(aff, 'gn affine')
    if tuple(aff.shape) != (n, c, 4):
        raise ValueError('gn affine: expected [%d, %d, 4] (from ops.gn_affine), got %s' % (n, c, tuple(aff.shape)))
    return aff.device


USE_FUSED_STATS = True          # producers attach (stats, tiles, version) to their outputs as ``tensor._rf_stats``


def _fresh_stats(t):
    """Producer-side statistics of ``t`` if they are attached and ``t`` has not been modified in place since."""
    st = getattr(t, '_rf_stats', None) if t is not None else None
    return st if st is not None and st[2] == t._version else None


def _conv_launch(lib, src0, c0, src1, c1, n, edge, aff, w_packed, cout, out):
    """MFMA conv launch; emits the output's GroupNorm statistics for the next layer when the tiling supports it."""
    tiles = lib.rf_conv3d_stats_tiles(c0, c1, n, edge, cout) if USE_FUSED_STATS else 0
    if tiles > 0:
        stats = torch.empty((n, cout, tiles, 2), dtype=torch.float64, device=out.device)
        _lib.check(lib.rf_conv3d_k3_gn_relu_stats(_p(src0), c0, _p(src1), c1, n, edge, _p(aff), _p(w_packed), cout, _p(out),
                                                  _p(stats), _stream()), 'rf_conv3d_k3_gn_relu_stats')
        out._rf_stats = (stats, tiles, out._version)
    else:
        _lib.check(lib.rf_conv3d_k3_gn_relu(_p(src0), c0, _p(src1), c1, n, edge, _p(aff), _p(w_packed), cout, _p(out),
                                            _stream()), 'rf_conv3d_k3_gn_relu')


# bench.py hook: time selected conv launches with HIP events recorded on the launch stream
conv_event_filter = None        # callable(cin, cout, edge, n) -> bool
conv_events = []                # [(start_event, end_event, flop issued, label)]; label = (entry point, arithmetic, (c0, c1, n, edge, cout))


def conv3d_gn_relu(src0, src1, aff, w_packed, cout, direct_weight=None):
    """ReLU(conv3(GN(cat(src0, up2(src1))))).  1^3 volumes (and ``direct_weight`` calls) use the direct kernel."""
    n, c0, c1, edge = _src_dims(src0, src1)
    dev = _check_affine(aff, n, c0 + c1)
    out = torch.empty((n, cout, edge, edge, edge), dtype=torch.float32, device=dev)
    lib = _lib.load()
    if conv_event_filter is not None and direct_weight is None and conv_event_filter(c0 + c1, cout, edge, n):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        _conv_launch(lib, src0, c0, src1, c1, n, edge, aff, w_packed, cout, out)
        ev1.record()
        conv_events.append((ev0, ev1, 2.0 * 27 * (c0 + c1) * cout * edge ** 3 * n, ('rf_conv3d_k3_gn_relu', 'fp32 MFMA', (c0, c1, n, edge, cout))))
        return out
    if direct_weight is not None:
        _lib.check(lib.rf_conv3d_k3_gn_relu_direct(_p(src0), c0, _p(src1), c1, n, edge, _p(aff), _p(direct_weight.detach()),
                                                   cout, _p(out), _stream()), 'rf_conv3d_k3_gn_relu_direct')
    else:
        _conv_launch(lib, src0, c0, src1, c1, n, edge, aff, w_packed, cout, out)
    return out


USE_FUSED_POOL = True           # False: encoders always run the stand-alone max-pool kernel


def conv_pool_supported(src0, src1, cout):
    """True when rf_conv3d_k3_gn_relu_pool (MaxPool3d(2) fused into the conv epilogue) takes this shape."""
    if not USE_FUSED_POOL:
        return False
    n, c0, c1, edge = _src_dims(src0, src1)
    return bool(_lib.load().rf_conv3d_pool_supported(c0, c1, n, edge, cout))


def conv3d_gn_relu_pool(src0, src1, aff, w_packed, cout, keep_full=True):
    """(ReLU(conv3(GN(x))) or None, its MaxPool3d(2)); with keep_full=False the full-resolution tensor is never written."""
    n, c0, c1, edge = _src_dims(src0, src1)
    dev = _check_affine(aff, n, c0 + c1)
    lib = _lib.load()
    out = torch.empty((n, cout, edge, edge, edge), dtype=torch.float32, device=dev) if keep_full else None
    pooled = torch.empty((n, cout, edge // 2, edge // 2, edge // 2), dtype=torch.float32, device=dev)
    stats = pstats = None
    if USE_FUSED_STATS:
        tiles = lib.rf_conv3d_stats_tiles(c0, c1, n, edge, cout)
        pstats = torch.empty((n, cout, tiles, 2), dtype=torch.float64, device=dev)
        if keep_full:
            stats = torch.empty((n, cout, tiles, 2), dtype=torch.float64, device=dev)
    _lib.check(lib.rf_conv3d_k3_gn_relu_pool(_p(src0), c0, _p(src1), c1, n, edge, _p(aff), _p(w_packed), cout, _p(out), _p(stats),
                                             _p(pooled), _p(pstats), _stream()), 'rf_conv3d_k3_gn_relu_pool')
    if stats is not None:
        out._rf_stats = (stats, tiles, out._version)
    if pstats is not None:
        pooled._rf_stats = (pstats, tiles, pooled._version)
    return out, pooled


def pack_conv3_up_weight(w, c0):
    """Weight image of the decoder-form conv: the first c0 input channels are the skip source, the rest the upsampled one."""
    _req(w.detach(), 'conv weight')
    cout, cin = w.shape[0], w.shape[1]
    if tuple(w.shape[2:]) != (3, 3, 3) or not 0 <= c0 < cin:
        raise ValueError('pack_conv3_up_weight: expected an OIDHW 3x3x3 weight and 0 <= c0 < cin, got %s, c0=%d' % (tuple(w.shape), c0))
    lib = _lib.load()
    out = torch.empty(lib.rf_conv3_up_packed_floats(cout, c0, cin - c0), dtype=torch.float32, device=w.device)
    _lib.check(lib.rf_conv3_up_pack_weight(_p(w.detach()), cout, c0, cin - c0, _p(out), _stream()), 'rf_conv3_up_pack_weight')
    return out


def conv_up_supported(src0, src1, cout):
    """True when the parity-split decoder kernel (rf_conv3d_up_k3_gn_relu) takes this (skip, low-res) pair."""
    if src1 is None or not USE_CONV_UP:
        return False
    n, c0, c1, edge = _src_dims(src0, src1)
    return bool(_lib.load().rf_conv3d_up_supported(c0, c1, n, edge, cout))


USE_CONV_UP = True              # False: decoder convs run the generic kernel on the (virtually) upsampled source


def conv3d_up_gn_relu(src0, src1, aff, w_up_packed, cout):
    """ReLU(conv3(GN(cat(src0, up2(src1))))) with the upsampled channels convolved in low resolution."""
    n, c0, c1, edge = _src_dims(src0, src1)
    out = torch.empty((n, cout, edge, edge, edge), dtype=torch.float32, device=_check_affine(aff, n, c0 + c1))
    lib = _lib.load()
    stats = None
    if USE_FUSED_STATS:
        tiles = lib.rf_conv3d_up_stats_tiles(c0, c1, n, edge, cout)
        stats = torch.empty((n, cout, tiles, 2), dtype=torch.float64, device=out.device)
    timed = conv_event_filter is not None and conv_event_filter(c0 + c1, cout, edge, n)
    if timed:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    _lib.check(lib.rf_conv3d_up_k3_gn_relu(_p(src0), c0, _p(src1), c1, n, edge, _p(aff), _p(w_up_packed), cout, _p(out),
                                           _p(stats), _stream()), 'rf_conv3d_up_k3_gn_relu')
    if timed:
        ev1.record()
        conv_events.append((ev0, ev1, conv_up_issued_flops(c0, c1, n, edge, cout), ('rf_conv3d_up_k3_gn_relu', 'fp32 MFMA', (c0, c1, n, edge, cout))))
    if stats is not None:
        out._rf_stats = (stats, tiles, out._version)
    return out


# Arithmetic of the heavy convolutions that have a split-operand form (csrc/conv3d_up_split.hip): 'split' = fp32 operands carried as
# two f16 pieces on the F16 matrix cores (exact products, fp32 accumulation; measured closer to float64 than the fp32 MFMA chain),
# 'fp32' = v_mfma_f32_16x16x4_f32 everywhere.  An API switch, not an environment variable.
CONV_ARITH = 'split'

# ---- value range of the split-operand (f16 x f16) forms.  They carry activations scaled by 2^-4 and weights scaled by 2^4 as pairs of f16 values
# (csrc/conv3d_split.hip:20-24): |GroupNorm output| above 65504 * 16 or a (pre-summed) weight above 65504 / 16 would CLAMP where the reference
# computes in fp32 range.  Both are decidable from the parameters alone, once per parameter version:
#   weights      max |w| <= SPLIT_MAX_ABS_WEIGHT (the decoder form adds up to 8 taps before it splits: 65504 / 16 / 8);
#   activations  a GroupNorm output is (x - mean) * rstd * gamma + beta with |x - mean| * rstd <= sqrt(elements of the group), so
#                max |gamma| * sqrt(group elements) + max |beta| <= SPLIT_MAX_ABS_ACT rules saturation out for every input.
# A layer outside the range runs the fp32 kernels (model/unet.py:SingleConv) -- same results as the reference's fp32 path, no clamp.
SPLIT_MAX_ABS_WEIGHT = 65504.0 / 16 / 8
SPLIT_MAX_ABS_ACT = 65504.0 * 16
_range_cache = {}                       # (data_ptr, shape, device) -> (version, max |t|)
_range_seen = weakref.WeakValueDictionary()     # id -> every tensor whose range was asked for (the parameters of the networks in use)


def _range_key(t):
    return (t.data_ptr(), tuple(t.shape), t.device)


def _abs_max(t):
    """max |t| of a parameter, cached per parameter version.  A training step changes every parameter at once (the optimiser's step): the
    first miss after it refreshes ALL the parameters seen so far on that device in one ``torch._foreach_norm`` and one host sync, instead
    of one sync per parameter (three per layer)."""
    key = _range_key(t)
    hit = _range_cache.get(key)
    if hit is not None and hit[0] == t._version:
        return hit[1]
    if t.is_cuda and torch.cuda.is_current_stream_capturing():
        raise RuntimeError('range check of a parameter inside a graph capture: run one step before capturing (RefinementEngine.capture_graph does)')
    if len(_range_cache) > 8192:
        _range_cache.clear()
    _range_seen[id(t)] = t
    stale = [u for u in list(_range_seen.values())
             if u.device == t.device and u.dtype == t.dtype and u.numel() > 0 and _range_cache.get(_range_key(u), (None,))[0] != u._version]
    if t.is_cuda and len(stale) > 1:
        vals = torch.stack(torch._foreach_norm([u.detach() for u in stale], float('inf'))).tolist()       # one host sync
        for u, val in zip(stale, vals):
            _range_cache[_range_key(u)] = (u._version, float(val))
        return _range_cache[key][1]
    v = float(t.detach().abs().max().item())                                  # one host sync
    _range_cache[key] = (t._version, v)
    return v


def split_range_ok(weight, gamma=None, beta=None, group_elements=0):
    """True when the split-operand forms cannot saturate for this layer, whatever the input (see SPLIT_MAX_ABS_* above)."""
    w_ok = _abs_max(weight) <= SPLIT_MAX_ABS_WEIGHT                      # NaN / inf weights compare False: fp32 path, like the reference
    if gamma is None:
        return w_ok
    return w_ok and _abs_max(gamma) * (float(group_elements) ** 0.5) + _abs_max(beta) <= SPLIT_MAX_ABS_ACT


def pack_conv3_split_weight(w):
    """f16 fragment-order weight image of the split-operand box conv (csrc/conv3d_split.hip)."""
    _req(w.detach(), 'conv weight')
    cout, cin = w.shape[0], w.shape[1]
    if tuple(w.shape[2:]) != (3, 3, 3):
        raise ValueError('pack_conv3_split_weight: expected an OIDHW 3x3x3 weight, got %s' % (tuple(w.shape),))
    lib = _lib.load()
    out = torch.empty(lib.rf_conv3_split_packed_bytes(cout, cin), dtype=torch.uint8, device=w.device)
    _lib.check(lib.rf_conv3_split_pack_weight(_p(w.detach()), cout, cin, _p(out), _stream()), 'rf_conv3_split_pack_weight')
    return out


def conv_split_supported(src0, src1, cout):
    """True when the split-operand box kernel (rf_conv3d_split_k3_gn_relu) takes this source and is switched on."""
    if src1 is not None or src0 is None or CONV_ARITH != 'split':
        return False
    n, c0, c1, edge = _src_dims(src0, src1)
    return bool(_lib.load().rf_conv3d_split_supported(c0, c1, n, edge, cout))


def conv3d_split_gn_relu(src, aff, w_split_packed, cout, pool=None):
    """ReLU(conv3(GN(src))) on the F16 matrix cores by operand splitting.  pool: None -> out; 'also' -> (out, maxpool2(out));
    'only' -> (None, maxpool2(out)) with the full-resolution tensor never written.  Statistics of what is written ride along."""
    n, cin, _, edge = _src_dims(src, None)
    if edge == 4 and pool is not None:                             # the 4^3 form has no fused max-pool: pool its output (statistics ride along)
        out = conv3d_split_gn_relu(src, aff, w_split_packed, cout)
        return (out if pool == 'also' else None), maxpool2(out)
    dev = _check_affine(aff, n, cin)
    lib = _lib.load()
    out = torch.empty((n, cout, edge, edge, edge), dtype=torch.float32, device=dev) if pool != 'only' else None
    pooled = torch.empty((n, cout, edge // 2, edge // 2, edge // 2), dtype=torch.float32, device=dev) if pool is not None else None
    tiles = max(1, (edge // 8) ** 3)
    stats = pstats = None
    if USE_FUSED_STATS:
        stats = torch.empty((n, cout, tiles, 2), dtype=torch.float64, device=dev) if out is not None else None
        pstats = torch.empty((n, cout, tiles, 2), dtype=torch.float64, device=dev) if pooled is not None else None
    timed = conv_event_filter is not None and conv_event_filter(cin, cout, edge, n)
    if timed:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    _lib.check(lib.rf_conv3d_split_k3_gn_relu(_p(src), cin, n, edge, _p(aff), _p(w_split_packed), cout, _p(out), _p(stats), _p(pooled), _p(pstats),
                                              _stream()), 'rf_conv3d_split_k3_gn_relu')
    if timed:
        ev1.record()
        conv_events.append((ev0, ev1, conv_split_issued_flops(cin, n, edge, cout), ('rf_conv3d_split_k3_gn_relu', 'f16 split', (cin, 0, n, edge, cout))))
    if stats is not None:
        out._rf_stats = (stats, tiles, out._version)
    if pstats is not None:
        pooled._rf_stats = (pstats, tiles, pooled._version)
    return out if pool is None else (out, pooled)


def pack_conv3_e2_split_weight(w, edge=2):
    """f16 fragment-order image of the dense GEMM form of a 3x3x3 conv on whole 2^3 (edge 2) or 1^3 (edge 1) volumes (csrc/conv3d_e2_split.hip)."""
    _req(w.detach(), 'conv weight')
    cout, cin = w.shape[0], w.shape[1]
    if tuple(w.shape[2:]) != (3, 3, 3) or edge not in (1, 2):
        raise ValueError('pack_conv3_e2_split_weight: expected an OIDHW 3x3x3 weight and edge 1 or 2, got %s, edge %r' % (tuple(w.shape), edge))
    lib = _lib.load()
    out = torch.empty(lib.rf_conv3_e2_split_packed_bytes(cout, cin, edge), dtype=torch.uint8, device=w.device)
    _lib.check(lib.rf_conv3_e2_split_pack_weight(_p(w.detach()), cout, cin, edge, _p(out), _stream()), 'rf_conv3_e2_split_pack_weight')
    return out


def conv_e2_split_supported(src, cout):
    """True when the 2^3 GEMM form (rf_conv3d_e2_split_k3_gn_relu) takes this source and is switched on."""
    if src is None or CONV_ARITH != 'split':
        return False
    n, c0, _, edge = _src_dims(src, None)
    return bool(_lib.load().rf_conv3d_e2_split_supported(c0, n, edge, cout))


def conv3d_e2_split_gn_relu(src, aff, w_e2_packed, cout):
    """ReLU(conv3(GN(src))) on whole 2^3 / 1^3 volumes as one dense GEMM on the F16 matrix cores; the output's statistics ride along."""
    n, cin, _, edge = _src_dims(src, None)
    dev = _check_affine(aff, n, cin)
    out = torch.empty((n, cout, edge, edge, edge), dtype=torch.float32, device=dev)
    stats = torch.empty((n, cout, 1, 2), dtype=torch.float64, device=dev) if USE_FUSED_STATS else None
    _lib.check(_lib.load().rf_conv3d_e2_split_k3_gn_relu(_p(src), cin, n, edge, _p(aff), _p(w_e2_packed), cout, _p(out), _p(stats), _stream()),
               'rf_conv3d_e2_split_k3_gn_relu')
    if stats is not None:
        out._rf_stats = (stats, 1, out._version)
    return out


USE_PRESPLIT = True             # False: every layer normalises and splits its own input (the round-2 routes; kept for cross-checks)


def cin1_presplit_supported(x, cout, next_groups, next_cout):
    """True when a level-0 DoubleConv on this input can run as rf_conv3d_cin1_presplit -> rf_conv3d_split_pre_k3_relu."""
    if not USE_PRESPLIT or CONV_ARITH != 'split' or x is None or x.shape[1] != 1:
        return False
    n, edge = x.shape[0], x.shape[2]
    lib = _lib.load()
    return bool(lib.rf_conv3d_cin1_presplit_supported(n, edge, cout, next_groups)) and bool(lib.rf_conv3d_split_pre_supported(cout, n, edge, next_cout))


def conv3d_cin1_presplit(x, in_gamma, in_beta, in_eps, w_packed, cout, next_gamma, next_beta, next_groups, eps):
    """ReLU(conv3(GN(x))) of a 1-channel input (its GroupNorm computed in the kernel), emitted as the pre-split input of the NEXT layer (that
    layer's GroupNorm applied): uint8 buffer."""
    _req(x, 'x')
    n, edge = x.shape[0], x.shape[2]
    lib = _lib.load()
    out = torch.empty(lib.rf_split_act_bytes(n, cout, edge), dtype=torch.uint8, device=x.device)
    _lib.check(lib.rf_conv3d_cin1_presplit(_p(x), n, edge, _p(in_gamma.detach()), _p(in_beta.detach()), in_eps, _p(w_packed), cout, _p(next_gamma.detach()),
                                           _p(next_beta.detach()), next_groups, eps, _p(out), _stream()), 'rf_conv3d_cin1_presplit')
    return out


def conv_split_pointwise_supported(x, cout):
    return CONV_ARITH == 'split' and x is not None and bool(_lib.load().rf_conv3d_split_pointwise_supported(x.shape[1], x.shape[0], x.shape[2], cout))


def conv3d_split_pointwise_tanh(x, gn_affine_t, w_split_packed, cout, pw_w, pw_b, post_add=0.0, post_mul=1.0):
    """(tanh(Conv3d(cout, 1, 1)(relu(conv(GN(x))))) + post_add) * post_mul with the pointwise head in the conv's epilogue: [n, 1, e, e, e]"""
    _req(x, 'x')
    n, cin, edge = x.shape[0], x.shape[1], x.shape[2]
    if pw_w.shape[0] != 1:
        raise NotImplementedError('conv3d_split_pointwise_tanh: one output channel (Conv3d(nf,1,1), model/refinement.py:54)')
    out = torch.empty((n, 1, edge, edge, edge), dtype=torch.float32, device=x.device)
    timed = conv_event_filter is not None and conv_event_filter(cin, cout, edge, n)
    if timed:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    _lib.check(_lib.load().rf_conv3d_split_k3_gn_relu_pointwise_tanh(_p(x), cin, n, edge, _p(gn_affine_t), _p(w_split_packed), cout, _p(pw_w.detach()), _p(pw_b.detach()),
                                                                     post_add, post_mul, _p(out), _stream()), 'rf_conv3d_split_k3_gn_relu_pointwise_tanh')
    if timed:
        ev1.record()
        conv_events.append((ev0, ev1, conv_split_issued_flops(cin, n, edge, cout), ('rf_conv3d_split_k3_gn_relu', 'f16 split', (cin, 0, n, edge, cout))))
    return out


def conv_split_presplit_supported(x, cout, next_groups):
    """the split box conv on whole 8^3 samples can hand its output to the next SingleConv pre-split (rf_conv3d_split_presplit)"""
    return (USE_PRESPLIT and CONV_ARITH == 'split' and x is not None
            and bool(_lib.load().rf_conv3d_split_presplit_supported(x.shape[1], x.shape[0], x.shape[2], cout, next_groups)))


def conv3d_split_presplit(x, gn_affine_t, w_split_packed, cout, next_gamma, next_beta, next_groups, eps):
    """relu(conv(GN(x))) of whole 8^3 samples (<= 16 couts), emitted as the pre-split input of the NEXT layer: uint8 buffer for conv3d_split_pre_relu"""
    _req(x, 'x')
    n, cin, edge = x.shape[0], x.shape[1], x.shape[2]
    lib = _lib.load()
    out = torch.empty(lib.rf_split_act_bytes(n, cout, edge), dtype=torch.uint8, device=x.device)
    _lib.check(lib.rf_conv3d_split_presplit(_p(x), cin, n, edge, _p(gn_affine_t), _p(w_split_packed), cout, _p(next_gamma.detach()), _p(next_beta.detach()),
                                            next_groups, eps, _p(out), _p(None), _stream()), 'rf_conv3d_split_presplit')
    return out


def conv_up_split_presplit_supported(x, upsampled, cout, next_groups):
    """the decoder-form split conv on whole 8^3 samples can hand its output to the next SingleConv pre-split (rf_conv3d_up_split_presplit)"""
    if not USE_PRESPLIT or CONV_ARITH != 'split' or upsampled is None:
        return False
    c0 = x.shape[1] if x is not None else 0
    return bool(_lib.load().rf_conv3d_up_split_presplit_supported(c0, upsampled.shape[1], upsampled.shape[0], 2 * upsampled.shape[2], cout, next_groups))


def conv_up_split_presplit_pm_supported(x, upsampled, cout, next_groups, next_cout):
    """the decoder pair can hand over in PARITY-MAJOR slot order: the persistent producer (rf_conv3d_up_split_presplit_pm) AND the persistent consumer
    (rf_conv3d_split_pre_pm_k3_relu) take the shapes"""
    if not USE_PRESPLIT or not USE_PRESPLIT_PM or CONV_ARITH != 'split' or upsampled is None or x is None:
        return False
    lib = _lib.load()
    n, edge = upsampled.shape[0], 2 * upsampled.shape[2]
    return bool(lib.rf_conv3d_up_split_presplit_pm_supported(x.shape[1], upsampled.shape[1], n, edge, cout, next_groups)) and \
        bool(lib.rf_conv3d_split_pre_pm_supported(cout, n, edge, next_cout))


def conv3d_up_split_presplit(x, upsampled, gn_affine_t, w_packed, cout, next_gamma, next_beta, next_groups, eps, parity_major=False):
    """relu(conv(GN([x, up(upsampled)]))) of whole 8^3 samples, emitted as the pre-split input of the NEXT layer (its GroupNorm applied from the
    sample's own statistics): uint8 buffer for conv3d_split_pre_relu (parity_major: its voxel slots in the order of rf_conv3d_up_split_presplit_pm, for
    conv3d_split_pre_relu(..., parity_major=True))"""
    _req(upsampled, 'upsampled')
    n, edge = upsampled.shape[0], 2 * upsampled.shape[2]
    c0 = x.shape[1] if x is not None else 0
    lib = _lib.load()
    out = torch.empty(lib.rf_split_act_bytes(n, cout, edge), dtype=torch.uint8, device=upsampled.device)
    c1 = upsampled.shape[1]
    timed = conv_event_filter is not None and conv_event_filter(c0 + c1, cout, edge, n)
    if timed:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    entry = 'rf_conv3d_up_split_presplit_pm' if parity_major else 'rf_conv3d_up_split_presplit'
    _lib.check(getattr(lib, entry)(_p(x), c0, _p(upsampled), c1, n, edge, _p(gn_affine_t), _p(w_packed), cout, _p(next_gamma.detach()),
                                   _p(next_beta.detach()), next_groups, eps, _p(out), _p(None), _stream()), entry)
    if timed:
        ev1.record()
        conv_events.append((ev0, ev1, conv_up_split_issued_flops(c0, c1, n, edge, cout), ('rf_conv3d_up_split_presplit', 'f16 split', (c0, c1, n, edge, cout))))
    return out


USE_PRESPLIT_PM = True          # False: the decoder pair hands over in the linear slot order (cross-checks)
USE_PREPOOL = True              # False: the level-0 max-pool hands an fp32 tensor to the next level (the round-3 route; kept for cross-checks)


class PreSplit:
    """A pre-split activation tensor (rf_split_act_bytes: per (sample, 8-channel group) an h plane and an l plane of 16-byte voxel slots), already normalised
    by its consumer's GroupNorm -- what the producers of DESIGN 4.8 hand to rf_conv3d_split_pre_*; the shape travels with the bytes."""

    def __init__(self, data, n, channels, edge):
        self.data, self.n, self.channels, self.edge = data, n, channels, edge

    @property
    def device(self):
        return self.data.device


def conv_split_supported_shape(n, cin, edge, cout):
    """conv_split_supported for a tensor that does not exist yet (a concatenation about to be written): [n, cin, edge^3] -> cout"""
    return CONV_ARITH == 'split' and bool(_lib.load().rf_conv3d_split_supported(int(cin), 0, int(n), int(edge), int(cout)))


def conv_split_pre_pool_presplit_supported(cin, n, edge, cout, next_groups):
    return USE_PREPOOL and USE_PRESPLIT and CONV_ARITH == 'split' and bool(_lib.load().rf_conv3d_split_pre_pool_presplit_supported(cin, n, edge, cout, next_groups))


def conv3d_split_pre_relu_pool_presplit(pre, cin, n, edge, w_split_packed, cout, next_gamma, next_beta, next_groups, eps):
    """conv3d_split_pre_relu(pool='only') whose pooled output is emitted pre-split for the next level's first conv: -> (None, PreSplit of the pooled tensor).
    No fp32 pooled tensor exists: the kernel keeps the pooled values of the sample in flight in a per-workgroup scratch slot (one slot of cout x 512 floats per
    persistent workgroup: 2048 workgroups on MI355X -> 64 MB allocated at cout = 16; a launch touches min(n, workgroups) slots, each re-used sample after sample, so
    the live part stays cache-resident)."""
    dev = pre.device
    lib = _lib.load()
    half = edge // 2
    scratch = torch.empty(lib.rf_conv3d_split_pre_pool_presplit_scratch_floats(cout), dtype=torch.float32, device=dev)
    out = torch.empty(lib.rf_split_act_bytes(n, cout, half), dtype=torch.uint8, device=dev)
    _lib.check(lib.rf_conv3d_split_pre_k3_relu_pool_presplit(_p(pre), cin, n, edge, _p(w_split_packed), cout, _p(scratch), _p(None), _p(next_gamma.detach()),
                                                             _p(next_beta.detach()), next_groups, eps, _p(out), _stream()), 'rf_conv3d_split_pre_k3_relu_pool_presplit')
    return None, PreSplit(out, n, cout, half)


def conv_split_pre_presplit_supported(cin, n, edge, cout, next_groups):
    return USE_PRESPLIT and CONV_ARITH == 'split' and bool(_lib.load().rf_conv3d_split_pre_presplit_supported(cin, n, edge, cout, next_groups))


def conv3d_split_pre_presplit(pre, w_split_packed, cout, next_gamma, next_beta, next_groups, eps):
    """relu(conv(.)) of a PreSplit input (whole 8^3 samples), emitted as the PreSplit input of the NEXT layer"""
    lib = _lib.load()
    out = torch.empty(lib.rf_split_act_bytes(pre.n, cout, pre.edge), dtype=torch.uint8, device=pre.device)
    _lib.check(lib.rf_conv3d_split_pre_presplit(_p(pre.data), pre.channels, pre.n, pre.edge, _p(w_split_packed), cout, _p(next_gamma.detach()), _p(next_beta.detach()),
                                                next_groups, eps, _p(out), _p(None), _stream()), 'rf_conv3d_split_pre_presplit')
    return PreSplit(out, pre.n, cout, pre.edge)


def conv3d_split_pre_relu(pre, cin, n, edge, w_split_packed, cout, pool=None, parity_major=False):
    """conv3d_split_gn_relu on a pre-split input (already normalised for this layer and split by its producer; parity_major: in the slot order of
    conv3d_up_split_presplit(..., parity_major=True))."""
    dev = pre.device
    lib = _lib.load()
    out = torch.empty((n, cout, edge, edge, edge), dtype=torch.float32, device=dev) if pool != 'only' else None
    pooled = torch.empty((n, cout, edge // 2, edge // 2, edge // 2), dtype=torch.float32, device=dev) if pool is not None else None
    tiles = 1 if parity_major else lib.rf_conv3d_split_pre_stats_tiles(cin, n, edge, cout)      # one per 8^3 box, or one per sample (persistent form)
    stats = pstats = None
    if USE_FUSED_STATS:
        stats = torch.empty((n, cout, tiles, 2), dtype=torch.float64, device=dev) if out is not None else None
        pstats = torch.empty((n, cout, tiles, 2), dtype=torch.float64, device=dev) if pooled is not None else None
    timed = conv_event_filter is not None and conv_event_filter(cin, cout, edge, n)
    if timed:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    entry = 'rf_conv3d_split_pre_pm_k3_relu' if parity_major else 'rf_conv3d_split_pre_k3_relu'
    _lib.check(getattr(lib, entry)(_p(pre), cin, n, edge, _p(w_split_packed), cout, _p(out), _p(stats), _p(pooled), _p(pstats), _stream()), entry)
    if timed:
        ev1.record()
        conv_events.append((ev0, ev1, conv_split_issued_flops(cin, n, edge, cout), ('rf_conv3d_split_pre_k3_relu', 'f16 split', (cin, 0, n, edge, cout))))
    if stats is not None:
        out._rf_stats = (stats, tiles, out._version)
    if pstats is not None:
        pooled._rf_stats = (pstats, tiles, pooled._version)
    if pool is None:
        return out
    return out, pooled


def conv_split_issued_flops(cin, n, edge, cout):
    """f16 flop rf_conv3d_split_k3_gn_relu ISSUES: three MFMAs per k-step of 32, 7 k-steps (28 tap slots) per 8 input channels, on
    round_up(cout, 16) columns."""
    return 2.0 * 3 * -(-cin // 8) * 7 * 32 * (-(-cout // 16) * 16) * edge ** 3 * n


def pack_conv3_up_split_weight(w, c0):
    """f16 fragment-order weight image of the split-operand decoder conv (pre-sums in float64, split from the float64 value)."""
    _req(w.detach(), 'conv weight')
    cout, cin = w.shape[0], w.shape[1]
    if tuple(w.shape[2:]) != (3, 3, 3) or not 0 <= c0 < cin:
        raise ValueError('pack_conv3_up_split_weight: expected an OIDHW 3x3x3 weight and 0 <= c0 < cin, got %s, c0=%d' % (tuple(w.shape), c0))
    lib = _lib.load()
    out = torch.empty(lib.rf_conv3_up_split_packed_bytes(cout, c0, cin - c0), dtype=torch.uint8, device=w.device)
    _lib.check(lib.rf_conv3_up_split_pack_weight(_p(w.detach()), cout, c0, cin - c0, _p(out), _stream()), 'rf_conv3_up_split_pack_weight')
    return out


def conv_up_split_supported(src0, src1, cout):
    """True when the split-operand decoder kernel (rf_conv3d_up_split_k3_gn_relu) takes this (skip, low-res) pair and is switched on."""
    if src1 is None or not USE_CONV_UP or CONV_ARITH != 'split':
        return False
    n, c0, c1, edge = _src_dims(src0, src1)
    return bool(_lib.load().rf_conv3d_up_split_supported(c0, c1, n, edge, cout))


def conv3d_up_split_gn_relu(src0, src1, aff, w_split_packed, cout):
    """conv3d_up_gn_relu on the F16 matrix cores by operand splitting (whole 8^3 samples)."""
    n, c0, c1, edge = _src_dims(src0, src1)
    out = torch.empty((n, cout, edge, edge, edge), dtype=torch.float32, device=_check_affine(aff, n, c0 + c1))
    lib = _lib.load()
    tiles = lib.rf_conv3d_up_split_stats_tiles(c0, c1, n, edge, cout)
    stats = torch.empty((n, cout, tiles, 2), dtype=torch.float64, device=out.device) if USE_FUSED_STATS else None
    timed = conv_event_filter is not None and conv_event_filter(c0 + c1, cout, edge, n)
    if timed:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    _lib.check(lib.rf_conv3d_up_split_k3_gn_relu(_p(src0), c0, _p(src1), c1, n, edge, _p(aff), _p(w_split_packed), cout, _p(out),
                                                 _p(stats), _stream()), 'rf_conv3d_up_split_k3_gn_relu')
    if timed:
        ev1.record()
        conv_events.append((ev0, ev1, conv_up_split_issued_flops(c0, c1, n, edge, cout), ('rf_conv3d_up_split_k3_gn_relu', 'f16 split', (c0, c1, n, edge, cout))))
    if stats is not None:
        out._rf_stats = (stats, tiles, out._version)
    return out


USE_CH8 = True                  # False: the final decoder's conv pair hands over an NCDHW tensor (the round-3 route; kept for cross-checks)


def conv_up_split_ch8_supported(src1, cout, next_cout):
    """True when the final decoder's pair can hand over channel-interleaved: the box form of the decoder-form split conv writes ch8
    (rf_conv3d_up_split_k3_gn_relu_ch8) and the persistent z-column form of the second conv + pointwise head reads it."""
    if not USE_CH8 or not USE_FUSED_STATS or not USE_CONV_UP or CONV_ARITH != 'split' or src1 is None:
        return False
    n, c1, edge = src1.shape[0], src1.shape[1], 2 * src1.shape[2]
    lib = _lib.load()
    return bool(lib.rf_conv3d_up_split_ch8_supported(0, c1, n, edge, cout)) and bool(lib.rf_conv3d_split_pointwise_ch8_supported(cout, n, edge, next_cout))


def conv3d_up_split_gn_relu_ch8(src1, aff, w_split_packed, cout):
    """relu(conv3(GN(up2(src1)))) written channel-interleaved: -> (out [n, cout / 8, e, e, e, 8] fp32, stats [n, cout, tiles, 2] float64, tiles)"""
    _req(src1, 'src1')
    n, c1, edge = src1.shape[0], src1.shape[1], 2 * src1.shape[2]
    dev = _check_affine(aff, n, c1)
    lib = _lib.load()
    out = torch.empty((n, cout // 8, edge, edge, edge, 8), dtype=torch.float32, device=dev)
    tiles = lib.rf_conv3d_up_split_stats_tiles(0, c1, n, edge, cout)
    stats = torch.empty((n, cout, tiles, 2), dtype=torch.float64, device=dev)
    _lib.check(lib.rf_conv3d_up_split_k3_gn_relu_ch8(_p(None), 0, _p(src1), c1, n, edge, _p(aff), _p(w_split_packed), cout, _p(out), _p(stats), _stream()),
               'rf_conv3d_up_split_k3_gn_relu_ch8')
    return out, stats, tiles


def gn_affine_from_stats(stats, tiles, n, c, edge, gamma, beta, groups, eps=1e-5):
    """the GroupNorm triples of a tensor known only through its producer's per-tile sums (a ch8 tensor: gn_affine reads shapes as NCDHW)"""
    if c < groups:
        groups = 1
    aff = torch.empty((n, c, 4), dtype=torch.float32, device=stats.device)
    _lib.check(_lib.load().rf_gn_from_stats(_p(stats), c, tiles, _p(None), 0, 0, n, edge, _p(gamma.detach()), _p(beta.detach()), groups, eps, _p(aff), _stream()), 'rf_gn_from_stats')
    return aff


def conv3d_split_pointwise_tanh_ch8(x_ch8, gn_affine_t, w_split_packed, cout, pw_w, pw_b, post_add=0.0, post_mul=1.0):
    """conv3d_split_pointwise_tanh on a channel-interleaved input [n, cin / 8, e, e, e, 8]"""
    _req(x_ch8, 'x_ch8')
    n, cin, edge = x_ch8.shape[0], x_ch8.shape[1] * 8, x_ch8.shape[2]
    _check_affine(gn_affine_t, n, cin)
    out = torch.empty((n, 1, edge, edge, edge), dtype=torch.float32, device=x_ch8.device)
    _lib.check(_lib.load().rf_conv3d_split_k3_gn_relu_pointwise_tanh_ch8(_p(x_ch8), cin, n, edge, _p(gn_affine_t), _p(w_split_packed), cout, _p(pw_w.detach()), _p(pw_b.detach()),
                                                                         post_add, post_mul, _p(out), _stream()), 'rf_conv3d_split_k3_gn_relu_pointwise_tanh_ch8')
    return out


def conv_up_split_issued_flops(c0, c1, n, edge, cout):
    """f16 multiply-adds (x2 = flop) rf_conv3d_up_split_k3_gn_relu ISSUES: three MFMAs per k-step of 32, k-steps = 7 per 8 skip
    channels (28 tap slots for 27 taps) + 2 per 8 upsampled channels, on round_up(cout, 16) columns, minus the z-border MFMAs the
    whole-sample kernel leaves out."""
    ksteps = (c0 // 8) * 7 + (c1 // 8) * 2
    if edge == 8 and _lib.load().rf_conv3d_up_split_stats_tiles(c0, c1, n, edge, cout) == 1:
        # whole 8^3 samples (k_conv3_up_split): the z-border taps of the first / last output plane are not issued -- 2 of the 28 (m-block, k-step)
        # pairs of a skip chunk, 1 of the 8 of an upsampled chunk
        ksteps = (c0 // 8) * 7 * 26.0 / 28 + (c1 // 8) * 2 * 7.0 / 8
    return 2.0 * 3 * ksteps * 32 * (-(-cout // 16) * 16) * edge ** 3 * n


def conv_up_issued_flops(c0, c1, n, edge, cout):
    """multiply-adds rf_conv3d_up_k3_gn_relu ISSUES (x2 = flop): decoder form 27*c0 + 8*c1 per (voxel, cout) minus the
    zero-padding taps it leaves out.  The position-major 4^3 tiling leaves out all of them (valid fraction (10/12)^3 of the 27
    taps and (1.5/2)^3 of the 8 low-res taps); the parity-split boxes only the z-border ones (edge 4: 1/6 and 1/4; edge >= 8:
    1/12 and 1/8 in the first / last box along z)."""
    variant = _lib.load().rf_conv3d_up_variant(c0, c1, n, edge, cout)
    if variant == 1:
        fa, fb = 1 - (10.0 / 12) ** 3, 1 - (1.5 / 2) ** 3
    else:
        fa, fb = (1.0 / 6, 1.0 / 4) if edge == 4 else (1.0 / (12 * (edge // 8)), 1.0 / (8 * (edge // 8)))
    return 2.0 * (27 * c0 * (1 - fa) + 8 * c1 * (1 - fb)) * cout * edge ** 3 * n


def maxpool2(x):
    _req(x, 'x')
    n, c, edge = x.shape[0], x.shape[1], x.shape[2]
    out = torch.empty((n, c, edge // 2, edge // 2, edge // 2), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    if USE_FUSED_STATS:
        tiles = lib.rf_maxpool_stats_tiles(edge)
        stats = torch.empty((n, c, tiles, 2), dtype=torch.float64, device=x.device)
        _lib.check(lib.rf_maxpool3d_2_stats(_p(x), n, c, edge, _p(out), _p(stats), _stream()), 'rf_maxpool3d_2_stats')
        out._rf_stats = (stats, tiles, out._version)
    else:
        _lib.check(lib.rf_maxpool3d_2(_p(x), n, c, edge, _p(out), _stream()), 'rf_maxpool3d_2')
    return out


def conv1x1_tanh(x, w, b, post_add=0.0, post_mul=1.0):
    _req(x, 'x'), _req(w.detach(), 'w'), _req(b.detach(), 'b')
    if w.shape[0] != 1:
        raise NotImplementedError('conv1x1_tanh: one output channel (Conv3d(nf,1,1), model/refinement.py:54)')
    n, c = x.shape[0], x.shape[1]
    vox = x[0, 0].numel()
    out = torch.empty((n, 1) + tuple(x.shape[2:]), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().rf_conv1x1_tanh(_p(x), n, c, vox, _p(w.detach()), _p(b.detach()), post_add, post_mul, _p(out), _stream()),
               'rf_conv1x1_tanh')
    return out


def pack_convv_weight(w):
    _req(w.detach(), 'conv weight')
    cout, cin, k = w.shape[0], w.shape[1], w.shape[2]
    lib = _lib.load()
    out = torch.empty(lib.rf_convv_packed_floats(cout, cin, k), dtype=torch.float32, device=w.device)
    _lib.check(lib.rf_convv_pack_weight(_p(w.detach()), cout, cin, k, _p(out), _stream()), 'rf_convv_pack_weight')
    return out


def conv3d_valid_leaky_mfma(x, w_packed, bias, cout, k, stride, slope):
    """valid strided conv + bias + LeakyReLU on the matrix cores (w_packed from pack_convv_weight)."""
    _req(x, 'x')
    n, cin, s = x.shape[0], x.shape[1], x.shape[2]
    so = (s - k) // stride + 1
    out = torch.empty((n, cout, so, so, so), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().rf_conv3d_valid_leaky_mfma(_p(x), n, cin, s, _p(w_packed), _p(bias.detach() if bias is not None else None), cout, k,
                                                      stride, slope, _p(out), _stream()), 'rf_conv3d_valid_leaky_mfma')
    return out


def pack_convv_lds_weight(w):
    _req(w.detach(), 'conv weight')
    cout, cin, k = w.shape[0], w.shape[1], w.shape[2]
    lib = _lib.load()
    out = torch.empty(lib.rf_convv_lds_packed_floats(cout, cin, k), dtype=torch.float32, device=w.device)
    _lib.check(lib.rf_convv_lds_pack_weight(_p(w.detach()), cout, cin, k, _p(out), _stream()), 'rf_convv_lds_pack_weight')
    return out


USE_CONVV_LDS = True            # False: every valid-conv layer runs the gather form


def conv_valid_lds_supported(x, cout, k, stride):
    """True when the LDS-staged form (rf_conv3d_valid_leaky_lds) takes this layer (output edge >= 8 and a tile that fits LDS)."""
    return USE_CONVV_LDS and bool(_lib.load().rf_conv3d_valid_lds_supported(x.shape[0], x.shape[1], x.shape[2], cout, k, stride))


def conv3d_valid_leaky_lds(x, w_packed, bias, cout, k, stride, slope):
    """valid strided conv + bias + LeakyReLU, LDS-staged fp32-MFMA form (w_packed from pack_convv_lds_weight)."""
    _req(x, 'x')
    n, cin, s = x.shape[0], x.shape[1], x.shape[2]
    so = (s - k) // stride + 1
    out = torch.empty((n, cout, so, so, so), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().rf_conv3d_valid_leaky_lds(_p(x), n, cin, s, _p(w_packed), _p(bias.detach() if bias is not None else None), cout, k,
                                                     stride, slope, _p(out), _stream()), 'rf_conv3d_valid_leaky_lds')
    return out


def pack_convv_split_weight(w, s, stride):
    """f16 fragment image of the split-operand valid conv for input edge s (the tile / chunk plan depends on the layer's input size)"""
    _req(w.detach(), 'conv weight')
    cout, cin, k = w.shape[0], w.shape[1], w.shape[2]
    lib = _lib.load()
    nbytes = lib.rf_convv_split_packed_bytes(cout, cin, k, s, stride)
    if nbytes == 0:
        raise ValueError('pack_convv_split_weight: layer %s @%d^3 stride %d is not taken by the split form' % (tuple(w.shape), s, stride))
    out = torch.empty(nbytes // 2, dtype=torch.float16, device=w.device)
    _lib.check(lib.rf_convv_split_pack_weight(_p(w.detach()), cout, cin, k, s, stride, _p(out), _stream()), 'rf_convv_split_pack_weight')
    return out


def conv_valid_split_supported(x, cout, k, stride):
    """True when the split-operand F16-MFMA form (rf_conv3d_valid_leaky_split) takes this layer and CONV_ARITH selects it."""
    return CONV_ARITH == 'split' and bool(_lib.load().rf_conv3d_valid_split_supported(x.shape[0], x.shape[1], x.shape[2], cout, k, stride))


class SplitActs:
    """Activations between two valid-conv layers in split form (include/rfuse.h, rf_conv3d_valid_leaky_split_ex): ``data`` has the shape and byte size
    of the fp32 tensor [n, c, s, s, s] it stands for, but holds [n][c/4][h | l][s^3] 8-byte slots.  Only the valid-conv kernels read it."""

    def __init__(self, data):
        self.data = data
        self.shape = data.shape
        self.device = data.device


USE_SPLIT_CHAIN = True          # False: every valid-conv layer writes fp32 and the next one converts what it stages


def conv3d_valid_leaky_split(x, w_packed, bias, cout, k, stride, slope, out_split=False):
    """valid strided conv + bias + LeakyReLU on the F16 matrix cores by operand splitting (w_packed from pack_convv_split_weight).  ``x`` may be a
    SplitActs (the previous layer wrote split form); ``out_split``: return a SplitActs for the next layer."""
    in_split = isinstance(x, SplitActs)
    xt = x.data if in_split else x
    _req(xt, 'x')
    n, cin, s = xt.shape[0], xt.shape[1], xt.shape[2]
    so = (s - k) // stride + 1
    out = torch.empty((n, cout, so, so, so), dtype=torch.float32, device=xt.device)
    _lib.check(_lib.load().rf_conv3d_valid_leaky_split_ex(_p(xt), int(in_split), n, cin, s, _p(w_packed), _p(bias.detach() if bias is not None else None), cout, k,
                                                          stride, slope, _p(out), int(out_split), _stream()), 'rf_conv3d_valid_leaky_split')
    return SplitActs(out) if out_split else out


USE_CONVV_PG = True             # False: grid-sized split-form layers stay on the tile-per-workgroup kernel


def conv_valid_split_pg_supported(shape, cout, k, stride):
    """True when the persistent grid form (rf_conv3d_valid_leaky_split_pg) takes a split-form input of shape (n, cin, s)"""
    return USE_CONVV_PG and CONV_ARITH == 'split' and bool(_lib.load().rf_conv3d_valid_split_pg_supported(max(shape[0], 1), shape[1], shape[2], cout, k, stride))


def pack_convv_split_pg_weight(w, s, stride):
    """weight image (tables + f16 fragments) of the persistent grid form for input edge s"""
    _req(w.detach(), 'conv weight')
    cout, cin, k = w.shape[0], w.shape[1], w.shape[2]
    lib = _lib.load()
    nbytes = lib.rf_convv_split_pg_packed_bytes(cout, cin, k, s, stride)
    if nbytes == 0:
        raise ValueError('pack_convv_split_pg_weight: layer %s @%d^3 stride %d is not taken by the persistent grid form' % (tuple(w.shape), s, stride))
    out = torch.empty(nbytes // 2, dtype=torch.float16, device=w.device)
    _lib.check(lib.rf_convv_split_pg_pack_weight(_p(w.detach()), cout, cin, k, s, stride, _p(out), _stream()), 'rf_convv_split_pg_pack_weight')
    return out


def conv3d_valid_leaky_split_pg(x, w_packed, bias, cout, k, stride, slope):
    """SplitActs -> SplitActs: valid conv + bias + LeakyReLU, persistent grid form (w_packed from pack_convv_split_pg_weight); equal to
    conv3d_valid_leaky_split(x, ..., out_split=True) within the last bits of an fp32 sum (another MFMA shape)"""
    if not isinstance(x, SplitActs):
        raise TypeError('conv3d_valid_leaky_split_pg: the input must be in split form (SplitActs)')
    xt = x.data
    _req(xt, 'x')
    n, cin, s = xt.shape[0], xt.shape[1], xt.shape[2]
    so = (s - k) // stride + 1
    out = torch.empty((n, cout, so, so, so), dtype=torch.float32, device=xt.device)
    _lib.check(_lib.load().rf_conv3d_valid_leaky_split_pg(_p(xt), n, cin, s, _p(w_packed), _p(bias.detach() if bias is not None else None), cout, k, stride, slope,
                                                          _p(out), _stream()), 'rf_conv3d_valid_leaky_split_pg')
    return SplitActs(out)


USE_CONVV_VALU = True           # False: the first layers of the patch encoders stay on the matrix cores


def conv_valid_valu_supported(x, cout, k, stride):
    """True when the packed-fp32 VALU form (rf_conv3d_valid_leaky_valu) takes this layer (first layers of the patch encoders)."""
    return USE_CONVV_VALU and bool(_lib.load().rf_conv3d_valid_valu_supported(x.shape[0], x.shape[1], x.shape[2], cout, k, stride))


def pack_convv_valu_weight(w):
    """OIDHW -> [cin][k^3][cout] (a permute: the VALU form reads whole cout vectors of a (channel, tap) through the scalar cache)"""
    _req(w.detach(), 'conv weight')
    return w.detach().permute(1, 2, 3, 4, 0).contiguous()


def conv3d_valid_leaky_valu(x, w_t, bias, stride, slope, out_split=False):
    """valid stride-1 conv + bias + LeakyReLU on the vector unit (w_t from pack_convv_valu_weight: [cin, k, k, k, cout]); ``out_split``: the output
    in split form (a SplitActs) for a split-operand layer behind it."""
    _req(x, 'x'), _req(w_t, 'w_t')
    n, cin, s = x.shape[0], x.shape[1], x.shape[2]
    cout, k = w_t.shape[4], w_t.shape[1]
    so = (s - k) // stride + 1
    out = torch.empty((n, cout, so, so, so), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().rf_conv3d_valid_leaky_valu_ex(_p(x), n, cin, s, _p(w_t), _p(bias.detach() if bias is not None else None), cout, k,
                                                         stride, slope, _p(out), int(out_split), _stream()), 'rf_conv3d_valid_leaky_valu')
    return SplitActs(out) if out_split else out


def conv3d_valid_leaky(x, w, bias, stride, slope):
    _req(x, 'x'), _req(w.detach(), 'w')
    n, cin, s = x.shape[0], x.shape[1], x.shape[2]
    cout, k = w.shape[0], w.shape[2]
    so = (s - k) // stride + 1
    out = torch.empty((n, cout, so, so, so), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().rf_conv3d_valid_leaky(_p(x), n, cin, s, _p(w.detach()), _p(bias.detach() if bias is not None else None), cout, k,
                                                 stride, slope, _p(out), _stream()), 'rf_conv3d_valid_leaky')
    return out


# --------------------------------------------------------------------------------------------------- fold / unfold

def unfold3d(x, e):
    _req(x, 'x')
    b, c, s = x.shape[0], x.shape[1], x.shape[2]
    r = s // e
    rows = torch.empty((b * r * r * r, c, e, e, e), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().rf_unfold3d(_p(x), b, c, s, e, _p(rows), _stream()), 'rf_unfold3d')
    return rows


def fold3d(rows, r, e, c):
    _req(rows, 'rows')
    s = r * e
    total = rows.numel()
    b = total // (c * s * s * s)
    if b * c * s * s * s != total:
        raise ValueError('fold3d: %d values do not tile [b,%d,%d^3]' % (total, c, s))
    x = torch.empty((b, c, s, s, s), dtype=torch.float32, device=rows.device)
    _lib.check(_lib.load().rf_fold3d(_p(rows), b, c, s, e, _p(x), _stream()), 'rf_fold3d')
    return x


# ------------------------------------------------------------------------------------------------------ Linear/MLP

def pack_linear_weight(w):
    _req(w.detach(), 'linear weight')
    nout, nin = w.shape
    lib = _lib.load()
    out = torch.empty(lib.rf_linear_packed_floats(nout, nin), dtype=torch.float32, device=w.device)
    _lib.check(lib.rf_linear_pack_weight(_p(w.detach()), nout, nin, _p(out), _stream()), 'rf_linear_pack_weight')
    return out


def linear(x, w_packed, bias, nout, act=ACT_NONE, slope=0.0):
    _req(x, 'x')
    rows, nin = x.shape
    y = torch.empty((rows, nout), dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().rf_linear(_p(x), rows, nin, _p(w_packed), _p(bias.detach() if bias is not None else None), nout, act, slope,
                                     _p(y), _stream()), 'rf_linear')
    return y


def linear_wgrad(a, b):
    """a [K, M], b [K, N] -> a^T b [M, N] (split-K MFMA GEMM, float64 slice sum): the weight gradient of a Linear layer"""
    _req(a, 'a'), _req(b, 'b')
    if a.shape[0] != b.shape[0]:
        raise ValueError('linear_wgrad: row counts differ (%d vs %d)' % (a.shape[0], b.shape[0]))
    k, m, n = a.shape[0], a.shape[1], b.shape[1]
    lib = _lib.load()
    nbytes = lib.rf_linear_wgrad_ws_bytes(k, m, n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=a.device)
    dw = torch.empty((m, n), dtype=torch.float32, device=a.device)
    _lib.check(lib.rf_linear_wgrad(_p(a), _p(b), k, m, n, _p(dw), _p(ws), nbytes, _stream()), 'rf_linear_wgrad')
    return dw


def l2_normalize_rows_(x, eps=1e-12):
    _req(x, 'x')
    _lib.check(_lib.load().rf_l2_normalize_rows(_p(x), x.shape[0], x.shape[1], eps, _stream()), 'rf_l2_normalize_rows')
    return x


# ------------------------------------------------------------------------------------------------------- attention

def attn_gather_retrieved(src, layout, b, k, c, s, e, t=0):
    _req(src, 'src')
    r = s // e
    p = torch.empty((b * r * r * r, k, c, e, e, e), dtype=torch.float32, device=src.device)
    _lib.check(_lib.load().rf_attn_gather_retrieved(_p(src), layout, b, k, c, s, e, t, _p(p), _stream()), 'rf_attn_gather_retrieved')
    return p


def attn_fuse(x, p, xf, pf, noise, mode, sharpness, debug=False):
    """x [b,d]-like, p [b,K,d]-like, xf [b,f], pf [b*K,f] (raw encoder outputs), noise [b,K] or None."""
    _req(x, 'x'), _req(p, 'p'), _req(xf, 'xf'), _req(pf, 'pf')
    if noise is not None:
        _req(noise, 'noise')
    b = x.shape[0]
    k = p.shape[1]
    d = x[0].numel()
    f = xf.shape[1]
    out = torch.empty_like(x)
    sc = torch.empty((b, k), dtype=torch.float32, device=x.device) if debug else None
    wt = torch.empty((b, k), dtype=torch.float32, device=x.device) if debug else None
    _lib.check(_lib.load().rf_attn_fuse(_p(x), _p(p), _p(xf), _p(pf), _p(noise), b, k, d, f, mode, sharpness, _p(out), _p(sc), _p(wt),
                                        _stream()), 'rf_attn_fuse')
    return (out, sc, wt) if debug else out


USE_FUSED_ATTN_MLP = True       # False: per-layer rf_linear + the row-domain attention kernels (kept for cross-checks)


class PackedAttnMLP:
    """MFMA operand image of one AttentionFeatureEncoder (4 Linear layers); re-packed when any parameter changes."""

    def __init__(self):
        self._key = None
        self._packed = None
        self._ready = _PackedReady()

    def get(self, layers):
        params = [t for l in layers for t in (l.weight, l.bias)]
        key = tuple((t.data_ptr(), t._version, tuple(t.shape), t.device) for t in params)
        if key != self._key:
            img, split = pack_attn_mlp(params)
            if not all(split_range_ok(w) for w in params[0::2]):
                split = None                                           # a weight outside the f16 pair's range: the fp32-MFMA form runs (no clamp)
            self._packed, self._key = (img, split), key
            self._ready.packed_on(params[0].device)
        else:
            self._ready.wait(params[0].device, self._packed)
        return self._packed


def pack_attn_mlp(params):
    """[w0, b0, w1, b1, w2, b2, w3, b3] of Linear(n_in,128) x (128,128) x (128,128) x (128,32) -> the fused kernel's operand images:
    (fp32 MFMA image incl. biases, split-operand f16 image)."""
    for t in params:
        _req(t.detach(), 'attention MLP parameter')
    n_in = params[0].shape[1]
    shapes = [tuple(t.shape) for t in params[0::2]]
    if shapes != [(128, n_in), (128, 128), (128, 128), (32, 128)] or n_in % 16 or not 16 <= n_in <= 128:
        raise ValueError('fused attention MLP needs Linear(n_in,128) x (128,128) x (128,128) x (128,32) with n_in a multiple of 16 <= 128, '
                         'got %s' % (shapes,))
    lib = _lib.load()
    out = torch.empty(lib.rf_attn_mlp_packed_floats(n_in), dtype=torch.float32, device=params[0].device)
    _lib.check(lib.rf_attn_mlp_pack(*[_p(t.detach()) for t in params], n_in, _p(out), _stream()), 'rf_attn_mlp_pack')
    split = torch.empty(lib.rf_attn_mlp_split_packed_floats(n_in), dtype=torch.float32, device=params[0].device)
    _lib.check(lib.rf_attn_mlp_split_pack(*[_p(t.detach()) for t in params[0::2]], n_in, _p(split), _stream()), 'rf_attn_mlp_split_pack')
    return out, split


def attn_mlp_rows(x, packed):
    """x [rows, n_in] -> [rows, 32] through the fused 4-layer encoder (split-operand form unless CONV_ARITH == 'fp32')."""
    _req(x, 'x')
    rows, n_in = x.shape
    out = torch.empty((rows, 32), dtype=torch.float32, device=x.device)
    img, split = packed
    if CONV_ARITH == 'split' and split is not None:
        _lib.check(_lib.load().rf_attn_mlp_split_rows(_p(x), rows, n_in, _p(img), _p(split), _p(out), _stream()), 'rf_attn_mlp_split_rows')
    else:
        _lib.check(_lib.load().rf_attn_mlp_rows(_p(x), rows, n_in, _p(img), _p(out), _stream()), 'rf_attn_mlp_rows')
    return out


def attn_mlp_volume(src, b, kv, c, s, t, packed):
    """Encoder over every 2^3 attention patch of b*kv feature volumes (patch-major [(b*kv*q^3), c, t,t,t], or NCDHW when t == s)
    -> [(b*r^3*kv), 32], rows ordered (b, prow, k)."""
    _req(src, 'src')
    if src.numel() != b * kv * c * s * s * s:
        raise ValueError('attn_mlp_volume: %d values are not %d volumes of [%d,%d^3]' % (src.numel(), b * kv, c, s))
    r = s // 2
    out = torch.empty((b * r * r * r * kv, 32), dtype=torch.float32, device=src.device)
    img, split = packed
    if CONV_ARITH == 'split' and split is not None:
        _lib.check(_lib.load().rf_attn_mlp_split_volume(_p(src), b, kv, c, s, t, _p(img), _p(split), _p(out), _stream()), 'rf_attn_mlp_split_volume')
    else:
        _lib.check(_lib.load().rf_attn_mlp_volume(_p(src), b, kv, c, s, t, _p(img), _p(out), _stream()), 'rf_attn_mlp_volume')
    return out


def attn_weights(xf, pf, noise, k, mode, sharpness, debug=False):
    """xf [rows,f], pf [rows*k,f] raw encoder outputs -> (weights [rows,k], switches [rows][, scores [rows,k]])."""
    _req(xf, 'xf'), _req(pf, 'pf')
    if noise is not None:
        _req(noise, 'noise')
    rows, f = xf.shape
    if pf.shape[0] != rows * k:
        raise ValueError('attn_weights: %d phi rows for %d theta rows and K=%d' % (pf.shape[0], rows, k))
    w = torch.empty((rows, k), dtype=torch.float32, device=xf.device)
    sw = torch.empty((rows,), dtype=torch.float32, device=xf.device)
    sc = torch.empty((rows, k), dtype=torch.float32, device=xf.device) if debug else None
    _lib.check(_lib.load().rf_attn_weights(_p(xf), _p(pf), _p(noise), rows, k, f, mode, sharpness, _p(w), _p(sw), _p(sc), _stream()),
               'rf_attn_weights')
    return (w, sw, sc) if debug else (w, sw)


def gumbel_rng_state(device, seed=None):
    """{seed, offset, 0} as int64[3] on ``device`` for attn_weights_sampled; the seed defaults to torch's CUDA seed of that device."""
    if seed is None:
        seed = torch.cuda.initial_seed() if torch.device(device).type == 'cuda' else torch.initial_seed()
    return torch.tensor([seed & 0x7FFFFFFFFFFFFFFF, 0, 0], dtype=torch.int64, device=device)


def attn_weights_sampled(xf, pf, k, sharpness, rng_state, want_noise=False):
    """Gumbel-hard weights with the noise drawn inside the kernel (Philox; advances ``rng_state``) -> (weights, switches[, noise used])."""
    _req(xf, 'xf'), _req(pf, 'pf'), _req(rng_state, 'rng_state', torch.int64)
    rows, f = xf.shape
    if pf.shape[0] != rows * k:
        raise ValueError('attn_weights_sampled: %d phi rows for %d theta rows and K=%d' % (pf.shape[0], rows, k))
    w = torch.empty((rows, k), dtype=torch.float32, device=xf.device)
    sw = torch.empty((rows,), dtype=torch.float32, device=xf.device)
    nz = torch.empty((rows, k), dtype=torch.float32, device=xf.device) if want_noise else None
    _lib.check(_lib.load().rf_attn_weights_sampled(_p(xf), _p(pf), rows, k, f, sharpness, _p(rng_state), _p(w), _p(sw), _p(None), _p(nz), _stream()),
               'rf_attn_weights_sampled')
    return (w, sw, nz) if want_noise else (w, sw)


def attn_blend(x, retrieved, k, t, weights, switches):
    """x [b,c,s,s,s]; retrieved features of b*k volumes (patch-major with patch edge t, or NCDHW when t == s) -> [b,c,s,s,s]."""
    _req(x, 'x'), _req(retrieved, 'retrieved'), _req(weights, 'weights'), _req(switches, 'switches')
    b, c, s = x.shape[0], x.shape[1], x.shape[2]
    if retrieved.numel() != b * k * c * s * s * s:
        raise ValueError('attn_blend: retrieved features do not match %d x %d volumes' % (b, k))
    out = torch.empty_like(x)
    _lib.check(_lib.load().rf_attn_blend(_p(x), _p(retrieved), b, k, c, s, t, _p(weights), _p(switches), _p(out), _stream()), 'rf_attn_blend')
    return out


# ------------------------------------------------------------------------------------------------------- retrieval

def query_windows(raw, ps, ctx, pad_value, mean, std):
    _req(raw, 'raw')
    b, s = raw.shape[0], raw.shape[-1]
    npatch, w = s // ps, ps + 2 * ctx
    out = torch.empty((b * npatch ** 3, 1, w, w, w), dtype=torch.float32, device=raw.device)
    _lib.check(_lib.load().rf_query_windows(_p(raw), b, s, ps, ctx, pad_value, mean, std, _p(out), _stream()), 'rf_query_windows')
    return out


def gather_windows(grid, w, step, npatch):
    """grid [n,c,g,g,g] (or a SplitActs of that shape) -> the npatch^3 windows of edge w at stride step of every sample: [n*npatch^3, c, w, w, w]
    (rf_query_windows order), in the form of the input"""
    split = isinstance(grid, SplitActs)
    gt = grid.data if split else grid
    _req(gt, 'grid')
    n, c, g = gt.shape[0], gt.shape[1], gt.shape[2]
    out = torch.empty((n * npatch ** 3, c, w, w, w), dtype=torch.float32, device=gt.device)
    if split:
        _lib.check(_lib.load().rf_gather_windows_split(_p(gt), n, c, g, w, step, npatch, _p(out), _stream()), 'rf_gather_windows_split')
        return SplitActs(out)
    _lib.check(_lib.load().rf_gather_windows(_p(gt), n, c, g, w, step, npatch, _p(out), _stream()), 'rf_gather_windows')
    return out


USE_FCN_ENCODER = True          # False: the conv patch encoders run on every window separately (the reference's form)


def embed_windows(encoder, raw, ps, ctx, pad_value, mean, std):
    """The embeddings of the (s/ps)^3 windows (edge ps + 2 ctx, stride ps, padded with pad_value, normalised) of raw chunks [b,s,s,s], in
    rf_query_windows order: [b * (s/ps)^3, latent, 1,1,1].  A conv patch encoder whose leading layers pay on the whole padded chunk
    (model/retrieval.py grid_plan) gets the chunk as ONE window with context -- the same padded, normalised voxels -- and cuts the windows
    out of its feature grid; everything else gets the windows."""
    s = raw.shape[-1]
    if USE_FCN_ENCODER and hasattr(encoder, 'grid_plan') and not getattr(encoder, 'BATCHNORM', False) and encoder.grid_plan(ps + 2 * ctx, ps, s // ps)[0] > 0:
        grid = query_windows(raw, s, ctx, pad_value, mean, std)
        return encoder.forward_grid(grid, ps + 2 * ctx, ps)
    return encoder(query_windows(raw, ps, ctx, pad_value, mean, std))


def db_pack_embeddings(emb):
    _req(emb, 'emb')
    n, dim = emb.shape
    lib = _lib.load()
    out = torch.empty(lib.rf_db_packed_floats(n, dim), dtype=torch.float32, device=emb.device)
    _lib.check(lib.rf_db_pack_embeddings(_p(emb), n, dim, _p(out), _stream()), 'rf_db_pack_embeddings')
    return out


TOPK_AUTO, TOPK_VALU_SCAN, TOPK_MFMA_SCAN, TOPK_MFMA16_SCAN = 0, 1, 2, 3      # 2: fp32-MFMA filter, 3: f16-MFMA filter (what AUTO takes for big shards)


def l2_topk(q, db_packed, n, row_base, k2, algo=TOPK_AUTO):
    """exact squared-L2 top-k2 -> (dist [nq,k2] f32, idx [nq,k2] i64 global row ids); both scans return the same bits"""
    _req(q, 'q'), _req(db_packed, 'db_packed')
    nq, dim = q.shape
    lib = _lib.load()
    dist = torch.empty((nq, k2), dtype=torch.float32, device=q.device)
    idx = torch.empty((nq, k2), dtype=torch.int64, device=q.device)
    nbytes = lib.rf_l2_topk_ws_bytes(nq, n, k2)
    ws = _workspace(q.device, nbytes)
    _lib.check(lib.rf_l2_topk(_p(q), nq, dim, _p(db_packed), n, row_base, k2, algo, _p(dist), _p(idx), _p(ws), ws.numel(), _stream()), 'rf_l2_topk')
    return dist, idx


def l2_topk_keys(q, db_packed, n, row_base, k2, algo=TOPK_AUTO):
    """the same search as packed keys [nq,k2] int64 (bit pattern: dist bits << 32 | global row id; -1 = no candidate)"""
    _req(q, 'q'), _req(db_packed, 'db_packed')
    nq, dim = q.shape
    lib = _lib.load()
    keys = torch.empty((nq, k2), dtype=torch.int64, device=q.device)
    nbytes = lib.rf_l2_topk_ws_bytes(nq, n, k2)
    ws = _workspace(q.device, nbytes)
    _lib.check(lib.rf_l2_topk_keys(_p(q), nq, dim, _p(db_packed), n, row_base, k2, algo, _p(keys), _p(ws), ws.numel(), _stream()), 'rf_l2_topk_keys')
    return keys


def topk_merge_keys(key_parts):
    """[parts, nq, k2] int64 packed keys -> (dist [nq,k2], idx [nq,k2]) best by (dist, row id)"""
    _req(key_parts, 'key_parts', torch.int64)
    parts, nq, k2 = key_parts.shape
    dist = torch.empty((nq, k2), dtype=torch.float32, device=key_parts.device)
    idx = torch.empty((nq, k2), dtype=torch.int64, device=key_parts.device)
    _lib.check(_lib.load().rf_topk_merge_keys(_p(key_parts), parts, nq, k2, _p(dist), _p(idx), _stream()), 'rf_topk_merge_keys')
    return dist, idx


def topk_merge(dist_parts, idx_parts):
    """[parts, nq, k2] candidate lists -> [nq, k2] best by (dist, idx)."""
    _req(dist_parts, 'dist_parts'), _req(idx_parts, 'idx_parts', torch.int64)
    parts, nq, k2 = dist_parts.shape
    dist = torch.empty((nq, k2), dtype=torch.float32, device=dist_parts.device)
    idx = torch.empty((nq, k2), dtype=torch.int64, device=dist_parts.device)
    _lib.check(_lib.load().rf_topk_merge(_p(dist_parts), _p(idx_parts), parts, nq, k2, _p(dist), _p(idx), _stream()), 'rf_topk_merge')
    return dist, idx


def demote_same_scene(dist, idx, db_meta, query_scene, K, query_keep=None):
    """query_keep [nq] bool/uint8: False = patch dropped by the occupancy filter -> K 'no neighbour' entries."""
    _req(dist, 'dist'), _req(idx, 'idx', torch.int64), _req(db_meta, 'db_meta', torch.int32)
    if query_scene is not None:
        _req(query_scene, 'query_scene', torch.int32)
    if query_keep is not None:
        if query_keep.dtype == torch.bool:
            query_keep = query_keep.view(torch.uint8)
        _req(query_keep, 'query_keep', torch.uint8)
        if query_keep.numel() != dist.shape[0]:
            raise ValueError('query_keep: %d flags for %d queries' % (query_keep.numel(), dist.shape[0]))
    nq, k2 = dist.shape
    dev = dist.device
    out_meta = torch.empty((nq, K, 7), dtype=torch.int32, device=dev)
    out_dist = torch.empty((nq, K), dtype=torch.float32, device=dev)
    out_idx = torch.empty((nq, K), dtype=torch.int64, device=dev)
    _lib.check(_lib.load().rf_demote_same_scene(_p(dist), _p(idx), nq, k2, _p(db_meta), _p(query_scene), _p(query_keep), K, _p(out_meta),
                                                _p(out_dist), _p(out_idx), _stream()), 'rf_demote_same_scene')
    return out_meta, out_dist, out_idx


def gather_rows(src, idx):
    """out[m] = src[idx[m]] for a row-major float32 ``src`` [R, ...] and int64 ``idx`` [M]."""
    _req(src, 'src'), _req(idx, 'idx', torch.int64)
    width = src[0].numel()
    out = torch.empty((idx.numel(),) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device)
    _lib.check(_lib.load().rf_gather_rows(_p(src), src.shape[0], _p(idx), idx.numel(), width, _p(out), _stream()), 'rf_gather_rows')
    return out


def gather_patches(db_volumes, meta, chunks, K, trunc_fill, trunc_ratio, mean, std, layout, no_overlap=True):
    """create_retrieval_from_mapping's copy loop (reference util/retrieval.py:145-164) for patch grids WITHOUT overlap (stride == patch size: every shipped
    config, dataset/patched_scene_dataset.py:113-115), where the branch at :156 always copies.  The overlapping branch (a later patch overwrites its box only
    while the mean stored distance of that box is above its own: an order-dependent reduction) is compose_overlap below, scene by scene."""
    if not no_overlap:
        raise NotImplementedError('gather_patches: the per-chunk gather is the no_overlap branch of create_retrieval_from_mapping (util/retrieval.py:156); an '
                                  'overlapping patch grid is composed scene by scene, in patch order: ops.compose_overlap / formats.compose_scene(no_overlap=False)')
    half = db_volumes.dtype == torch.float16                    # the voxel store in the reference's own precision (PatchDatabase(half_store=True))
    _req(db_volumes, 'db_volumes', torch.float16 if half else torch.float32), _req(meta, 'meta', torch.int32)
    dev = db_volumes.device
    if layout == 1:
        out = torch.empty((chunks * K * 64, 1, 16, 16, 16), dtype=torch.float32, device=dev)
    else:
        out = torch.empty((chunks, K, 64, 64, 64), dtype=torch.float32, device=dev)
    lib = _lib.load()
    fn, name = (lib.rf_gather_patches_f16, 'rf_gather_patches_f16') if half else (lib.rf_gather_patches, 'rf_gather_patches')
    _lib.check(fn(_p(db_volumes), db_volumes.shape[0], _p(meta), chunks, K, trunc_fill, trunc_ratio, mean, std, layout, _p(out), _stream()), name)
    return out


def compose_overlap(db_volumes, mapping, boxes, K, size, trunc_fill, trunc_ratio=1.0):
    """create_retrieval_from_mapping (reference util/retrieval.py:145-164) for ONE scene whose patch grid overlaps (``dataset.no_overlap`` False): ``mapping``
    [P, K, 8] float32 rows (scene index, X0, X1, Y0, Y1, Z0, Z1, distance) of the patches ``patch_from_scene_lookup`` holds, in its order, ``boxes`` [P, 6] int32
    their unpadded target boxes in the scene -> [K, *size] float32.  A patch overwrites its box of retrieval k only while the mean of the distances stored
    there is above its own (:156); the patches are walked in order by one workgroup per k."""
    half = db_volumes.dtype == torch.float16
    _req(db_volumes, 'db_volumes', torch.float16 if half else torch.float32), _req(mapping, 'mapping', torch.float32), _req(boxes, 'boxes', torch.int32)
    P = mapping.shape[0]
    if tuple(mapping.shape) != (P, K, 8) or tuple(boxes.shape) != (P, 6):
        raise ValueError('compose_overlap: mapping must be [P, %d, 8] and boxes [P, 6] (got %s, %s)' % (K, tuple(mapping.shape), tuple(boxes.shape)))
    sx, sy, sz = (int(v) for v in size)
    out = torch.empty((K, sx, sy, sz), dtype=torch.float32, device=db_volumes.device)
    ws = torch.empty((K, sx, sy, sz), dtype=torch.float32, device=db_volumes.device)
    _lib.check(_lib.load().rf_compose_overlap(_p(db_volumes), int(half), db_volumes.shape[0], _p(mapping), _p(boxes), P, K, sx, sy, sz, trunc_fill, trunc_ratio,
                                              _p(out), _p(ws), _stream()), 'rf_compose_overlap')
    return out


def paste_chunks(df, sel, dst, sx, sy, flat, round_half=True):
    """combine_predictions on the device: chunks ``sel`` (int32, indices into the refined batch df [b, 1, 64, 64, 64]) -> float16 rounding -> float64 ->
    ``flat`` (float64) at element offsets ``dst`` (int64) with canvas strides sx, sy (reference dataset/patched_scene_dataset.py:160-186)."""
    _req(df, 'df'), _req(sel, 'sel', torch.int32), _req(dst, 'dst', torch.int64), _req(flat, 'flat', torch.float64)
    if tuple(df.shape[1:]) != (1, 64, 64, 64) or sel.numel() != dst.numel():
        raise ValueError('paste_chunks: df must be [b, 1, 64, 64, 64] and sel / dst of one length (got %s, %d, %d)' % (tuple(df.shape), sel.numel(), dst.numel()))
    _lib.check(_lib.load().rf_paste_chunks(_p(df), df.shape[0], _p(sel), _p(dst), sel.numel(), int(sx), int(sy), 1 if round_half else 0, _p(flat), _stream()),
               'rf_paste_chunks')


# every public tensor op is scoped to its tensors' device (see _device_scoped)
for _name, _fn in list(vars(sys.modules[__name__]).items()):
    if isinstance(_fn, type(_device_scoped)) and _fn.__module__ == __name__ and not _name.startswith('_'):
        setattr(sys.modules[__name__], _name, _device_scoped(_fn))
del _name, _fn
