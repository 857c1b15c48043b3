"""ctypes binding of librfuse_hip.so (the C ABI declared in include/rfuse.h).

There is NO fallback: if the library is missing or a call fails, a RuntimeError is raised.  ``import torch`` must come
first so that the HIP runtime the library binds to is the one PyTorch-ROCm already loaded (same SONAME).
"""
import ctypes
import os
from pathlib import Path

import torch  # noqa: F401  (loads libamdhip64 before we dlopen)

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get('RFUSE_LIB', _HERE / 'librfuse_hip.so'))

c_fp = ctypes.c_void_p     # device float*
c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_f = ctypes.c_float
c_sz = ctypes.c_size_t
c_i64 = ctypes.c_int64

# name -> (restype, argtypes); one entry per symbol declared in include/rfuse.h
SIGNATURES = {
    'rf_abi_version': (c_i, []),
    'rf_last_error': (ctypes.c_char_p, []),
    'rf_conv3_pack_weight': (c_i, [c_fp, c_i, c_i, c_fp, c_p]),
    'rf_conv3_packed_floats': (c_sz, [c_i, c_i]),
    'rf_gn_stats': (c_i, [c_fp, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_f, c_fp, c_p, c_sz, c_p]),
    'rf_gn_stats_ws_bytes': (c_sz, [c_i, c_i]),
    'rf_conv3d_k3_gn_relu': (c_i, [c_fp, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_fp, c_p]),
    'rf_conv3d_k3_gn_relu_stats': (c_i, [c_fp, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_fp, c_p, c_p]),
    'rf_conv3d_stats_tiles': (c_i, [c_i, c_i, c_i, c_i, c_i]),
    'rf_gn_from_stats': (c_i, [c_p, c_i, c_i, c_p, c_i, c_i, c_i, c_i, c_fp, c_fp, c_i, c_f, c_fp, c_p]),
    'rf_maxpool3d_2_stats': (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_p, c_p]),
    'rf_maxpool_stats_tiles': (c_i, [c_i]),
    'rf_conv3d_k3_gn_relu_direct': (c_i, [c_fp, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_fp, c_p]),
    'rf_maxpool3d_2': (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_p]),
    'rf_conv1x1_tanh': (c_i, [c_fp, c_i, c_i, c_sz, c_fp, c_fp, c_f, c_f, c_fp, c_p]),
    'rf_conv3d_valid_leaky': (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_i, c_i, c_f, c_fp, c_p]),
    'rf_conv3d_valid_leaky_mfma': (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_i, c_i, c_f, c_fp, c_p]),
    'rf_convv_pack_weight': (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_p]),
    'rf_convv_packed_floats': (c_sz, [c_i, c_i, c_i]),
    'rf_conv3d_valid_lds_supported': (c_i, [c_i, c_i, c_i, c_i, c_i, c_i]),
    'rf_conv3d_valid_leaky_lds': (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_i, c_i, c_f, c_fp, c_p]),
    'rf_conv3d_valid_valu_supported': (c_i, [c_i, c_i, c_i, c_i, c_i, c_i]),
    'rf_conv3d_valid_leaky_valu': (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_i, c_i, c_f, c_fp, c_p]),
    'rf_conv3d_valid_leaky_valu_ex': (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_i, c_i, c_f, c_p, c_i, c_p]),
    'rf_convv_lds_pack_weight': (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_p]),
    'rf_convv_lds_packed_floats': (c_sz, [c_i, c_i, c_i]),
    'rf_conv3d_valid_split_supported': (c_i, [c_i, c_i, c_i, c_i, c_i, c_i]),
    'rf_conv3d_valid_leaky_split': (c_i, [c_fp, c_i, c_i, c_i, c_p, c_fp, c_i, c_i, c_i, c_f, c_fp, c_p]),
    'rf_conv3d_valid_leaky_split_ex': (c_i, [c_p, c_i, c_i, c_i, c_i, c_p, c_fp, c_i, c_i, c_i, c_f, c_p, c_i, c_p]),
    'rf_convv_split_pack_weight': (c_i, [c_fp, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    'rf_convv_split_packed_bytes': (c_sz, [c_i, c_i, c_i, c_i, c_i]),
    'rf_conv3d_valid_split_pg_supported': (c_i, [c_i, c_i, c_i, c_i, c_i, c_i]),
    'rf_conv3d_valid_leaky_split_pg': (c_i, [c_p, c_i, c_i, c_i, c_p, c_fp, c_i, c_i, c_i, c_f, c_p, c_p]),
    'rf_convv_split_pg_pack_weight': (c_i, [c_fp, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    'rf_convv_split_pg_packed_bytes': (c_sz, [c_i, c_i, c_i, c_i, c_i]),
    'rf_conv3d_pool_supported': (c_i, [c_i, c_i, c_i, c_i, c_i]),
    'rf_conv3d_k3_gn_relu_pool': (c_i, [c_fp, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_fp, c_p, c_fp, c_p, c_p]),
    'rf_conv3_up_packed_floats': (c_sz, [c_i, c_i, c_i]),
    'rf_conv3_up_pack_weight': (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_p]),
    'rf_conv3d_up_supported': (c_i, [c_i, c_i, c_i, c_i, c_i]),
    'rf_conv3d_up_stats_tiles': (c_i, [c_i, c_i, c_i, c_i, c_i]),
    'rf_conv3d_up_variant': (c_i, [c_i, c_i, c_i, c_i, c_i]),
    'rf_conv3d_up_k3_gn_relu': (c_i, [c_fp, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_fp, c_p, c_p]),
    'rf_conv3_split_packed_bytes': (c_sz, [c_i, c_i]),
    'rf_conv3_split_pack_weight': (c_i, [c_fp, c_i, c_i, c_p, c_p]),
    'rf_conv3d_split_supported': (c_i, [c_i, c_i, c_i, c_i, c_i]),
    'rf_conv3d_split_k3_gn_relu': (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_p, c_i, c_fp, c_p, c_fp, c_p, c_p]),
    'rf_conv3_e2_split_packed_bytes': (c_sz, [c_i, c_i, c_i]),
    'rf_conv3_e2_split_pack_weight': (c_i, [c_fp, c_i, c_i, c_i, c_p, c_p]),
    'rf_conv3d_e2_split_supported': (c_i, [c_i, c_i, c_i, c_i]),
    'rf_conv3d_e2_split_k3_gn_relu': (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_p, c_i, c_fp, c_p, c_p]),
    'rf_split_act_bytes': (c_sz, [c_i, c_i, c_i]),
    'rf_conv3d_cin1_presplit_supported': (c_i, [c_i, c_i, c_i, c_i]),
    'rf_conv3d_cin1_presplit': (c_i, [c_fp, c_i, c_i, c_fp, c_fp, c_f, c_fp, c_i, c_fp, c_fp, c_i, c_f, c_p, c_p]),
    'rf_conv3d_split_pointwise_supported': (c_i, [c_i, c_i, c_i, c_i]),
    'rf_conv3d_split_k3_gn_relu_pointwise_tanh': (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_p, c_i, c_fp, c_fp, c_f, c_f, c_fp, c_p]),
    'rf_conv3d_split_presplit_supported': (c_i, [c_i, c_i, c_i, c_i, c_i]),
    'rf_conv3d_split_presplit': (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_p, c_i, c_fp, c_fp, c_i, c_f, c_p, c_p, c_p]),
    'rf_conv3d_up_split_presplit_supported': (c_i, [c_i, c_i, c_i, c_i, c_i, c_i]),
    'rf_conv3d_up_split_presplit': (c_i, [c_fp, c_i, c_fp, c_i, c_i, c_i, c_fp, c_p, c_i, c_fp, c_fp, c_i, c_f, c_p, c_p, c_p]),
    'rf_conv3d_split_pre_supported': (c_i, [c_i, c_i, c_i, c_i]),
    'rf_conv3d_split_pre_stats_tiles': (c_i, [c_i, c_i, c_i, c_i]),
    'rf_conv3d_split_pre_k3_relu': (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_fp, c_p, c_fp, c_p, c_p]),
    'rf_conv3d_up_split_presplit_pm_supported': (c_i, [c_i, c_i, c_i, c_i, c_i, c_i]),
    'rf_conv3d_up_split_presplit_pm': (c_i, [c_fp, c_i, c_fp, c_i, c_i, c_i, c_fp, c_p, c_i, c_fp, c_fp, c_i, c_f, c_p, c_p, c_p]),
    'rf_conv3d_split_pre_pm_supported': (c_i, [c_i, c_i, c_i, c_i]),
    'rf_conv3d_split_pre_pm_k3_relu': (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_fp, c_p, c_fp, c_p, c_p]),
    'rf_conv3_up_split_packed_bytes': (c_sz, [c_i, c_i, c_i]),
    'rf_conv3_up_split_pack_weight': (c_i, [c_fp, c_i, c_i, c_i, c_p, c_p]),
    'rf_conv3d_up_split_supported': (c_i, [c_i, c_i, c_i, c_i, c_i]),
    'rf_conv3d_up_split_stats_tiles': (c_i, [c_i, c_i, c_i, c_i, c_i]),
    'rf_conv3d_up_split_k3_gn_relu': (c_i, [c_fp, c_i, c_fp, c_i, c_i, c_i, c_fp, c_p, c_i, c_fp, c_p, c_p]),
    'rf_conv3d_k3_gn': (c_i, [c_fp, c_i, c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_i, c_fp, c_p]),
    'rf_relu_backward': (c_i, [c_fp, c_fp, c_sz, c_fp, c_p]),
    'rf_relu_backward_amax_slots': (c_i, []),
    'rf_relu_backward_amax': (c_i, [c_fp, c_fp, c_sz, c_fp, c_fp, c_p]),
    'rf_dgrad_scale_affine': (c_i, [c_fp, c_i, c_fp, c_fp, c_p]),
    'rf_conv3d_split_k3_gn_supported': (c_i, [c_i, c_i, c_i, c_i]),
    'rf_conv3d_split_k3_gn': (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_p, c_i, c_i, c_fp, c_p]),
    'rf_maxpool3d_2_backward': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_fp, c_p]),
    'rf_upsample3d_2': (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_p]),
    'rf_sumpool3d_2': (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_p]),
    'rf_gn_backward': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_fp, c_i, c_f, c_fp, c_fp, c_fp, c_fp, c_p, c_sz, c_p]),
    'rf_gn_backward_ws_bytes': (c_sz, [c_i, c_i, c_i]),
    'rf_conv3d_k3_wgrad': (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_fp, c_p, c_sz, c_p]),
    'rf_conv3d_k3_wgrad_ws_bytes': (c_sz, [c_i, c_i, c_i, c_i]),
    'rf_conv3d_k3_wgrad_split_supported': (c_i, [c_i, c_i, c_i, c_i]),
    'rf_conv3d_k3_wgrad_split_ws_bytes': (c_sz, [c_i, c_i, c_i, c_i]),
    'rf_conv3d_k3_wgrad_split': (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_fp, c_fp, c_p, c_sz, c_p]),
    'rf_unfold3d': (c_i, [c_fp, c_i, c_i, c_i, c_i, c_fp, c_p]),
    'rf_fold3d': (c_i, [c_fp, c_i, c_i, c_i, c_i, c_fp, c_p]),
    'rf_linear_pack_weight': (c_i, [c_fp, c_i, c_i, c_fp, c_p]),
    'rf_linear_packed_floats': (c_sz, [c_i, c_i]),
    'rf_linear': (c_i, [c_fp, c_i, c_i, c_fp, c_fp, c_i, c_i, c_f, c_fp, c_p]),
    'rf_linear_wgrad': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_fp, c_p, c_sz, c_p]),
    'rf_linear_wgrad_ws_bytes': (c_sz, [c_i, c_i, c_i]),
    'rf_l2_normalize_rows': (c_i, [c_fp, c_i, c_i, c_f, c_p]),
    'rf_attn_fuse': (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_f, c_fp, c_fp, c_fp, c_p]),
    'rf_attn_gather_retrieved': (c_i, [c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_fp, c_p]),
    'rf_attn_mlp_packed_floats': (c_sz, [c_i]),
    'rf_attn_mlp_pack': (c_i, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i, c_fp, c_p]),
    'rf_attn_mlp_rows': (c_i, [c_fp, c_i, c_i, c_fp, c_fp, c_p]),
    'rf_attn_mlp_split_packed_floats': (c_sz, [c_i]),
    'rf_attn_mlp_split_pack': (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_fp, c_p]),
    'rf_attn_mlp_split_rows': (c_i, [c_fp, c_i, c_i, c_fp, c_fp, c_fp, c_p]),
    'rf_attn_mlp_split_volume': (c_i, [c_fp, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_p]),
    'rf_attn_mlp_volume': (c_i, [c_fp, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp, c_p]),
    'rf_attn_weights': (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_i, c_f, c_fp, c_fp, c_fp, c_p]),
    'rf_attn_weights_sampled': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_f, c_p, c_fp, c_fp, c_fp, c_fp, c_p]),
    'rf_attn_blend': (c_i, [c_fp, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp, c_fp, c_fp, c_p]),
    'rf_query_windows': (c_i, [c_fp, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_fp, c_p]),
    'rf_gather_windows': (c_i, [c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp, c_p]),
    'rf_gather_windows_split': (c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    'rf_db_pack_embeddings': (c_i, [c_fp, c_i64, c_i, c_fp, c_p]),
    'rf_db_packed_floats': (c_sz, [c_i64, c_i]),
    'rf_l2_topk': (c_i, [c_fp, c_i, c_i, c_fp, c_i64, c_i64, c_i, c_i, c_fp, c_p, c_p, c_sz, c_p]),
    'rf_l2_topk_keys': (c_i, [c_fp, c_i, c_i, c_fp, c_i64, c_i64, c_i, c_i, c_p, c_p, c_sz, c_p]),
    'rf_topk_merge_keys': (c_i, [c_p, c_i, c_i, c_i, c_fp, c_p, c_p]),
    'rf_l2_topk_ws_bytes': (c_sz, [c_i, c_i64, c_i]),
    'rf_topk_merge': (c_i, [c_fp, c_p, c_i, c_i, c_i, c_fp, c_p, c_p]),
    'rf_demote_same_scene': (c_i, [c_fp, c_p, c_i, c_i, c_p, c_p, c_p, c_i, c_p, c_fp, c_p, c_p]),
    'rf_gather_rows': (c_i, [c_fp, c_i64, c_p, c_i64, c_i, c_fp, c_p]),
    'rf_conv3d_split_pre_pool_presplit_supported': (c_i, [c_i, c_i, c_i, c_i, c_i]),
    'rf_conv3d_split_pre_k3_relu_pool_presplit': (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_fp, c_p, c_fp, c_fp, c_i, c_f, c_p, c_p]),
    'rf_conv3d_split_pre_pool_presplit_scratch_floats': (ctypes.c_size_t, [c_i]),
    'rf_conv3d_split_pre_presplit_supported': (c_i, [c_i, c_i, c_i, c_i, c_i]),
    'rf_conv3d_split_pre_presplit': (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_fp, c_fp, c_i, c_f, c_p, c_p, c_p]),
    'rf_conv3d_up_split_ch8_supported': (c_i, [c_i, c_i, c_i, c_i, c_i]),
    'rf_conv3d_up_split_k3_gn_relu_ch8': (c_i, [c_fp, c_i, c_fp, c_i, c_i, c_i, c_fp, c_p, c_i, c_fp, c_p, c_p]),
    'rf_conv3d_split_pointwise_ch8_supported': (c_i, [c_i, c_i, c_i, c_i]),
    'rf_conv3d_split_k3_gn_relu_pointwise_tanh_ch8': (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_p, c_i, c_fp, c_fp, c_f, c_f, c_fp, c_p]),
    'rf_mc_classify': (c_i, [c_fp, c_i, c_i, c_i, c_f, c_p, c_p, c_p, c_p]),
    'rf_mc_emit': (c_i, [c_fp, c_i, c_i, c_i, c_f, c_p, c_p, c_p, c_p, c_p, c_fp, c_p, c_p]),
    'rf_gather_patches': (c_i, [c_fp, c_i64, c_p, c_i, c_i, c_f, c_f, c_f, c_f, c_i, c_fp, c_p]),
    'rf_gather_patches_f16': (c_i, [c_p, c_i64, c_p, c_i, c_i, c_f, c_f, c_f, c_f, c_i, c_fp, c_p]),
    'rf_compose_overlap': (c_i, [c_p, c_i, c_i64, c_fp, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_fp, c_fp, c_p]),
    'rf_paste_chunks': (c_i, [c_fp, c_i, c_p, c_p, c_i, c_i64, c_i64, c_i, c_p, c_p]),
}

_lib = None


class _Library:
    """The bound entry points as plain attributes (no indirection on the hot path).  ``start_profile`` swaps every stream-ordered
    entry point for a wrapper that brackets the call with HIP events on the launch stream and records
    (name, integer arguments, start, end, positions of the null pointer arguments) -- bench.py's per-kernel table; ``stop_profile`` restores the direct bindings."""

    def __init__(self, cdll):
        self._cdll = cdll
        self._direct = {}
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(cdll, name)      # AttributeError here == header/library mismatch
            fn.restype = res
            fn.argtypes = args
            self._direct[name] = fn
            setattr(self, name, fn)

    def start_profile(self, records, only=None):
        """``only``: a set of entry-point names -- bracket just those (the timed region of bench.py times its one dominant entry point this way)."""
        def timed(name, fn):
            def call(*args):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = fn(*args)
                e1.record()
                records.append((name, tuple(a for a in args if isinstance(a, int)), e0, e1, tuple(i for i, a in enumerate(args) if a is None or (isinstance(a, ctypes.c_void_p) and not a.value))))
                return rc
            return call
        for name, (res, args) in SIGNATURES.items():
            if only is not None and name not in only:
                continue
            if res is c_i and args and name not in ('rf_abi_version',):     # int-returning launches (all take the stream last)
                setattr(self, name, timed(name, self._direct[name]))

    def stop_profile(self):
        for name, fn in self._direct.items():
            setattr(self, name, fn)


def load():
    """dlopen the library (once) and attach argument/return types.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(
            'librfuse_hip.so not found at %s -- build it with `python retrieval-fuse_amd/csrc/build.py` '
            '(or __graft_entry__.build()).  There is no CPU fallback for the refinement hot path.' % LIB_PATH)
    _lib = _Library(ctypes.CDLL(str(LIB_PATH)))
    return _lib


def check(rc, what):
    if rc != 0:
        msg = load().rf_last_error()
        raise RuntimeError('%s failed (rc=%d): %s' % (what, rc, msg.decode() if msg else ''))
