"""Patch attention -- MI355X-native stand-ins for the reference's model/attention.py (same names, constructor
arguments, ``state_dict`` keys incl. the unused-but-serialised sig_scale / sig_shift).  The no-grad route below is the
hot path; with grad enabled and inputs that require it the modules take rfuse/autograd.py (HIP kernels for the Linear layers,
plain differentiable torch ops for the per-row weights and the blend: the N4 training slice).

  AttentionFeatureEncoder  = 4 x rf_linear (fp32 MFMA GEMM, LeakyReLU(0.01) fused)          reference :29-46
  AttentionBlock.forward   = theta/phi encoders + rf_attn_fuse                               reference :84-113
  PatchedAttentionBlock    = rf_unfold3d / rf_attn_gather_retrieved / AttentionBlock / rf_fold3d   reference :141-157
  Fold3D / Unfold3D        = rf_fold3d / rf_unfold3d (pure index remaps)                     reference :160-188

Gumbel-hard attention (``retrieval_mode=True``, ShapeNet configs) is stochastic in the reference
(``gumbel_softmax`` draws noise inside forward, reference :101-102); here the noise is an explicit tensor: pass
``gumbel_noise`` to forward, or leave it None to draw ``-log(Exponential(1))`` samples on the device.
"""
import math

import torch
from torch import nn

from rfuse import ops


class LinearParams(nn.Module):
    """nn.Linear parameter holder (keys ``weight`` [out,in], ``bias`` [out]; default Linear initialisation)."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_features) if in_features > 0 else 0
        nn.init.uniform_(self.bias, -bound, bound)
        self._packed = ops.PackedWeight('linear')

    def apply_to(self, x, act=ops.ACT_NONE, slope=0.0):
        return ops.linear(x, self._packed.get(self.weight), self.bias, self.out_features, act, slope)

    def extra_repr(self):
        return f'in_features={self.in_features}, out_features={self.out_features}'


class ActivationMarker(nn.Module):
    """Parameter-free place-holder that keeps Sequential / ModuleList indices aligned with the reference."""

    def __init__(self, kind, slope=0.0):
        super().__init__()
        self.kind, self.slope = kind, slope

    def extra_repr(self):
        return f'{self.kind}, slope={self.slope}'


class AttentionFeatureEncoder(nn.Module):
    """Linear(n_in,128) LeakyReLU Linear(128,128) LeakyReLU Linear(128,128) LeakyReLU Linear(128,n_out); reference :29-46."""

    def __init__(self, n_in, n_out, e):
        super().__init__()
        self.n_in = n_in * (e ** 3)
        self.n_out = n_out
        print("Attention Feature: ", self.n_in, "-->", self.n_out)
        self.encoder = nn.ModuleList([
            LinearParams(self.n_in, 128), ActivationMarker('leaky_relu', 0.01),
            LinearParams(128, 128), ActivationMarker('leaky_relu', 0.01),
            LinearParams(128, 128), ActivationMarker('leaky_relu', 0.01),
            LinearParams(128, self.n_out),
        ])
        self._fused = ops.PackedAttnMLP()

    def fusable(self):
        """True when the 4 layers fit the fused MFMA kernel (rf_attn_mlp_*): n_in a multiple of 16 up to 128, 32 outputs."""
        return ops.USE_FUSED_ATTN_MLP and self.n_in % 16 == 0 and 16 <= self.n_in <= 128 and self.n_out == 32

    def packed_fused(self):
        return self._fused.get([self.encoder[i] for i in (0, 2, 4, 6)])

    def forward(self, x):
        x = x.reshape((x.shape[0], self.n_in))
        if ops.needs_grad(x, self.encoder[0].weight):
            # training slice: per-layer rf_linear behind torch.autograd.Function (rfuse/autograd.py)
            from rfuse import autograd as rf_autograd
            for i in (0, 2, 4):
                x = rf_autograd.Linear.apply(x, self.encoder[i].weight, self.encoder[i].bias, ops.ACT_LEAKY, 0.01)
            return rf_autograd.Linear.apply(x, self.encoder[6].weight, self.encoder[6].bias, ops.ACT_NONE, 0.0)
        if self.fusable():
            return ops.attn_mlp_rows(x.contiguous(), self.packed_fused())
        for i in (0, 2, 4):
            x = self.encoder[i].apply_to(x, ops.ACT_LEAKY, 0.01)
        return self.encoder[6].apply_to(x)


class AttentionBlock(nn.Module):
    """reference model/attention.py:49-116.  Built for the configuration every shipped config uses: normalize=True,
    no_output_mapping=True (g = o = Identity), blend=True; the switch is relu(max_k scores) regardless of use_switching
    (reference :97-99)."""

    def __init__(self, num_output_channels, patch_extent, K, normalize, use_switching, retrieval_mode, no_output_mapping, blend):
        super().__init__()
        if not (normalize and no_output_mapping and blend):
            raise NotImplementedError('only normalize=True, no_output_mapping=True, blend=True is built (all shipped configs)')
        self.cf_op = num_output_channels
        self.cf_feat = 32
        self.K = K
        self.patch_extent = patch_extent
        self.theta = AttentionFeatureEncoder(num_output_channels, self.cf_feat, patch_extent)
        self.phi = AttentionFeatureEncoder(num_output_channels, self.cf_feat, patch_extent)
        self.init_scale = 35
        self.init_shift = -27
        self.sig_scale = nn.Parameter(torch.ones(1) * self.init_scale)
        self.sig_shift = nn.Parameter(torch.ones(1) * self.init_shift)
        self.retrieval_mode = retrieval_mode
        self.blend_mode = blend
        self.use_switching = use_switching
        self.normalize = normalize

    def _gumbel_state(self, device):
        """per-device generator state of the in-kernel Gumbel sampler, seeded from torch's seed at first use (torch.manual_seed before that)"""
        states = self.__dict__.setdefault('_gumbel_states', {})
        key = str(device)
        if key not in states:
            states[key] = ops.gumbel_rng_state(device)
        return states[key]

    @staticmethod
    def sample_gumbel(rows, k, device):
        """-log(Exponential(1)) samples, the draw torch's gumbel_softmax makes (reference :102)."""
        return -torch.empty(rows, k, device=device, dtype=torch.float32).exponential_().log()

    def get_features(self, x, p):
        """x, p: [B,C,E,E,E] -> L2-normalised theta(x), phi(p) [B,32]; reference :72-82."""
        b = x.shape[0]
        x_feat = self.theta(x.contiguous()).reshape((b, -1))
        p_feat = self.phi(p.contiguous()).reshape((b, -1))
        return ops.l2_normalize_rows_(x_feat), ops.l2_normalize_rows_(p_feat)

    def _forward_autograd(self, x, p, gumbel_noise):
        """Grad mode (training slice): the two feature encoders run rf_linear behind rfuse.autograd.Linear (the
        GEMMs are 99 % of this block's work); the per-row rest -- normalise, K scores, switch, softmax / straight-through
        Gumbel-hard, mix, blend -- is a few elementwise torch ops that autograd differentiates."""
        b, k = p.shape[0], p.shape[1]
        xf = nn.functional.normalize(self.theta(x), dim=1)
        pf = nn.functional.normalize(self.phi(p.reshape((b * k,) + tuple(p.shape[2:]))), dim=1).reshape(b, k, -1)
        scores = (xf.unsqueeze(1) * pf).sum(dim=2)                                   # [b, K]
        switch = scores.amax(dim=1, keepdim=True).clamp_min(0.0)
        if self.retrieval_mode:
            if gumbel_noise is None:
                gumbel_noise = self.sample_gumbel(b, k, x.device)
            soft = torch.softmax(scores * 25.0 + gumbel_noise, dim=1)
            hard = torch.zeros_like(soft).scatter_(1, soft.argmax(dim=1, keepdim=True), 1.0)
            weights = hard - soft.detach() + soft                                    # straight-through estimator of gumbel_softmax(hard=True)
        else:
            weights = torch.softmax(scores * float((self.cf_feat * self.patch_extent ** 3) * 4), dim=1)
        mixed = (weights.unsqueeze(2) * p.reshape(b, k, -1)).sum(dim=1)
        flat = x.reshape(b, -1)
        return (flat * (1.0 - switch) + mixed * switch).reshape(x.shape)

    def forward(self, x, p, gumbel_noise=None, debug=None):
        """x: [B,C,E,E,E]; p: [B,K,C,E,E,E] -> [B,C,E,E,E]; reference :84-113."""
        b, k, c, e = p.shape[0], p.shape[1], p.shape[2], p.shape[3]
        if k != self.K:
            raise ValueError(f'expected K={self.K} retrieved patches per row, got {k} (reference MaxPool1d(K) would window silently)')
        if ops.needs_grad(x, p, self.theta.encoder[0].weight):
            return self._forward_autograd(x, p, gumbel_noise)
        x, p = x.contiguous(), p.contiguous()
        x_feat = self.theta(x)
        p_feat = self.phi(p.reshape(b * k, c, e, e, e))
        if self.retrieval_mode:
            if gumbel_noise is None:
                gumbel_noise = self.sample_gumbel(b, k, x.device)
            out = ops.attn_fuse(x, p, x_feat, p_feat, gumbel_noise.contiguous(), ops.ATTN_GUMBEL_HARD, 25.0, debug=debug is not None)
        else:
            sharpness = float((self.cf_feat * e * e * e) * 4)
            out = ops.attn_fuse(x, p, x_feat, p_feat, None, ops.ATTN_SOFTMAX, sharpness, debug=debug is not None)
        if debug is not None:
            out, debug['scores'], debug['weights'] = out
        return out

    def volume_route_ok(self):
        """The volume-domain kernels (rf_attn_mlp_volume / rf_attn_weights / rf_attn_blend) cover attention patch extent 2."""
        return self.patch_extent == 2 and self.cf_op % 2 == 0 and self.theta.fusable() and self.phi.fusable()

    def forward_volumes(self, x_predicted, retrieved, patch_edge, gumbel_noise=None, debug=None):
        """Same result as unfold -> forward -> fold, computed in the folded layout: x_predicted [B,C,S,S,S]; ``retrieved`` the
        features of the B*K retrieved volumes, patch-major with patch edge ``patch_edge`` (== S: plain NCDHW volumes)."""
        ops._no_grad_only(x_predicted, retrieved, self.theta.encoder[0].weight)
        b, c, s = x_predicted.shape[0], x_predicted.shape[1], x_predicted.shape[2]
        x_predicted, retrieved = x_predicted.contiguous(), retrieved.contiguous()
        x_feat = ops.attn_mlp_volume(x_predicted, b, 1, c, s, s, self.theta.packed_fused())
        p_feat = ops.attn_mlp_volume(retrieved, b, self.K, c, s, patch_edge, self.phi.packed_fused())
        rows = x_feat.shape[0]
        if self.retrieval_mode and gumbel_noise is None and debug is None:
            # the noise of gumbel_softmax drawn inside the weights kernel (Philox; no sampling launches, fresh noise under graph replay)
            res = ops.attn_weights_sampled(x_feat, p_feat, self.K, 25.0, self._gumbel_state(x_predicted.device))
        elif self.retrieval_mode:
            if gumbel_noise is None:
                gumbel_noise = self.sample_gumbel(rows, self.K, x_predicted.device)
            res = ops.attn_weights(x_feat, p_feat, gumbel_noise.contiguous(), self.K, ops.ATTN_GUMBEL_HARD, 25.0, debug=debug is not None)
        else:
            sharpness = float((self.cf_feat * self.patch_extent ** 3) * 4)
            res = ops.attn_weights(x_feat, p_feat, None, self.K, ops.ATTN_SOFTMAX, sharpness, debug=debug is not None)
        if debug is not None:
            debug['scores'], debug['weights'] = res[2], res[0]
        return ops.attn_blend(x_predicted, retrieved, self.K, patch_edge, res[0], res[1])

    def get_regularization_losses(self):
        return ((self.sig_scale - self.init_scale) ** 2 + (self.sig_shift - self.init_shift) ** 2) if self.use_switching else 0


class PatchedAttentionBlock(nn.Module):
    """reference model/attention.py:119-157."""

    def __init__(self, nf, num_patch_x, patch_extent, num_nearest_neighbors, attention_block):
        super().__init__()
        self.num_patch_x = num_patch_x
        self.patch_extent = patch_extent
        self.num_nearest_neighbors = num_nearest_neighbors
        self.nf = nf
        self.attention_blocks_layer = attention_block
        self.fold_3d = Fold3D(num_patch_x, patch_extent, self.nf)
        self.unfold_3d = Unfold3D(patch_extent, self.nf)
        self.unfold_3d_occ = Unfold3D(patch_extent, 1)

    def get_features(self, x_predicted, x_target, occupancy):
        """reference :132-139 -> (theta feats [N,32], phi feats [N,32], per-patch occupancy any() [N] bool)."""
        x_predicted_feat_ = self.unfold_3d(x_predicted)
        x_target_feat_ = self.unfold_3d(x_target)
        occupancy_ = self.unfold_3d_occ(occupancy.to(torch.float32))
        x_feat_flat, p_feat_flat = self.attention_blocks_layer.get_features(x_predicted_feat_, x_target_feat_)
        occupancy_flat = occupancy_.reshape((x_predicted_feat_.shape[0], -1)).ne(0).any(dim=1)
        return x_feat_flat, p_feat_flat, occupancy_flat

    def _forward_autograd(self, x_predicted, x_retrieved, gumbel_noise):
        """grad mode: unfold / regroup / fold as differentiable views (no kernels), AttentionBlock._forward_autograd in between"""
        b, c, s = x_predicted.shape[0], x_predicted.shape[1], x_predicted.shape[-1]
        e, k = self.patch_extent, self.num_nearest_neighbors
        r = s // e

        def rows(v):                                                    # [n, c, s,s,s] -> [n, r,r,r, c, e,e,e]
            return v.reshape(v.shape[0], c, r, e, r, e, r, e).permute(0, 2, 4, 6, 1, 3, 5, 7)
        x_rows = rows(x_predicted).reshape(-1, c, e, e, e)
        p_rows = rows(x_retrieved).reshape(b, k, r, r, r, c, e, e, e).permute(0, 2, 3, 4, 1, 5, 6, 7, 8).reshape(-1, k, c, e, e, e)
        out = self.attention_blocks_layer(x_rows, p_rows, gumbel_noise)
        return out.reshape(b, r, r, r, c, e, e, e).permute(0, 4, 1, 5, 2, 6, 3, 7).reshape(b, c, s, s, s)

    def forward(self, x_predicted, x_retrieved, gumbel_noise=None, debug=None):
        """x_predicted [B,F,S,S,S]; x_retrieved [B*K,F,S,S,S] (folded volumes) -> [B,F,S,S,S]."""
        b, s = x_predicted.shape[0], x_predicted.shape[-1]
        if ops.needs_grad(x_predicted, x_retrieved, self.attention_blocks_layer.theta.encoder[0].weight):
            return self._forward_autograd(x_predicted, x_retrieved, gumbel_noise)
        if self.attention_blocks_layer.volume_route_ok():
            return self.attention_blocks_layer.forward_volumes(x_predicted, x_retrieved, s, gumbel_noise, debug)
        x_rows = self.unfold_3d(x_predicted)
        p_rows = ops.attn_gather_retrieved(x_retrieved.contiguous(), 0, b, self.num_nearest_neighbors, self.nf, s, self.patch_extent)
        out_rows = self.attention_blocks_layer(x_rows, p_rows, gumbel_noise, debug)
        return self.fold_3d(out_rows)

    def forward_patch_major(self, x_predicted, retrieved_patch_features, patch_edge, gumbel_noise=None):
        """Same as forward, but the retrieved features come straight from the retrieval backbone in its patch-major
        layout [(B*K*q^3), F, t,t,t] (what Fold3D(q, t, F) would consume, trainer/train_refinement.py:37,112): the fold
        is never materialised."""
        b, s = x_predicted.shape[0], x_predicted.shape[-1]
        if ops.needs_grad(x_predicted, retrieved_patch_features, self.attention_blocks_layer.theta.encoder[0].weight):
            q, c = s // patch_edge, x_predicted.shape[1]               # fold the patch-major features (Fold3D semantics) as a view
            vols = retrieved_patch_features.reshape(-1, q, q, q, c, patch_edge, patch_edge, patch_edge).permute(0, 4, 1, 5, 2, 6, 3, 7).reshape(-1, c, s, s, s)
            return self._forward_autograd(x_predicted, vols, gumbel_noise)
        if self.attention_blocks_layer.volume_route_ok():
            return self.attention_blocks_layer.forward_volumes(x_predicted, retrieved_patch_features, patch_edge, gumbel_noise)
        x_rows = self.unfold_3d(x_predicted)
        p_rows = ops.attn_gather_retrieved(retrieved_patch_features.contiguous(), 1, b, self.num_nearest_neighbors, self.nf, s,
                                           self.patch_extent, patch_edge)
        out_rows = self.attention_blocks_layer(x_rows, p_rows, gumbel_noise)
        return self.fold_3d(out_rows)


class Fold3D(nn.Module):
    """rows [(B*R^3), nf, e,e,e] -> [B, nf, R*e, R*e, R*e]; exact inverse of Unfold3D (reference :160-176)."""

    def __init__(self, num_patch_x, patch_extent, nf):
        super().__init__()
        self.nf = nf
        self.num_patch_x = num_patch_x
        self.patch_extent = patch_extent

    def forward(self, x):
        if ops.needs_grad(x):                                           # grad mode: the same index map as a differentiable view
            r, e, c = self.num_patch_x, self.patch_extent, self.nf
            return x.reshape(-1, r, r, r, c, e, e, e).permute(0, 4, 1, 5, 2, 6, 3, 7).reshape(-1, c, r * e, r * e, r * e)
        return ops.fold3d(x.contiguous(), self.num_patch_x, self.patch_extent, self.nf)


class Unfold3D(nn.Module):
    """[B, nf, S,S,S] -> non-overlapping patches [(B*R^3), nf, e,e,e], rows ordered (b,px,py,pz) (reference :179-188)."""

    def __init__(self, patch_extent, nf):
        super().__init__()
        self.patch_extent = patch_extent
        self.nf = nf

    def forward(self, x):
        if ops.needs_grad(x):
            e, c, r = self.patch_extent, x.shape[1], x.shape[2] // self.patch_extent
            return x.reshape(x.shape[0], c, r, e, r, e, r, e).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(-1, c, e, e, e)
        return ops.unfold3d(x.contiguous(), self.patch_extent)
