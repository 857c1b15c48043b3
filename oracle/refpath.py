"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path (retrieval-fuse_amd/).

A CPU restatement, written from scratch in functional PyTorch (fp32) and numpy (float64 / integers), of the
refinement-inference hot path of nihalsid/retrieval-fuse.  Every function cites the reference file:line it
follows.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it,
and only as the checker / the reported CPU baseline.

Pinning status
  * network path (A1-A11, H of SURVEY.md section 8): PINNED -- checked against golden vectors captured from the
    reference's own ``model`` package imported in the build container (tests/golden/*.npz, generator
    oracle/gen_golden.py, test tests/test_oracle_golden.py).
  * same-scene demotion / compose (A13, A15): PINNED -- golden outputs captured from the reference's
    ``flann_knn_worker`` / ``create_retrieval_from_mapping`` driven with an exact brute-force FLANN stand-in.
  * kNN itself (A12): PARITY UNPINNED at the FLANN boundary -- the reference calls pyflann (third-party C++,
    not vendored, not pinned in requirements.txt, approximate kd-tree).  The contract used here is exact squared
    L2 top-k in float64 with ties broken towards the lower row index; FLANN's published semantics
    (``nn_index`` returns squared Euclidean distances, ascending) are restated, its approximation is not.

The whole path is weight-shape driven: channel counts, DoubleConv vs StepDownDoubleConv, number of levels are
all read off the ``state_dict`` tensors, so the oracle shares no topology code with the product.
"""
import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------- U-Net pieces

def _groups(channels, num_groups):
    # model/unet.py:61-63 -- a single group when there are fewer channels than groups
    return 1 if channels < num_groups else num_groups


def single_conv_gcr(x, sd, prefix, num_groups):
    """GroupNorm(eps 1e-5, affine) -> Conv3d(k3, pad 1, no bias) -> ReLU.  model/unet.py:19-76 with order 'gcr'."""
    g = _groups(x.shape[1], num_groups)
    x = F.group_norm(x, g, sd[prefix + '.groupnorm.weight'], sd[prefix + '.groupnorm.bias'], eps=1e-5)
    x = F.conv3d(x, sd[prefix + '.conv.weight'], None, stride=1, padding=1)
    return F.relu(x)


def double_conv(x, sd, prefix, num_groups):
    """DoubleConv / StepDownDoubleConv: two SingleConvs; the channel plan lives in the weight shapes.
    model/unet.py:125-144, 149-159."""
    x = single_conv_gcr(x, sd, prefix + '.SingleConv1', num_groups)
    return single_conv_gcr(x, sd, prefix + '.SingleConv2', num_groups)


def _count(sd, prefix, what):
    n = 0
    while (prefix + '%s.%d.basic_module.SingleConv1.conv.weight' % (what, n)) in sd:
        n += 1
    return n


def unet3d(x, sd, prefix, num_groups):
    """Abstract3DUNet.forward, model/unet.py:492-520 (final_conv = Identity, no final activation):
    encoders = [DoubleConv] + [MaxPool3d(2) -> DoubleConv]*; encoder outputs reversed, first dropped (:500-504);
    decoders: nearest upsample to the skip's size (:354-360), concat (skip, upsampled) (:306), DoubleConv;
    zip() stops at the shorter list (:507) which is how remove_n_final_layers drops decoders."""
    n_enc, n_dec = _count(sd, prefix, 'encoders'), _count(sd, prefix, 'decoders')
    feats = []
    for i in range(n_enc):
        if i > 0:
            x = F.max_pool3d(x, kernel_size=2)                                   # model/unet.py:237,250-251
        x = double_conv(x, sd, prefix + 'encoders.%d.basic_module' % i, num_groups)
        feats.insert(0, x)
    feats = feats[1:]
    for j in range(min(n_dec, len(feats))):
        skip = feats[j]
        x = F.interpolate(x, size=skip.shape[2:], mode='nearest')
        x = torch.cat((skip, x), dim=1)
        x = double_conv(x, sd, prefix + 'decoders.%d.basic_module' % j, num_groups)
    return x


def decoder_no_joining(x, sd, prefix, num_groups):
    """DecoderNoJoining.forward, model/unet.py:319-322: nearest x2 upsample, then DoubleConv (no skip)."""
    x = F.interpolate(x, size=[2 * s for s in x.shape[2:]], mode='nearest')
    return double_conv(x, sd, prefix + '.basic_module', num_groups)


# ---------------------------------------------------------------------------------------- refinement wrappers

def unet_backbone(x, sd, config):
    """model/refinement.py:6-45; selection by task / input_chunk_size as in model/__init__.py:41-48."""
    g = config['nf'] // 2                                                       # model/refinement.py:11-13,27-28,41
    if config['task'] == 'surface_reconstruction':
        return unet3d(x, sd, 'network.', g)
    x = unet3d(x, sd, 'network.0.', g)
    x = decoder_no_joining(x, sd, 'network.1', g)
    if config['dataset_train']['input_chunk_size'] == 8:
        x = decoder_no_joining(x, sd, 'network.2', g)
    return x


def retrieval_backbone(x, sd, config):
    """RetrievalUNetBackbone.forward, model/refinement.py:64-73."""
    return unet3d(x, sd, 'network.', config['nf'] // 2)


def final_decoder(x, sd, config):
    """Superresolution08FinalDecoder.forward, model/refinement.py:48-61: DecoderNoJoining, 1x1x1 conv + bias, tanh."""
    x = decoder_no_joining(x, sd, 'network.0', config['nf'] // 2)
    x = F.conv3d(x, sd['network.1.weight'], sd['network.1.bias'])
    return torch.tanh(x)


def network_pred_to_df(pred, target_trunc):
    """trainer/train_refinement.py:242-243."""
    return (pred + 1) * target_trunc / 2


# ------------------------------------------------------------------------------------------- fold / unfold

def unfold3d(x, e):
    """Unfold3D.forward, model/attention.py:186-188: non-overlapping e^3 patches, rows ordered (b,px,py,pz)."""
    b, c, s = x.shape[0], x.shape[1], x.shape[2]
    r = s // e
    x = x.reshape(b, c, r, e, r, e, r, e).permute(0, 2, 4, 6, 1, 3, 5, 7)
    return x.reshape(-1, c, e, e, e)


def fold3d(rows, r, e, c):
    """Fold3D.forward, model/attention.py:170-176: exact inverse of unfold3d."""
    x = rows.reshape(-1, r, r, r, c, e, e, e).permute(0, 4, 1, 5, 2, 6, 3, 7)
    return x.reshape(-1, c, r * e, r * e, r * e)


# --------------------------------------------------------------------------------------------------- attention

def attention_feature_encoder(x, sd, prefix):
    """AttentionFeatureEncoder.forward, model/attention.py:29-46: Linear/LeakyReLU(0.01) x3, Linear."""
    x = x.reshape(x.shape[0], -1)
    for i in (0, 2, 4):
        x = F.leaky_relu(F.linear(x, sd[prefix + '.encoder.%d.weight' % i], sd[prefix + '.encoder.%d.bias' % i]), 0.01)
    return F.linear(x, sd[prefix + '.encoder.6.weight'], sd[prefix + '.encoder.6.bias'])


def attention_block(x, p, sd, prefix, retrieval_mode, gumbel_noise=None, details=None):
    """AttentionBlock.forward, model/attention.py:84-113 with normalize=True, g = o = Identity, blend=True.

    x: [b,C,e,e,e]  p: [b,K,C,e,e,e].  ``gumbel_noise`` [b,K] replaces the -log(Exp(1)) draw inside
    torch's gumbel_softmax for retrieval_mode=True (model/attention.py:100-103)."""
    b, k, c, e = p.shape[0], p.shape[1], p.shape[2], p.shape[3]
    x_feat = attention_feature_encoder(x, sd, prefix + '.theta').reshape(b, -1)
    p_feat = attention_feature_encoder(p.reshape(b * k, -1, e, e, e), sd, prefix + '.phi').reshape(b, k, -1)
    x_feat = F.normalize(x_feat, dim=1)                                          # :92
    p_feat = F.normalize(p_feat, dim=2)                                          # :93
    values = p.reshape(b, k, -1)                                                 # :94, g = Identity
    scores = torch.einsum('ij,ijk->ik', x_feat, p_feat.permute(0, 2, 1))         # :96
    switch = F.relu(scores.max(dim=1, keepdim=True).values)                      # :99 (MaxPool1d(K) over all K)
    if retrieval_mode:
        logits = scores * 25                                                     # :101
        y_soft = ((logits + gumbel_noise) / 1.0).softmax(dim=-1)                 # torch gumbel_softmax, tau = 1
        index = y_soft.max(dim=-1, keepdim=True)[1]
        y_hard = torch.zeros_like(logits).scatter_(-1, index, 1.0)
        weights = y_hard - y_soft.detach() + y_soft                              # hard=True: forward value y_hard, gradient of y_soft (torch's straight-through form)
    else:
        sharpness = (32 * e * e * e) * 4                                         # :105, cf_feat = 32
        weights = torch.softmax(sharpness * scores, dim=1)                       # :106
    weighted = torch.einsum('ij,ijk->ik', weights, values)                       # :103 / :107
    xf = x.reshape(b, -1)
    out = xf * (1 - switch) + weighted * switch                                  # :109-110, blend
    if details is not None:
        details.update(scores=scores, switch=switch, weights=weights, x_feat=x_feat, p_feat=p_feat)
    return out.reshape(b, c, e, e, e)


def patched_attention_block(x_pred, x_retr, sd, config, gumbel_noise=None, details=None):
    """PatchedAttentionBlock.forward, model/attention.py:141-157 (prefix 'attention_blocks_layer')."""
    nf, k = config['nf'], config['K']
    e, r = config['attn_patch_extent'] // 2, config['attn_num_patch']           # model/__init__.py:60-61
    xp = unfold3d(x_pred, e)
    pr = unfold3d(x_retr.reshape(-1, nf, x_retr.shape[-1], x_retr.shape[-1], x_retr.shape[-1]), e)
    pr = pr.reshape(-1, k, r, r, r, nf, e, e, e).permute(0, 2, 3, 4, 1, 5, 6, 7, 8).reshape(-1, k, nf, e, e, e)
    out = attention_block(xp, pr, sd, 'attention_blocks_layer', config['attn_retrieval_mode'], gumbel_noise, details)
    return fold3d(out, r, e, nf)


def patched_get_features(x_pred, x_target, occupancy, sd, config):
    """PatchedAttentionBlock.get_features -> AttentionBlock.get_features, model/attention.py:132-139, 72-82:
    theta on the e^3 patches of x_pred, phi on those of x_target, both L2-normalised; per-patch any() of ``occupancy``."""
    e = config['attn_patch_extent'] // 2
    xp, xt = unfold3d(x_pred, e), unfold3d(x_target, e)
    occ = unfold3d(occupancy, e)
    x_feat = F.normalize(attention_feature_encoder(xp, sd, 'attention_blocks_layer.theta').reshape(xp.shape[0], -1), dim=1)
    p_feat = F.normalize(attention_feature_encoder(xt, sd, 'attention_blocks_layer.phi').reshape(xp.shape[0], -1), dim=1)
    return x_feat, p_feat, occ.reshape(xp.shape[0], -1).any(dim=1)


# ------------------------------------------------------------------------------- runner (trainer.forward_full)

def forward_full(sds, config, x_in, retrievals, target_trunc, gumbel_noise=None, stages=None):
    """RefinementTrainingModule.forward_full main output, trainer/train_refinement.py:108-116, then
    network_pred_to_df (:242-243).  x_in [B,1,S,S,S]; retrievals [B,K,64,64,64] (normalised).
    The reference also pushes batch['target'] through the retrieval backbone (:111); GroupNorm is per sample so
    leaving it out does not change pred_shape."""
    b, k = retrievals.shape[0], config['K']
    x_back = unet_backbone(x_in, sds['unet_backbone'], config)
    retr = retrievals[:, :k].reshape(b * k, 1, 64, 64, 64)                       # get_retrievals, :255-257
    feat = retrieval_backbone(unfold3d(retr, 16), sds['retrieval_backbone'], config)   # Unfold3D(16,1), :34,112
    x_retr = fold3d(feat, 4, 8, config['nf'])                                    # Fold3D(4,8,nf), :37,112
    x = patched_attention_block(x_back, x_retr, sds['patched_attention_block'], config, gumbel_noise)
    pred = final_decoder(x, sds['decoder'], config)
    df = network_pred_to_df(pred, target_trunc)
    if stages is not None:
        stages.update(x_back=x_back, x_retrieval=x_retr, x_attn=x, pred=pred, df=df)
    return df


# -------------------------------------------------------------------------------------------- query embedding

def extract_query_windows(input_raw, config, input_trunc):
    """Query-side patches of one chunk as the retrieval dataset cuts them: pad the raw input by
    patch_context_input with input_trunc (dataset/scene.py:61), enumerate windows with get_extents_for_size
    (dataset/scene.py:152-160, meshgrid 'ij', stride = patch_size_input because patch_stride == patch_size_target),
    normalise (dataset/patched_scene_dataset.py:127).  Returns [P,1,w,w,w] float32, P = (S/ps)^3."""
    g = config['query_geometry']
    ps, pc = g['patch_size_input'], g['patch_context_input']
    d = config['dataset_train']
    s = input_raw.shape[0]
    # surface reconstruction pads the occupancy grid with zeros (util/misc.py:73-78, pad argument)
    pad_val = 0.0 if config['task'] == 'surface_reconstruction' else input_trunc
    padded = np.pad(input_raw.astype(np.float32), pc, mode='constant', constant_values=pad_val)
    n = s // ps
    w = ps + 2 * pc
    out = np.empty((n * n * n, 1, w, w, w), dtype=np.float32)
    i = 0
    for ix in range(n):
        for iy in range(n):
            for iz in range(n):
                out[i, 0] = padded[ix * ps: ix * ps + w, iy * ps: iy * ps + w, iz * ps: iz * ps + w]
                i += 1
    return ((out - np.float32(d['input_mean'])) / np.float32(d['input_std'])).astype(np.float32)


def patch04_embed(x, sd):
    """Patch04.forward, model/retrieval.py:64-84: Linear/ReLU x4, Linear."""
    x = x.reshape(x.shape[0], -1)
    for i in (0, 2, 4, 6):
        x = F.relu(F.linear(x, sd['layers.%d.weight' % i], sd['layers.%d.bias' % i]))
    return F.linear(x, sd['layers.8.weight'], sd['layers.8.bias'])


def conv_patch_embed(x, sd):
    """Valid-conv patch encoders (Patch08 model/retrieval.py:136-156, PCPatch48 :217-243, Patch32 :4-28, ...):
    Conv3d(+bias) / LeakyReLU(0.2) pairs at the even indices of ``layers``, then ``final_layer``.
    Strides are not stored in the weights, so they come from the table below keyed by the kernel-size sequence."""
    ks = []
    i = 0
    while ('layers.%d.weight' % i) in sd:
        ks.append(sd['layers.%d.weight' % i].shape[-1])
        i += 2
    strides = _CONV_ENCODER_STRIDES[tuple(ks)]
    for j, st in enumerate(strides):
        x = F.leaky_relu(F.conv3d(x, sd['layers.%d.weight' % (2 * j)], sd['layers.%d.bias' % (2 * j)], stride=st), 0.2)
    x = x.reshape(x.shape[0], -1)
    return F.linear(x, sd['final_layer.weight'], sd['final_layer.bias'])


_CONV_ENCODER_STRIDES = {
    (3, 3, 3, 2): (1, 1, 1, 1),                 # Patch08      model/retrieval.py:140-147
    (5, 3, 3, 3, 3, 3, 2): (1, 1, 2, 2, 2, 1, 1),   # PCPatch48    model/retrieval.py:221-234
    (5, 3, 3, 3, 3, 4): (1, 1, 2, 1, 2, 1),     # Patch32      model/retrieval.py:8-19
    (3, 3, 3, 3, 3, 3, 3): (1, 1, 2, 1, 1, 1, 1),   # Patch24V2    model/retrieval.py:339-352
}


def embed_queries(windows, sd, config):
    """extract_features core, util/retrieval.py:66: encoder -> [P,z,1,1,1] -> L2-normalise rows."""
    x = torch.from_numpy(windows) if isinstance(windows, np.ndarray) else windows
    if config['retrieval_model']['network_input'] == '2+1':
        z = patch04_embed(x, sd)
    else:
        z = conv_patch_embed(x, sd)
    return F.normalize(z, dim=1)


# ------------------------------------------------------------------------------------------------- retrieval

def knn_exact(queries, db_emb, n_neighbors, block=4096):
    """Exact squared-L2 top-n in float64, ascending, ties -> lower row index.
    Stands where the reference calls FLANN ``nn_index(feats, 2K)`` (util/retrieval.py:92; FLANN returns squared
    Euclidean distances in ascending order).  PARITY UNPINNED against FLANN itself (approximate kd-tree)."""
    q = np.asarray(queries, dtype=np.float64)
    d = np.asarray(db_emb, dtype=np.float64)
    nq = q.shape[0]
    idx = np.empty((nq, n_neighbors), dtype=np.int64)
    dist = np.empty((nq, n_neighbors), dtype=np.float64)
    for s in range(0, nq, block):
        qs = q[s:s + block]
        dd = ((qs[:, None, :] - d[None, :, :]) ** 2).sum(-1) if d.shape[0] * qs.shape[0] <= 2_000_000 else \
            np.stack([((d - qi) ** 2).sum(-1) for qi in qs])
        order = np.argsort(dd, axis=1, kind='stable')[:, :n_neighbors]
        idx[s:s + block] = order
        dist[s:s + block] = np.take_along_axis(dd, order, axis=1)
    return idx, dist


def mapping_rows(idx, dist, db_meta):
    """all_extents, util/retrieval.py:93: per query, per neighbour [scene_idx, x0,x1,y0,y1,z0,z1, dist] float32."""
    rows = np.concatenate([db_meta[idx].astype(np.float32), dist[..., None].astype(np.float32)], axis=-1)
    return rows                                                                  # [Q, n, 8]


def demote_same_scene(rows, query_scene_index, K):
    """util/retrieval.py:94-100: neighbours from the query's own scene go (stably) to the back, keep the first K.
    ``query_scene_index`` [Q] int, -1 = query scene not in the database index (no demotion for it)."""
    out = np.empty((rows.shape[0], K, rows.shape[2]), dtype=rows.dtype)
    for i in range(rows.shape[0]):
        r = rows[i]
        if query_scene_index[i] >= 0:
            m = r[:, 0] == query_scene_index[i]
            r = np.concatenate((r[~m], r[m]))
        out[i] = r[:K]
    return out


def compose_retrieval(mapping, db_volumes, K, target_trunc, trunc_ratio=1.0, patch_keep=None):
    """create_retrieval_from_mapping for one 64^3 chunk with non-overlapping patches (no_overlap is True for every
    shipped config), util/retrieval.py:145-164.  mapping [64,K,8] in the chunk's patch order; returns [K,64,64,64].
    idx < 0 (sentinel row) -> a volume of target_trunc (:160-161).
    ``patch_keep`` [64] bool: the query-side occupancy filter (dataset/patched_scene_dataset.py:28-32) -- patches it drops
    are not in ``patch_from_scene_lookup`` (:151 never visits them) and keep the target_trunc initialisation (:148)."""
    out = np.ones((K, 64, 64, 64), dtype=np.float32) * np.float32(target_trunc)
    o = np.arange(0, 64, 16)
    slots = [(x, y, z) for x in o for y in o for z in o]
    for k in range(K):
        for p, (xx, yy, zz) in enumerate(slots):
            if patch_keep is not None and not patch_keep[p]:
                continue
            sidx = int(mapping[p, k, 0])
            x0, x1, y0, y1, z0, z1 = mapping[p, k, 1:7].astype(np.int32).tolist()
            if sidx >= 0:
                src = db_volumes[sidx][x0:x1, y0:y1, z0:z1]
            else:
                src = np.ones((64, 64, 64), dtype=np.float64)[x0:x1, y0:y1, z0:z1] * target_trunc
            # float32 tensor * python float, as torch does it at :162
            out[k, xx:xx + 16, yy:yy + 16, zz:zz + 16] = (torch.from_numpy(np.ascontiguousarray(src)) * float(trunc_ratio)).numpy()
    return out


def compose_retrieval_overlap(mapping, boxes, db_volumes, K, target_trunc, trunc_ratio=1.0, size=(64, 64, 64), no_overlap=False):
    """create_retrieval_from_mapping (util/retrieval.py:145-164) for ANY patch grid of one scene, the OVERLAPPING one included (patch stride < patch size,
    ``dataset.no_overlap`` False): the patches are visited in the order of ``patch_from_scene_lookup``, and patch p overwrites its box of retrieval k only while the
    MEAN of the distances stored in that box is above its own distance (:156) -- an order-dependent sequential reduction.
    mapping [P,K,8] (scene index, X0, X1, Y0, Y1, Z0, Z1, distance) of the P patches the lookup holds, in its order; boxes [P,6] their unpadded target boxes
    (xx0, xx1, yy0, yy1, zz0, zz1) in the scene (:155).  Returns [K, *size] float32.  The mean is taken in float64 here (torch takes it in float32 with its own
    summation tree: the two agree unless a mean lies within rounding of the patch's distance; oracle/gen_golden.py asserts a margin for the fixture)."""
    out = np.ones((K,) + tuple(size), dtype=np.float32) * np.float32(target_trunc)
    dist = np.ones((K,) + tuple(size), dtype=np.float32) * np.float32(100)
    for k in range(K):
        for p in range(mapping.shape[0]):
            x0, x1, y0, y1, z0, z1 = mapping[p, k, 1:7].astype(np.int32).tolist()
            cur = mapping[p, k, 7]
            xx0, xx1, yy0, yy1, zz0, zz1 = [int(v) for v in boxes[p]]
            if no_overlap or dist[k, xx0:xx1, yy0:yy1, zz0:zz1].astype(np.float64).mean() > float(cur):
                sidx = int(mapping[p, k, 0])
                if sidx >= 0:
                    src = db_volumes[sidx][x0:x1, y0:y1, z0:z1]
                else:
                    src = np.ones(tuple(size), dtype=np.float64)[x0:x1, y0:y1, z0:z1] * target_trunc
                out[k, xx0:xx1, yy0:yy1, zz0:zz1] = (torch.from_numpy(np.ascontiguousarray(src)) * float(trunc_ratio)).numpy()
                dist[k, xx0:xx1, yy0:yy1, zz0:zz1] = float(cur)
    return out


def knn_cdist_f32(queries, db_emb, n_neighbors):
    """fp32 multi-threaded exact kNN (torch.cdist + topk) -- the CPU-baseline kNN named in BASELINE.md section 3.
    Used ONLY for timing the CPU baseline in bench.py; parity checks use knn_exact (float64)."""
    q = torch.as_tensor(queries, dtype=torch.float32)
    d = torch.as_tensor(db_emb, dtype=torch.float32)
    dist = torch.cdist(q, d) ** 2
    vals, idx = torch.topk(dist, n_neighbors, dim=1, largest=False, sorted=True)
    return idx.numpy(), vals.numpy()


# ------------------------------------------------------------------------------- database build ("next" row N1)

def database_rows(volumes, sd_target, config, target_trunc):
    """create_dictionary + get_zero_patch_entry restated (util/retrieval.py:21-45): [S*64 + 1, 7 + latent] float32 rows.
    volumes [S,64,64,64] raw target chunks; windows = patch_size_target + 2*context, padded with target_trunc
    (dataset/scene.py:71,94), normalised (dataset/patched_scene_dataset.py:128); boxes stored un-padded (:41-44)."""
    g, d = config['query_geometry'], config['dataset_train']
    ps, pc = g['patch_size_target'], g['patch_context_target']
    w = ps + 2 * pc
    rows = []
    for s, vol in enumerate(volumes):
        padded = np.pad(np.asarray(vol, dtype=np.float32), pc, mode='constant', constant_values=target_trunc)
        wins, boxes = [], []
        for x in range(0, 64, ps):
            for y in range(0, 64, ps):
                for z in range(0, 64, ps):
                    wins.append(padded[x:x + w, y:y + w, z:z + w])
                    boxes.append([s, x, x + ps, y, y + ps, z, z + ps])
        wins = ((np.stack(wins)[:, None] - np.float32(d['target_mean'])) / np.float32(d['target_std'])).astype(np.float32)
        with torch.no_grad():
            emb = F.normalize(conv_patch_embed(torch.from_numpy(wins), sd_target), dim=1).numpy()
        rows.append(np.concatenate([np.asarray(boxes, dtype=np.float32), emb], axis=1))
    with torch.no_grad():
        zero = F.normalize(conv_patch_embed(torch.ones(1, 1, w, w, w), sd_target), dim=1).numpy()
    rows.append(np.concatenate([np.array([[-1, 0, ps, 0, ps, 0, ps]], dtype=np.float32), zero], axis=1))
    return np.concatenate(rows).astype(np.float32)
