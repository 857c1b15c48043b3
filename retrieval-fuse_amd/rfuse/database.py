"""Patch database resident in HBM, and the (optionally sharded) exact top-k search over it.

Row semantics follow the reference's ``database.npy`` (util/retrieval.py:32,39-45): per row
``[scene_idx, x0,x1,y0,y1,z0,z1, emb_0..emb_63]`` plus a final sentinel row (scene -1, an all-trunc patch,
util/retrieval.py:21-26,45).  On the device it is kept as structure-of-arrays:

  emb_packed  float32 [ceil(n_local/64)][64 dims][64 rows]   this rank's shard of the embedding matrix (scan layout)
  meta        int32   [N+1][7]                               replicated (28 B/row)
  volumes     float32 [S][64][64][64]                        replicated scene chunks the 16^3 boxes point into

Sharding: rank g scans rows [g*N/W, (g+1)*N/W) for ALL queries and the per-shard top-2K
(dist, global row id) lists are exchanged with one all-gather and merged -- the only collective on the path.
The voxel store is replicated (1 M patches = 15 625 chunks = 16.4 GB fp32, trivial against 288 GB), so no payload
exchange is needed.
"""
import torch

from . import ops


def shard_bounds(n_rows, rank, world):
    """Contiguous row range of ``rank``; the remainder goes to the low ranks."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allgather_merge(q_local, local_topk, merge, k2, group=None):
    """The sharded search protocol, independent of how the local scan / merge are computed (so it runs on gloo/CPU
    in tests with numpy callables and on RCCL with the HIP kernels):

      1. all-gather the queries (every shard must see every query)
      2. local_topk(all_queries) -> (dist [Q,k2] f32, idx [Q,k2] i64 global ids) over this rank's rows
      3. all-gather the candidate lists -> [W,Q,k2]
      4. merge(dist_parts, idx_parts) restricted to this rank's own queries -> ([q_local,k2], [q_local,k2])

    All ranks must pass the same number of local queries (chunk batches are split evenly)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nq = q_local.shape[0]
    q_all = torch.empty((world * nq,) + tuple(q_local.shape[1:]), dtype=q_local.dtype, device=q_local.device)
    dist.all_gather_into_tensor(q_all, q_local.contiguous(), group=group)
    d_loc, i_loc = local_topk(q_all)
    nq_all, k2_ = d_loc.shape
    d_flat = torch.empty((world * nq_all, k2_), dtype=d_loc.dtype, device=d_loc.device)      # rank-major concatenation
    i_flat = torch.empty((world * nq_all, k2_), dtype=i_loc.dtype, device=i_loc.device)
    dist.all_gather_into_tensor(d_flat, d_loc.contiguous(), group=group)
    dist.all_gather_into_tensor(i_flat, i_loc.contiguous(), group=group)
    d_all, i_all = d_flat.view(world, nq_all, k2_), i_flat.view(world, nq_all, k2_)
    mine = slice(rank * nq, (rank + 1) * nq)
    return merge(d_all[:, mine].contiguous(), i_all[:, mine].contiguous())


@torch.no_grad()
def build_database_rows(config, fenc_target, volumes, device, patch_mask=None, chunks_per_batch=8):
    """``create_dictionary`` on the device (reference util/retrieval.py:29-45).

    volumes [S,64,64,64] raw (un-normalised) target chunks.  For every chunk the 64 target windows
    (patch_size_target + 2*context, padded with the truncation value, dataset/scene.py:71,94; stride = patch size) are
    cut and normalised by rf_query_windows, embedded by ``fenc_target`` (valid-conv patch encoder, HIP kernels) and
    L2-normalised; rows get the UN-padded boxes (util/retrieval.py:41-44).  The last row is the reference's "zero patch"
    sentinel: scene -1, box (0,ps)^3, embedding of an all-ones un-normalised window (util/retrieval.py:21-26,45).
    ``patch_mask`` [S,64] bool keeps a subset (the dataset's occupancy filter lives outside the hot path).
    Returns (emb [N+1,latent] float32, meta [N+1,7] int32) on ``device``."""
    from .configs import truncations
    from .synthetic import patch_boxes_64
    g, d = config['query_geometry'], config['dataset_train']
    ps, ctx = g['patch_size_target'], g['patch_context_target']
    if ps != 16:
        raise NotImplementedError('the database rows assume 16^3 target patches (every shipped config)')
    _, trunc_t = truncations(config)
    device = torch.device(device)
    fenc_target = fenc_target.to(device).eval()
    vols = torch.as_tensor(volumes)
    n_scenes = vols.shape[0]
    embs = []
    for s0 in range(0, n_scenes, chunks_per_batch):
        raw = vols[s0:s0 + chunks_per_batch].to(device, torch.float32).contiguous()
        win = ops.query_windows(raw, ps, ctx, trunc_t, d['target_mean'], d['target_std'])
        z = fenc_target(win)
        embs.append(ops.l2_normalize_rows_(z.reshape(z.shape[0], z.shape[1])))
    w = ps + 2 * ctx
    ones = torch.ones((1, 1, w, w, w), dtype=torch.float32, device=device)
    z = fenc_target(ones)
    embs.append(ops.l2_normalize_rows_(z.reshape(1, -1)))
    emb = torch.cat(embs)
    boxes = torch.from_numpy(patch_boxes_64())
    scene_idx = torch.arange(n_scenes, dtype=torch.int32).repeat_interleave(64)[:, None]
    meta = torch.cat([scene_idx, boxes.repeat(n_scenes, 1)], dim=1)
    if patch_mask is not None:
        keep = torch.as_tensor(patch_mask).reshape(-1).bool()
        meta = meta[keep]
        emb = torch.cat([emb[:-1][keep.to(device)], emb[-1:]])
    sentinel = torch.tensor([[-1, 0, ps, 0, ps, 0, ps]], dtype=torch.int32)
    meta = torch.cat([meta, sentinel]).to(device)
    return emb.contiguous(), meta.contiguous()


class PatchDatabase:
    @classmethod
    def build(cls, config, fenc_target, volumes, device, rank=0, world=1, group=None, patch_mask=None):
        """Database straight from scene chunks: embeddings computed on the device (see build_database_rows)."""
        emb, meta = build_database_rows(config, fenc_target, volumes, device, patch_mask)
        return cls(emb, meta, volumes, device, rank, world, group)

    def __init__(self, emb, meta, volumes, device, rank=0, world=1, group=None):
        """emb [N+1,64] float32 (unit rows), meta [N+1,7] int32, volumes [S,64,64,64] float32 -- host or device tensors /
        numpy arrays of the FULL database; this rank keeps its embedding shard and replicas of meta/volumes."""
        emb = torch.as_tensor(emb)
        self.n_rows = emb.shape[0]
        self.dim = emb.shape[1]
        self.rank, self.world, self.group = rank, world, group
        self.force_collectives = False          # dev/test: run the all-gather + merge protocol even with one rank
        self.lo, self.hi = shard_bounds(self.n_rows, rank, world)
        self.device = torch.device(device)
        shard = emb[self.lo:self.hi].to(self.device, torch.float32).contiguous()
        self.emb_packed = ops.db_pack_embeddings(shard)
        self.meta = torch.as_tensor(meta).to(self.device, torch.int32).contiguous()
        self.volumes = torch.as_tensor(volumes).to(self.device, torch.float32).contiguous()
        self.n_scenes = self.volumes.shape[0]
        self.feature_cache = None        # see build_feature_cache

    @torch.no_grad()
    def build_feature_cache(self, retrieval_backbone, config, rows_per_batch=4096):
        """OPTIONAL serving mode.  The retrieval backbone's output for a database patch does not depend on the query
        (GroupNorm is per sample, patches are processed independently, trainer/train_refinement.py:112), so it can be
        computed once per database row and kept in HBM: [N+1, nf, 8,8,8] fp32 = 32 KB/row (50 k rows 1.6 GB, 1 M rows
        32 GB of 288 GB).  ``RefinementEngine.refine(use_feature_cache=True)`` then replaces the whole retrieval backbone
        (76 of 87 GFLOP per chunk) by a row gather.  This skips work: bench.py reports it separately, never as `value`."""
        from .configs import truncations
        d = config['dataset_train']
        _, trunc_t = truncations(config)
        n = self.n_rows
        feats = None
        for lo in range(0, n, rows_per_batch):
            hi = min(lo + rows_per_batch, n)
            m = self.meta[lo:hi]
            pad = (-m.shape[0]) % 64
            if pad:
                m = torch.cat([m, self.meta[-1:].expand(pad, 7)])
            patches = ops.gather_patches(self.volumes, m.reshape(-1, 1, 7).contiguous(), m.shape[0] // 64, 1, trunc_t, 1.0,
                                         d['target_mean'], d['target_std'], layout=1)
            f = retrieval_backbone(patches)[:hi - lo]
            if feats is None:
                feats = torch.empty((n,) + tuple(f.shape[1:]), dtype=torch.float32, device=self.device)
            feats[lo:hi] = f
        self.feature_cache = feats
        return feats

    def local_topk(self, q, k2):
        """Exact squared-L2 top-k2 of q against this rank's shard, global row ids."""
        return ops.l2_topk(q.contiguous(), self.emb_packed, self.hi - self.lo, self.lo, k2)

    def search(self, q, k2):
        """Top-k2 over the whole database for this rank's queries.  One process: a single scan.  W processes:
        all-gather(queries) -> shard scans -> all-gather(candidates) -> merge."""
        if self.world == 1 and not self.force_collectives:
            return self.local_topk(q, k2)
        return allgather_merge(q, lambda qa: self.local_topk(qa, k2), ops.topk_merge, k2, self.group)

    def retrieve(self, q, K, query_scene=None, query_keep=None):
        """flann_knn_worker semantics (util/retrieval.py:92-100): top-2K, same-scene demotion, keep K.
        ``query_keep`` [Q] bool: False = patch dropped by the query-side occupancy filter (no neighbours, trunc fill).
        Returns (meta [Q,K,7] int32, dist [Q,K], idx [Q,K])."""
        dist, idx = self.search(q, 2 * K)
        return ops.demote_same_scene(dist, idx, self.meta, query_scene, K, query_keep)
