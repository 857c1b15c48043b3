"""Patch database resident in HBM, and the (optionally sharded) exact top-k search over it.

Row semantics follow the reference's ``database.npy`` (util/retrieval.py:32,39-45): per row
``[scene_idx, x0,x1,y0,y1,z0,z1, emb_0..emb_63]`` plus a final sentinel row (scene -1, an all-trunc patch,
util/retrieval.py:21-26,45).  On the device it is kept as structure-of-arrays:

  emb_packed  float32 [ceil(n_local/64)][64 dims][64 rows]   this rank's shard of the embedding matrix (scan layout)
  meta        int32   [N+1][7]                               replicated (28 B/row)
  volumes     float32 [S][64][64][64]                        replicated scene chunks the 16^3 boxes point into

Sharding: rank g scans rows [g*N/W, (g+1)*N/W) for ALL queries and the per-shard top-2K
(dist, global row id) lists are exchanged with one all-to-all (each rank receives the candidates of its own queries) and merged.
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


def make_host_group(group=None):
    """A gloo twin of ``group`` for host-side agreement checks (no GPU work, no device synchronisation).  ``dist.new_group`` is COLLECTIVE OVER THE
    DEFAULT GROUP: every rank of the world must call this at the same point of its program (``PatchDatabase.__init__`` does, for world > 1).
    A gloo group is its own twin."""
    import torch.distributed as dist
    if dist.get_backend(group) == 'gloo':
        return group if group is not None else dist.group.WORLD
    ranks = dist.get_process_group_ranks(group) if group is not None else list(range(dist.get_world_size()))
    return dist.new_group(ranks=ranks, backend='gloo')


class QueryCountCheck:
    """all_gather_into_tensor needs the same number of queries on every rank (chunk batches are split evenly); unequal counts would hang or
    mis-slice silently.  Every call of the sharded search posts ONE non-blocking gloo all-gather of its count (a CPU int64, no device
    synchronisation) -- the same collective sequence on every rank, whatever each rank has seen before -- and

      * whenever this rank's count CHANGES -- a count it has not used before, or one that differs from its previous call's -- it waits for the exchange
        and raises before any device collective is issued (round 4 waited only for a never-seen count: two ranks that had both used 64 and 128 before
        could then walk into an all-gather of 64 against 128);
      * a count equal to the previous call's is not waited for: the exchange of call i is examined at ``flush()`` -- ``PatchDatabase.check()``, which
        the engine calls once the whole step is enqueued and before it hands results out, by which time the peers have long posted theirs -- or at the
        next call.  In steady state no rank's enqueue thread blocks on the slowest rank ahead of its launches (round 3 did one blocking
        ``all_gather_object`` per step).

    A rank whose own count did not change cannot see a peer's change on its own: the peer (whose count changed) refuses at once and never enters the
    device collectives; this rank has only ENQUEUED them and gets the ValueError from ``check()`` before any host synchronisation on the results
    (engine.refine / refine_stream / scene.refine_scene(s) all end a step with it), not a hang."""

    def __init__(self, host_group):
        import torch.distributed as dist
        self.host_group = host_group
        self.world = dist.get_world_size(host_group)
        self.seen = set()
        self.last = None               # this rank's count at the previous call
        self.pending = None            # (work, counts tensors, mine tensor) of the previous call

    @staticmethod
    def _verify(counts):
        vals = [int(c.item()) for c in counts]
        if len(set(vals)) != 1:
            raise ValueError('sharded search: every rank must pass the same number of queries, got %s' % (vals,))

    def flush(self):
        """Wait for and examine the exchange of the previous call (raises ValueError on a mismatch)."""
        if self.pending is not None:
            work, counts, _ = self.pending
            self.pending = None
            work.wait()
            self._verify(counts)

    def post(self, nq, defer=True):
        """``defer=False``: always wait (used when the data-path collectives block the host anyway -- a gloo / CPU data path -- so that no rank
        walks into a collective its peer has refused)."""
        import torch.distributed as dist
        self.flush()
        mine = torch.tensor([int(nq)], dtype=torch.int64)
        counts = [torch.zeros(1, dtype=torch.int64) for _ in range(self.world)]
        work = dist.all_gather(counts, mine, group=self.host_group, async_op=True)
        self.pending = (work, counts, mine)
        changed = nq not in self.seen or nq != self.last
        self.last = nq
        if changed or not defer:
            self.flush()
            self.seen.add(nq)


_fallback_checks = []                  # (group, QueryCountCheck) for direct callers of sharded_search that pass no checker


def _fallback_check(group):
    for g, c in _fallback_checks:
        if g is group:
            return c
    c = QueryCountCheck(make_host_group(group))
    _fallback_checks.append((group, c))
    return c


def sharded_search(q_local, local_topk_keys, merge_keys, k2, group=None, timings=None, count_check=None):
    """The sharded search protocol, independent of how the local scan / merge are computed (so it runs on
    gloo/CPU in tests with numpy stand-ins of the same contract, and on RCCL with the HIP kernels):

      1. all-gather the queries (every shard must see every query)                              [W * nq, 64] float32
      2. local_topk_keys(all_queries) -> packed 64-bit keys (dist bits << 32 | GLOBAL row id, -1 = none) over this rank's rows
      3. ONE all-to-all of the candidate keys: rank r receives, from every shard, the candidates of ITS nq queries  [W, nq, k2] int64
         (an all-gather would move W times as much -- every rank's candidates for every rank's queries -- to use 1/W of it;
         xGMI is point-to-point, an all-to-all of W-1 messages of nq * k2 * 8 bytes is its natural pattern)
      4. merge_keys(the W candidate lists of this rank's own queries) -> (dist [nq, k2], idx [nq, k2])

    The unsigned order of the keys is the (distance, row id) order, so the merge of W shard lists equals a single scan
    bit for bit.  ``timings``: optional list; (start, end) CUDA event pairs of the two collectives are appended.  ``count_check``: the
    caller's QueryCountCheck (PatchDatabase owns one per database); without it a per-group one is created on first use, which is collective over
    the world for a non-gloo group (see make_host_group)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nq = q_local.shape[0]
    # device data path: the collectives below are only enqueued, so the check may trail by one call; host (gloo) data path: they block, check first
    (count_check if count_check is not None else _fallback_check(group)).post(nq, defer=q_local.is_cuda)
    q_local = q_local.contiguous()
    ev = None
    if timings is not None and q_local.is_cuda:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    q_all = torch.empty((world * nq,) + tuple(q_local.shape[1:]), dtype=q_local.dtype, device=q_local.device)
    if ev:
        ev[0].record()
    dist.all_gather_into_tensor(q_all, q_local, group=group)
    if ev:
        ev[1].record()
    keys = local_topk_keys(q_all).contiguous()                                  # [W*nq, k2] int64: rows [r*nq, (r+1)*nq) = rank r's queries
    recv = torch.empty_like(keys)                                               # [W, nq, k2]: part w = shard w's candidates for MY queries
    if ev:
        ev[2].record()
    dist.all_to_all_single(recv, keys, group=group)
    if ev:
        ev[3].record()
        timings.append((ev[0], ev[1], ev[2], ev[3]))
    return merge_keys(recv.view(world, nq, keys.shape[1]))


class HipSearchBackend:
    """The product's scan / merge: librfuse_hip.so through rfuse.ops.  (tests/test_distributed_cpu.py injects a numpy object with
    the same four methods to run the protocol over gloo; the product has no CPU path.)"""

    @staticmethod
    def pack(emb_shard):
        return ops.db_pack_embeddings(emb_shard) if emb_shard.shape[0] else None

    @staticmethod
    def topk(q, packed, n, row_base, k2, algo=0):
        return ops.l2_topk(q, packed, n, row_base, k2, algo)

    @staticmethod
    def topk_keys(q, packed, n, row_base, k2, algo=0):
        if n == 0:                                                              # empty shard (fewer rows than ranks): no candidates
            return torch.full((q.shape[0], k2), -1, dtype=torch.int64, device=q.device)
        return ops.l2_topk_keys(q, packed, n, row_base, k2, algo)

    @staticmethod
    def choose_scan(emb_shard, packed, row_base):
        """Which exact scan serves this shard (0 = by size, ops.TOPK_AUTO).  The matrix-core scans FILTER with a dot product whose slack in the squared distance is
        ~6e-5 for unit rows and re-check every survivor one by one; a shard with CLUMPS -- many rows within that slack of each other: thousands of copies of an
        empty-space patch, a collapsed encoder -- makes every row of a clump a survivor of every query near it (measured: 1024 queries against 50 k rows of one
        1e-6-wide clump 35 ms, against 0.3 ms of the VALU scan, whose cost does not depend on the data).  Probed once, at construction: 512 rows of the shard as
        queries, exact top-16; where the 16th neighbour of more than 2 % of them lies within 2e-4 the shard is searched by the VALU scan."""
        n = emb_shard.shape[0]
        if n < 4096:
            return 0                                                             # (the VALU scan by size anyway)
        probe = emb_shard[torch.linspace(0, n - 1, 512, device=emb_shard.device).long()].contiguous()
        d, _ = ops.l2_topk(probe, packed, n, row_base, 16, ops.TOPK_VALU_SCAN)
        return ops.TOPK_VALU_SCAN if (d[:, 15] < 2e-4).float().mean().item() > 0.02 else 0

    @staticmethod
    def merge_keys(key_parts):
        return ops.topk_merge_keys(key_parts)

    @staticmethod
    def demote(dist, idx, meta, query_scene, K, query_keep):
        return ops.demote_same_scene(dist, idx, meta, query_scene, K, query_keep)


@torch.no_grad()
def build_database_rows(config, fenc_target, volumes, device, patch_mask=None, chunks_per_batch=32):
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
        z = ops.embed_windows(fenc_target, raw, ps, ctx, trunc_t, d['target_mean'], d['target_std'])
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
    def build(cls, config, fenc_target, volumes, device, rank=0, world=1, group=None, patch_mask=None, half_store=None):
        """Database straight from scene chunks: embeddings computed on the device (see build_database_rows)."""
        emb, meta = build_database_rows(config, fenc_target, volumes, device, patch_mask)
        return cls(emb, meta, volumes, device, rank, world, group, half_store=half_store)

    @staticmethod
    def _fits_float16(vols, rows_per_piece=1024):
        """every voxel of the store survives float16 (true for anything read from the reference's float16 scenes) -- checked where the store lives"""
        for lo in range(0, vols.shape[0], rows_per_piece):
            piece = vols[lo:lo + rows_per_piece].to(torch.float32)
            if not torch.equal(torch.nan_to_num(piece.to(torch.float16).to(torch.float32), nan=0.0), torch.nan_to_num(piece, nan=0.0)):
                return False
        return True

    def __init__(self, emb, meta, volumes, device, rank=0, world=1, group=None, backend=HipSearchBackend, host_group=None, half_store=None):
        """emb [N+1,64] float32 (unit rows), meta [N+1,7] int32, volumes [S,64,64,64] float32 -- host or device tensors /
        numpy arrays of the FULL database; this rank keeps its embedding shard and replicas of meta/volumes.

        With world > 1 (or an initialised process group) the constructor is COLLECTIVE: unless ``host_group`` (a gloo group over the ranks of
        ``group``) is passed, it creates the gloo twin of ``group`` for the query-count check, and ``dist.new_group`` must be entered by every
        rank of the default group -- construct the database on all ranks at the same point of the program.

        ``half_store``: keep the replicated voxel store as float16, the precision the reference holds scenes in (dataset/scene.py:61,71): half the HBM
        (1 M patches: 8.2 GB instead of 16.4 GB per GPU) and half the bytes the patch gather reads, the same gathered bits.  None (default): float16 whenever
        every voxel survives the round trip (true for anything that came out of the reference's float16 scenes), float32 otherwise; True: float16 or a
        ValueError; False: float32."""
        emb = torch.as_tensor(emb)
        self.count_check = None
        if world > 1:
            self.count_check = QueryCountCheck(host_group if host_group is not None else make_host_group(group))
        self.backend = backend
        self.collective_events = None    # bench.py: a list to collect (start, end) event pairs of the two collectives
        self.n_rows = emb.shape[0]
        self.dim = emb.shape[1]
        self.rank, self.world, self.group = rank, world, group
        self.force_collectives = False          # dev/test: run the all-gather + merge protocol even with one rank
        self.lo, self.hi = shard_bounds(self.n_rows, rank, world)
        self.device = torch.device(device)
        shard = emb[self.lo:self.hi].to(self.device, torch.float32).contiguous()
        self.emb_packed = backend.pack(shard)
        self.scan_algo = backend.choose_scan(shard, self.emb_packed, self.lo) if hasattr(backend, 'choose_scan') and self.emb_packed is not None else 0
        self._scan_kw = {'algo': self.scan_algo} if self.scan_algo else {}
        self.meta = torch.as_tensor(meta).to(self.device, torch.int32).contiguous()
        vols = torch.as_tensor(volumes)
        if half_store is None:
            half_store = vols.dtype == torch.float16 or self._fits_float16(vols)
        if half_store and vols.dtype != torch.float16:
            # piece by piece FROM WHERE THE STORE IS (copying the whole fp32 store to the device first costs 16.4 + 8.2 GB at 1 M patches -- more than a GPU
            # that would hold the narrowed store may have left)
            narrowed = torch.empty(vols.shape, dtype=torch.float16, device=self.device)
            for lo in range(0, vols.shape[0], 1024):
                piece = vols[lo:lo + 1024].to(self.device, torch.float32)
                half = piece.to(torch.float16)
                if not torch.equal(torch.nan_to_num(half.to(torch.float32), nan=0.0), torch.nan_to_num(piece, nan=0.0)) or not torch.equal(half.isnan(), piece.isnan()):
                    raise ValueError('half_store=True: the voxel store holds values float16 cannot represent (the reference\'s scenes are float16, '
                                     'dataset/scene.py:61,71); keep the float32 store for this database')
                narrowed[lo:lo + 1024] = half
            vols = narrowed
        self.half_store = bool(half_store)
        self.volumes = vols.to(self.device, torch.float16 if half_store else torch.float32).contiguous()
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
        """Exact squared-L2 top-k2 of q against this rank's shard, global row ids -> (dist, idx)."""
        return self.backend.topk(q.contiguous(), self.emb_packed, self.hi - self.lo, self.lo, k2, **self._scan_kw)

    def local_topk_keys(self, q, k2):
        """... as packed 64-bit keys [Q, k2] int64 (what the ranks exchange)."""
        return self.backend.topk_keys(q.contiguous(), self.emb_packed, self.hi - self.lo, self.lo, k2, **self._scan_kw)

    def search(self, q, k2):
        """Top-k2 over the whole database for this rank's queries.  One process: a single scan.  W processes:
        all-gather(queries) -> shard scans -> ONE all-to-all of the packed candidate keys -> merge (sharded_search).
        The collectives are issued from the caller's stream (torch runs them on RCCL's own stream, ordered by events), so the
        U-Net backbone the engine forked onto its side stream keeps the GPU busy while they are in flight."""
        if self.world == 1 and not self.force_collectives:
            return self.local_topk(q, k2)
        if self.count_check is None:                               # one rank with the protocol forced (bench.py --force-collectives, tests)
            self.count_check = QueryCountCheck(make_host_group(self.group))
        return sharded_search(q, lambda qa: self.local_topk_keys(qa, k2), self.backend.merge_keys, k2, self.group, self.collective_events,
                              self.count_check)

    def check(self):
        """Examine the query-count exchange of the last search now (it is otherwise examined at the next search); raises ValueError on a mismatch."""
        if self.count_check is not None:
            self.count_check.flush()

    def retrieve(self, q, K, query_scene=None, query_keep=None):
        """flann_knn_worker semantics (util/retrieval.py:92-100): top-2K, same-scene demotion, keep K.
        ``query_keep`` [Q] bool: False = patch dropped by the query-side occupancy filter (no neighbours, trunc fill).
        Returns (meta [Q,K,7] int32, dist [Q,K], idx [Q,K])."""
        dist, idx = self.search(q, 2 * K)
        return self.backend.demote(dist, idx, self.meta, query_scene, K, query_keep)
