"""CPU, world_size 2 over gloo: the sharded-database search exactly as the product runs it -- ``PatchDatabase.search`` /
``retrieve`` -> ``sharded_search``: all-gather of queries, per-shard top-2K as packed 64-bit keys with global row ids, ONE
all-gather of the keys, per-rank merge, same-scene demotion -- gives every rank exactly the single-process result.

The scan / merge / demotion kernels are replaced AT THE BACKEND SEAM (rfuse.database.HipSearchBackend) by numpy stand-ins
with the same contracts (test infrastructure, defined here), so what runs on the CPU is the product's own wiring that runs
over RCCL on the GPUs.  tests/test_multigpu.py runs the same thing with the HIP backend on 2 GPUs when they exist."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = Path(__file__).resolve().parents[1]
NONE = np.uint64(0xFFFFFFFFFFFFFFFF)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


class NumpySearchBackend:
    """rf_l2_topk_keys / rf_topk_merge_keys / rf_demote_same_scene contracts in numpy (float64 distances rounded to fp32)."""

    @staticmethod
    def pack(emb_shard):
        return emb_shard

    @staticmethod
    def _keys(q, emb, row_base, k2):
        q, emb = q.numpy(), emb.numpy()
        keys = np.full((q.shape[0], k2), NONE, dtype=np.uint64)
        if emb.shape[0]:
            d = ((q[:, None, :].astype(np.float64) - emb[None].astype(np.float64)) ** 2).sum(-1).astype(np.float32)
            all_keys = (d.view(np.uint32).astype(np.uint64) << np.uint64(32)) | (np.arange(emb.shape[0], dtype=np.uint64) + np.uint64(row_base))[None]
            all_keys.sort(axis=1)
            m = min(k2, emb.shape[0])
            keys[:, :m] = all_keys[:, :m]
        return keys

    @classmethod
    def topk_keys(cls, q, packed, n, row_base, k2):
        return torch.from_numpy(cls._keys(q, packed, row_base, k2).view(np.int64))

    @staticmethod
    def _unpack(keys):
        none = keys == NONE
        d = (keys >> np.uint64(32)).astype(np.uint32).view(np.float32)
        i = (keys & np.uint64(0xFFFFFFFF)).astype(np.int64)
        return torch.from_numpy(np.where(none, np.float32(np.inf), d)), torch.from_numpy(np.where(none, -1, i))

    @classmethod
    def topk(cls, q, packed, n, row_base, k2):
        return cls._unpack(cls._keys(q, packed, row_base, k2))

    @classmethod
    def merge_keys(cls, key_parts):
        k = key_parts.numpy().view(np.uint64)                           # [parts, nq, k2]
        parts, nq, k2 = k.shape
        flat = np.sort(k.transpose(1, 0, 2).reshape(nq, parts * k2), axis=1)[:, :k2]
        return cls._unpack(flat)

    @staticmethod
    def demote(dist_, idx, meta, query_scene, K, query_keep):
        sys.path.insert(0, str(REPO))
        from oracle import refpath
        idx_n, meta_n = idx.numpy(), meta.numpy()
        rows = np.concatenate([np.where(idx_n[..., None] >= 0, meta_n[np.maximum(idx_n, 0)], np.array([-1, 0, 16, 0, 16, 0, 16])).astype(np.float32),
                               dist_.numpy()[..., None]], axis=-1)
        qs = query_scene.numpy() if query_scene is not None else np.full(idx_n.shape[0], -1)
        out = refpath.demote_same_scene(rows, qs, K)
        return torch.from_numpy(out[..., :7].astype(np.int32)), torch.from_numpy(out[..., 7].copy()), None


def _problem(n_rows, world, nq_local):
    rng = np.random.default_rng(0)
    emb = rng.standard_normal((n_rows, 64)).astype(np.float32)
    if n_rows > 12:
        emb[11] = emb[3]                                     # a tie inside a shard ... and one across shards:
        emb[n_rows - 2] = emb[3]
    q_all = rng.standard_normal((world * nq_local, 64)).astype(np.float32)
    if n_rows > 12:
        q_all[0] = emb[3]
    meta = np.concatenate([np.repeat(np.arange((n_rows + 63) // 64), 64)[:n_rows, None], np.tile([0, 16, 0, 16, 0, 16], (n_rows, 1))], 1).astype(np.int32)
    qscene = np.where(np.arange(world * nq_local) % 3 == 0, 0, -1).astype(np.int32)     # every third query comes from scene 0
    return emb, meta, q_all, qscene


def _worker(rank, world, port, n_rows, nq_local, K, out_dir, unequal):
    for p in (str(REPO), str(REPO / 'retrieval-fuse_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from rfuse.database import PatchDatabase
    emb, meta, q_all, qscene = _problem(n_rows, world, nq_local)
    db = PatchDatabase(emb, meta, np.zeros((1, 64, 64, 64), np.float32), 'cpu', rank, world, None, backend=NumpySearchBackend)
    sl = slice(rank * nq_local, (rank + 1) * nq_local)
    q_local = torch.from_numpy(q_all[sl])
    if unequal:
        try:
            if unequal == 2:                                   # a first call with equal counts, then rank 0 passes a NEW count and rank 1 one it has seen
                db.search(q_local, 2 * K)
                db.search(q_local[: nq_local - 1] if rank == 0 else q_local, 2 * K)
            else:
                db.search(q_local[: nq_local - rank], 2 * K)
            msg = 'no error'
        except ValueError as e:
            msg = str(e)
        Path(out_dir, f'rank{rank}.txt').write_text(msg)
    else:
        d, i = db.search(q_local, 2 * K)
        m, dd, _ = db.retrieve(q_local, K, torch.from_numpy(qscene[sl]))
        np.savez(Path(out_dir) / f'rank{rank}.npz', d=d.numpy(), i=i.numpy(), meta=m.numpy(), dk=dd.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize('world,n_rows,nq_local,K', [(2, 1001, 64, 4), (2, 13, 5, 4), (2, 1, 3, 4), (4, 1003, 16, 8), (8, 1003, 8, 8), (8, 5, 2, 4)])
def test_sharded_search_equals_single_process(tmp_path, world, n_rows, nq_local, K):
    """(1 row, 2 ranks): rank 1's shard is EMPTY -- it contributes all-NONE lists.  (4 ranks, 1003 rows, K = 8): shards of unequal size
    (n % W != 0), the C4 list length.  (8 ranks): the node size of BASELINE configs[2] -- 1003 % 8 = 3 rows go to the low ranks; 5 rows on 8 ranks
    leave three shards empty."""
    mp.spawn(_worker, args=(world, _free_port(), n_rows, nq_local, K, str(tmp_path), False), nprocs=world, join=True)
    sys.path.insert(0, str(REPO / 'retrieval-fuse_amd'))
    emb, meta, q_all, qscene = _problem(n_rows, world, nq_local)
    B = NumpySearchBackend
    d_ref, i_ref = B.topk(torch.from_numpy(q_all), torch.from_numpy(emb), n_rows, 0, 2 * K)
    m_ref, dk_ref, _ = B.demote(d_ref, i_ref, torch.from_numpy(meta), torch.from_numpy(qscene), K, None)
    for rank in range(world):
        z = np.load(tmp_path / f'rank{rank}.npz')
        sl = slice(rank * nq_local, (rank + 1) * nq_local)
        np.testing.assert_array_equal(z['i'], i_ref.numpy()[sl])
        np.testing.assert_array_equal(z['d'], d_ref.numpy()[sl])
        np.testing.assert_array_equal(z['meta'], m_ref.numpy()[sl])
        np.testing.assert_array_equal(z['dk'], dk_ref.numpy()[sl])
    z0 = np.load(tmp_path / 'rank0.npz')
    if n_rows > 100:
        assert z0['i'][0, :3].tolist() == [3, 11, n_rows - 2], 'ties must resolve to the lower global row id across shards'
        assert (z0['meta'][0, :, 0] != 0).all(), 'query 0 comes from scene 0: its same-scene neighbours are demoted'
    if n_rows == 1:
        assert (z0['i'][:, 0] == 0).all() and (z0['i'][:, 1:] == -1).all()


@pytest.mark.parametrize('mode', [1, 2])
def test_unequal_query_counts_are_refused(tmp_path, mode):
    """mode 2: the counts differ only on a LATER call, and only one rank's count is new (a per-rank cache of checked counts would let the
    other rank skip the check and walk into mismatched collectives)"""
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), 100, 6, 4, str(tmp_path), mode), nprocs=world, join=True)
    for rank in range(world):
        assert 'same number of queries' in (tmp_path / f'rank{rank}.txt').read_text()


def _deferred_worker(rank, world, port, out_dir):
    for p in (str(REPO), str(REPO / 'retrieval-fuse_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from rfuse.database import QueryCountCheck, make_host_group
    chk = QueryCountCheck(make_host_group(None))
    log = []
    try:
        chk.post(6)                                  # first use of 6 everywhere: checked at once, equal
        log.append('a')
        chk.post(6)                                  # seen: not waited for
        log.append('b' if chk.pending is not None else 'b-waited')
        chk.post(5 if rank == 0 else 6)              # rank 0: a new count -> waits, sees (5, 6, ...) and raises; the others defer
        log.append('c')
        chk.post(6)                                  # ... and raise here, one call later, when they examine the previous exchange
        log.append('d')
    except ValueError as e:
        log.append('refused: ' + str(e))
    Path(out_dir, f'rank{rank}.txt').write_text('|'.join(log))
    dist.barrier()
    dist.destroy_process_group()


def test_query_count_check_deferred_mode(tmp_path):
    """The device data path's mode (collectives only enqueued): a seen count is not waited for, the rank with the new count refuses at once and
    its peers one call later; every rank posts the same sequence of host exchanges whatever it has seen (no rank skips one the other enters)."""
    world = 3
    mp.spawn(_deferred_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [(tmp_path / f'rank{k}.txt').read_text() for k in range(world)]
    assert r[0].startswith('a|b|refused') and 'same number of queries' in r[0] and '[5, 6, 6]' in r[0]
    for k in (1, 2):
        assert r[k].startswith('a|b|c|refused') and '[5, 6, 6]' in r[k]


def _both_seen_worker(rank, world, port, out_dir):
    for p in (str(REPO), str(REPO / 'retrieval-fuse_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from rfuse.database import QueryCountCheck, make_host_group
    chk = QueryCountCheck(make_host_group(None))
    log = []
    try:
        for nq in (6, 5, 5):                         # every rank has now used both counts; the last call was 5
            chk.post(nq)
        log.append('warm' if chk.pending is not None else 'warm-waited')
        chk.post(6 if rank == 0 else 5)              # rank 0 goes back to 6 (a SEEN count, but not its previous one): must wait and refuse at once
        log.append('posted')
        chk.flush()                                  # PatchDatabase.check(): the ranks whose count did not change learn of it here, before any host sync
        log.append('flushed')
    except ValueError as e:
        log.append('refused: ' + str(e))
    Path(out_dir, f'rank{rank}.txt').write_text('|'.join(log))
    dist.barrier()
    dist.destroy_process_group()


def test_query_count_check_both_counts_seen_but_different(tmp_path):
    """with 'wait only for a never-seen count' two ranks that had both used 6 and 5 walked into mismatched device collectives.  Now a rank
    whose count differs from its own previous call waits and refuses before enqueuing; its peers (count unchanged, deferred) are told by check()."""
    world = 2
    mp.spawn(_both_seen_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [(tmp_path / f'rank{k}.txt').read_text() for k in range(world)]
    assert r[0].startswith('warm|refused') and '[6, 5]' in r[0], r
    assert r[1].startswith('warm|posted|refused') and '[6, 5]' in r[1], r


def test_shard_bounds_cover_rows_exactly():
    sys.path.insert(0, str(REPO / 'retrieval-fuse_amd'))
    from rfuse.database import shard_bounds
    for n, w in ((50_001, 8), (7, 8), (1_000_001, 4), (64, 1)):
        b = [shard_bounds(n, r, w) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(b[r][1] == b[r + 1][0] for r in range(w - 1))
        assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1
