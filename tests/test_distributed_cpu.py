"""CPU, world_size 2 over gloo: the sharded-database search protocol (rfuse.database.allgather_merge) -- all-gather
of queries, per-shard top-2K with global row ids, all-gather of the candidate lists, per-rank merge -- gives every
rank exactly the single-process result.  The local scan / merge are numpy stand-ins with the SAME contract as the
HIP kernels (rf_l2_topk / rf_topk_merge), so what is tested here is the wiring that runs over RCCL on the GPUs."""
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


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _np_topk(q, emb, row_base, k2):
    d = ((q[:, None, :].astype(np.float64) - emb[None].astype(np.float64)) ** 2).sum(-1)
    order = np.argsort(d, axis=1, kind='stable')[:, :k2]
    dist_ = np.take_along_axis(d, order, 1).astype(np.float32)
    idx = (order + row_base).astype(np.int64)
    if emb.shape[0] < k2:                                   # contract: missing candidates are (inf, -1)
        pad = k2 - emb.shape[0]
        dist_ = np.concatenate([dist_, np.full((q.shape[0], pad), np.inf, np.float32)], 1)
        idx = np.concatenate([idx, np.full((q.shape[0], pad), -1, np.int64)], 1)
    return torch.from_numpy(dist_), torch.from_numpy(idx)


def _np_merge(d_parts, i_parts):
    parts, nq, k2 = d_parts.shape
    d = d_parts.permute(1, 0, 2).reshape(nq, parts * k2).numpy()
    i = i_parts.permute(1, 0, 2).reshape(nq, parts * k2).numpy()
    key_i = np.where(i < 0, np.iinfo(np.int64).max, i)
    order = np.lexsort((key_i, d), axis=1)[:, :k2]          # by (dist, idx), as the 64-bit key compare does
    return torch.from_numpy(np.take_along_axis(d, order, 1)), torch.from_numpy(np.take_along_axis(i, order, 1))


def _worker(rank, world, port, n_rows, nq_local, k2, out_dir):
    for p in (str(REPO), str(REPO / 'retrieval-fuse_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from rfuse.database import allgather_merge, shard_bounds
    rng = np.random.default_rng(0)
    emb = rng.standard_normal((n_rows, 64)).astype(np.float32)
    emb[11] = emb[3]                                         # a tie across... the same shard; and one across shards:
    emb[n_rows - 2] = emb[3]
    q_all = rng.standard_normal((world * nq_local, 64)).astype(np.float32)
    q_all[0] = emb[3]
    lo, hi = shard_bounds(n_rows, rank, world)
    q_local = torch.from_numpy(q_all[rank * nq_local:(rank + 1) * nq_local])
    d, i = allgather_merge(q_local, lambda qa: _np_topk(qa.numpy(), emb[lo:hi], lo, k2), _np_merge, k2)
    np.savez(Path(out_dir) / f'rank{rank}.npz', d=d.numpy(), i=i.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize('n_rows,nq_local,k2', [(1001, 64, 8), (13, 5, 8)])
def test_sharded_search_world2_equals_single_process(tmp_path, n_rows, nq_local, k2):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_rows, nq_local, k2, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(0)
    emb = rng.standard_normal((n_rows, 64)).astype(np.float32)
    emb[11] = emb[3]
    emb[n_rows - 2] = emb[3]
    q_all = rng.standard_normal((world * nq_local, 64)).astype(np.float32)
    q_all[0] = emb[3]
    d_ref, i_ref = _np_topk(q_all, emb, 0, k2)
    for rank in range(world):
        z = np.load(tmp_path / f'rank{rank}.npz')
        sl = slice(rank * nq_local, (rank + 1) * nq_local)
        np.testing.assert_array_equal(z['i'], i_ref.numpy()[sl])
        np.testing.assert_array_equal(z['d'], d_ref.numpy()[sl])
    z0 = np.load(tmp_path / 'rank0.npz')
    if n_rows > 100:
        assert z0['i'][0, :3].tolist() == [3, 11, n_rows - 2], 'ties must resolve to the lower global row id across shards'


def test_shard_bounds_cover_rows_exactly():
    sys.path.insert(0, str(REPO / 'retrieval-fuse_amd'))
    from rfuse.database import shard_bounds
    for n, w in ((50_001, 8), (7, 8), (1_000_001, 4), (64, 1)):
        b = [shard_bounds(n, r, w) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(b[r][1] == b[r + 1][0] for r in range(w - 1))
        assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1
