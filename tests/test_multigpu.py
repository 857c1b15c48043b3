"""RCCL runs of the sharded search with the HIP backend (the gloo test covers the same wiring on CPU):
  * one rank: the whole protocol -- process group on RCCL, all-gather of queries, shard scan to packed keys, all-gather of the
    keys, key merge, demotion -- inside the engine, against the plain single-scan engine.  Runs on the 1-GPU test box.
  * two ranks (skipped when fewer than 2 GPUs are visible): DB sharded 2 ways, every rank must get the single-scan result."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, n_patches, B, backend='nccl'):
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
    for p in (str(REPO), str(REPO / 'retrieval-fuse_amd'), str(REPO / 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    # 'nccl' (RCCL): one GPU per rank.  'gloo': every rank on cuda:0 -- the collectives stage the device tensors through the host, the shard scans, the key
    # merge and the demotion are the HIP kernels: the sharded protocol with W > 1 on ONE GPU (RCCL refuses two ranks on one device)
    dev = torch.device('cuda', rank if backend == 'nccl' else 0)
    torch.cuda.set_device(dev)
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    import helpers
    from rfuse import configs, synthetic
    from rfuse.database import PatchDatabase
    from rfuse.engine import RefinementEngine
    cfg = configs.get_config('C3')
    db = synthetic.make_database(41, cfg, n_patches)
    sharded = PatchDatabase(db['emb'], db['meta'], db['volumes'], dev, rank, world)
    sharded.force_collectives = True
    single = PatchDatabase(db['emb'], db['meta'], db['volumes'], dev, 0, 1)
    raws = np.stack([synthetic.make_chunk(9000 + rank * B + b, cfg)['input_raw'] for b in range(B)])
    qs = torch.from_numpy(np.where(np.arange(B * 64) % 5 == 0, 1, -1).astype(np.int32)).to(dev)
    outs = {}
    for name, pdb in (('sharded', sharded), ('single', single)):
        eng = RefinementEngine(cfg, dev, pdb)
        eng.load_state_dicts({n: helpers.seeded_sd({k: tuple(v.shape) for k, v in m.state_dict().items()}, 60 + i)
                              for i, (n, m) in enumerate(eng.modules().items())})
        x = torch.from_numpy(raws).to(dev)
        q = eng.embed_queries(x)
        d, i = pdb.search(q, 2 * cfg['K'])
        meta, _, _ = pdb.retrieve(q, cfg['K'], qs)
        df = eng.refine(x, qs)
        torch.cuda.synchronize()
        outs[name] = (d.cpu(), i.cpu(), meta.cpu(), df.cpu())
    ok = all(torch.equal(a, b) for a, b in zip(outs['sharded'], outs['single']))
    Path(out_dir, f'rank{rank}.txt').write_text('ok' if ok else 'MISMATCH')
    dist.destroy_process_group()


def _run(world, tmp_path, n_patches=64 * 30 + 7, B=2, backend='nccl'):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), n_patches, B, backend), nprocs=world, join=True)
    for rank in range(world):
        assert (tmp_path / f'rank{rank}.txt').read_text() == 'ok'


def test_rccl_protocol_with_one_rank(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip('no GPU visible')
    _run(1, tmp_path)


@pytest.mark.parametrize('world,n_patches', [(2, 64 * 30 + 7), (4, 64 * 50 + 3), (3, 5)])
def test_sharded_search_several_ranks_on_one_gpu(tmp_path, world, n_patches):
    """W ranks sharing cuda:0 over gloo: every rank's HIP shard scan (packed keys), the all-to-all of the keys, rf_topk_merge_keys and the demotion against the
    single-scan engine, end to end through refine() -- the multi-rank protocol on the device kernels without needing W GPUs (5 rows on 3 ranks: an empty
    shard and lists shorter than 2K)."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU visible')
    _run(world, tmp_path, n_patches=n_patches, backend='gloo')


def test_rccl_sharded_search_two_ranks(tmp_path):
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs')
    _run(2, tmp_path)
