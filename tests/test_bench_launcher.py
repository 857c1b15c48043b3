"""bench.py's self-launcher: `python bench.py --gpus N` without WORLD_SIZE / RANK in the environment starts its own N ranks under
torch.distributed.run on 127.0.0.1; a process that already is a rank (the driver's torchrun command) does not re-launch.

CPU: the argument plumbing (`launch_plan`).  GPU: the whole route with one rank and the RCCL protocol forced -- `python bench.py --gpus 1
--force-collectives` from a clean environment must print exactly one JSON line on stdout with `rccl_ranks` and per-rank `collectives`."""
import importlib.util
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]


def _bench():
    spec = importlib.util.spec_from_file_location('bench_under_test', REPO / 'bench.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_launch_plan_plumbing():
    bench = _bench()
    argv = ['--gpus', '8', '--steps', '20', '--warmup', '2', '--config', 'C3']
    plan = bench.launch_plan(bench.parse(argv), argv, {})
    assert plan[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert plan[plan.index('--nproc-per-node') + 1] == '8'
    assert plan[plan.index('--master-addr') + 1] == '127.0.0.1'
    assert 0 < int(plan[plan.index('--master-port') + 1]) < 65536
    script = plan.index(str(REPO / 'bench.py'))
    assert plan[script + 1:] == argv                               # the ranks see the caller's own arguments, unchanged
    # a process that already is a rank never re-launches
    assert bench.launch_plan(bench.parse(argv), argv, {'WORLD_SIZE': '8', 'RANK': '3', 'LOCAL_RANK': '3'}) is None
    assert bench.launch_plan(bench.parse(argv), argv, {'RANK': '0'}) is None
    # the plain one-GPU run stays in this process; the forced protocol needs a rendezvous and takes the launcher with one rank
    assert bench.launch_plan(bench.parse([]), [], {}) is None
    one = bench.launch_plan(bench.parse(['--force-collectives']), ['--force-collectives'], {})
    assert one[one.index('--nproc-per-node') + 1] == '1' and one[-1] == '--force-collectives'
    with pytest.raises(SystemExit):
        bench.launch_plan(bench.parse(['--gpus', '0']), ['--gpus', '0'], {})


def test_rank_count_mismatch_is_refused():
    """--gpus 4 inside a 2-rank job is a launcher error, caught before any GPU work (runs without a GPU)"""
    env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, str(REPO / 'bench.py'), '--gpus', '4'], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and 'WORLD_SIZE is 2' in r.stderr and r.stdout.strip() == ''


@pytest.mark.gpu
def test_self_launched_ranks_print_one_json_line():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU visible')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT') and not k.startswith('RFUSE_')}
    r = subprocess.run([sys.executable, str(REPO / 'bench.py'), '--gpus', '1', '--force-collectives', '--steps', '4', '--warmup', '1', '--repeats', '0', '--batch', '4',
                        '--db', '4096', '--no-extras', '--no-cpu-baseline', '--resident-batches', '2'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 1 and out['rccl_ranks'] == 1 and out['steps'] == 4 and out['value'] > 0
    assert [c['rank'] for c in out['collectives']['per_rank_ms']] == [0]
    assert out['collectives']['all_to_all_keys_ms'] > 0 and out['collectives']['all_gather_queries_ms'] > 0


@pytest.mark.gpu
def test_two_self_launched_ranks_on_one_gpu():
    """The multi-rank bench path end to end on a one-GPU box: `python bench.py --gpus 2 --ranks-share-gpu` (both ranks on cuda:0, collectives over gloo -- RCCL
    refuses two ranks on one device) launches its ranks, shards the database two ways, runs the timed loops, the per-rank collective timings and the
    recall leg (a collective search), and prints exactly one JSON line with n_gpus = 2 and recall 1.0.  A functional run, not a scaling measurement."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU visible')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT') and not k.startswith('RFUSE_')}
    r = subprocess.run([sys.executable, str(REPO / 'bench.py'), '--gpus', '2', '--ranks-share-gpu', '--steps', '3', '--warmup', '1', '--repeats', '0', '--batch', '4',
                        '--db', '4099', '--no-cpu-baseline', '--resident-batches', '2'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['value'] > 0 and 'ranks_share_gpu' in out
    assert [c['rank'] for c in out['collectives']['per_rank_ms']] == [0, 1]
    assert out['recall_at_k']['recall@1'] == 1.0 and out['recall_at_k']['exact_order'] == 1.0


@pytest.mark.gpu
def test_the_north_star_multi_gpu_configuration_through_the_self_launcher():
    """BASELINE.json configs[2] -- 3DFront 008->064, 1 M patches sharded over the ranks -- is `python bench.py --gpus 8 --config C3 --db 1000000` (README, INTEGRATION).
    The same command shape on the one-GPU test box: two ranks sharing cuda:0 (collectives over gloo), C3, 200 k patches (100 k rows per shard: the f16-MFMA-filtered
    scan, as a 1 M database's 125 k-row shards take it), default batch; the line must name the workload and report recall 1.0 with the exact order."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU visible')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT') and not k.startswith('RFUSE_')}
    r = subprocess.run([sys.executable, str(REPO / 'bench.py'), '--gpus', '2', '--ranks-share-gpu', '--config', 'C3', '--db', '200000', '--steps', '2', '--warmup', '1',
                        '--repeats', '0', '--no-cpu-baseline', '--resident-batches', '2'], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['value'] > 0 and out['scaling'] == 'weak'
    assert 'C3' in out['config']['workload'] and out['config']['db_patches'] == 200000
    assert [c['rank'] for c in out['collectives']['per_rank_ms']] == [0, 1]
    assert out['recall_at_k']['recall@1'] == 1.0 and out['recall_at_k']['exact_order'] == 1.0
