"""bench.py -- chunks/sec of the online retrieve -> attend -> refine path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--config C2|C3|C4|C5] [--db PATCHES]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the whole hot path over a batch of B synthetic 64^3 chunks per GPU:
query windows -> query encoder -> exact L2 top-2K over the patch database (sharded N ways, RCCL all-gather of the
per-shard candidates when N > 1) -> same-scene demotion -> patch gather -> retrieval backbone (K*64 patches per chunk)
|| U-Net backbone -> patch attention -> decoder -> df.  Inputs, weights and the database are resident in HBM before
the timed region; weights are random-init (torch.manual_seed(0)), data synthetic (no datasets / checkpoints here).

Workload at N=1 (default): BASELINE.json configs[1] -- ShapeNetV2 super-res 008->064, synthetic batch, K=4, DB=50k patches.
Prints ONE JSON line (rank 0) with the bench contract fields plus `roofline` (dominant kernel, live HIP-event timing)
and `cpu_baseline` (the oracle -- a CPU port of the reference path -- timed on the host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
for p in (str(REPO), str(REPO / 'retrieval-fuse_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
# The engine runs the U-Net backbone on a second HIP stream next to the retrieval path.  ROCm maps streams onto
# GPU_MAX_HW_QUEUES (default 4) hardware queues; once RCCL has created its own streams the two compute streams can land
# on the same queue and serialise (measured: -5.5 % with a process group initialised, back to par with 8 queues).
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

import numpy as np
import torch
import torch.distributed as dist

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X dense fp32 MFMA (= fp32 vector) peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=32, help='64^3 chunks per GPU per step')
    ap.add_argument('--config', default='C2')
    ap.add_argument('--db', type=int, default=0, help='database patches (default: the config\'s)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--feature-cache', action='store_true', help='also time the optional cached-retrieval-features serving mode (reported separately)')
    ap.add_argument('--cpu-chunks', type=int, default=0, help='chunks for the CPU baseline sample (0 = sized to ~15 s)')
    ap.add_argument('--force-collectives', action='store_true', help='dev: with one rank under torch.distributed.run, still run the all-gather + merge protocol')
    return ap.parse_args()


def synthetic_database(cfg, n_patches, device, seed=1234):
    """Seeded synthetic DB built on the device: unit Gaussian embeddings (DB-side encoder = 'next' row N1), reference row
    semantics for meta (util/retrieval.py:32,39-45 + sentinel), U(0,trunc) fp16-rounded scene chunks as the voxel store."""
    from rfuse import configs, synthetic
    _, trunc_t = configs.truncations(cfg)
    g = torch.Generator(device=device).manual_seed(seed)
    emb = torch.randn(n_patches + 1, 64, generator=g, device=device, dtype=torch.float32)
    emb = emb / emb.norm(dim=1, keepdim=True).clamp_min(1e-12)
    meta = torch.from_numpy(synthetic.make_database(seed, cfg, n_patches, with_volumes=False)['meta'])
    n_scenes = (n_patches + 63) // 64
    vols = (torch.rand(n_scenes, 64, 64, 64, generator=g, device=device) * trunc_t).half().float()
    return emb, meta, vols


def cpu_baseline(cfg, eng_state, db_host, raws, target_s=15.0, n_chunks=0):
    """The oracle (CPU port of the reference path, oracle/refpath.py) on the host cores: same stages as one GPU step.
    kNN = torch.cdist + topk fp32; everything else the pinned oracle."""
    from oracle import refpath
    from rfuse import configs, synthetic
    trunc_i, trunc_t = configs.truncations(cfg)
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    d, K = cfg['dataset_train'], cfg['K']
    noise_gen = torch.Generator().manual_seed(0)

    def one_chunk(raw):
        with torch.no_grad():
            q = refpath.embed_queries(refpath.extract_query_windows(raw, cfg, trunc_i), eng_state['fenc_input'], cfg).numpy()
            idx, dd = refpath.knn_cdist_f32(q, db_host['emb'], 2 * K)
            mapping = refpath.demote_same_scene(refpath.mapping_rows(idx, dd, db_host['meta']), np.full(q.shape[0], -1), K)
            retr = refpath.compose_retrieval(mapping, db_host['volumes'], K, trunc_t)[None]
            retr = ((retr - np.float32(d['target_mean'])) / np.float32(d['target_std'])).astype(np.float32)
            x_in = synthetic.normalise_input(cfg, raw)[None, None]
            noise = None
            if cfg['attn_retrieval_mode']:
                noise = -torch.empty(cfg['attn_num_patch'] ** 3, K).exponential_(generator=noise_gen).log()
            return refpath.forward_full(eng_state, cfg, torch.from_numpy(x_in), torch.from_numpy(retr), trunc_t, noise)

    # PyTorch-CPU on these small volumes is fastest well below the core count (oversubscription): probe a few thread
    # counts with one chunk each and keep the best -- the baseline should be the CPU path at its best, not at its worst.
    torch.set_num_threads(min(8, cores))
    one_chunk(raws[0])                                        # warm-up
    best_t, best_threads = None, None
    for threads in sorted({min(c, cores) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        one_chunk(raws[1 % len(raws)])
        t1 = time.perf_counter() - t0
        if best_t is None or t1 < best_t:
            best_t, best_threads = t1, threads
    torch.set_num_threads(best_threads)
    n = n_chunks or int(max(4, min(160, round(target_s / max(best_t, 1e-3)))))
    t0 = time.perf_counter()
    for i in range(n):
        one_chunk(raws[i % len(raws)])
    el = time.perf_counter() - t0
    return {'value': n / el, 'unit': 'chunks/s', 'cores': best_threads, 'kind': 'port',
            'sample': '%d chunks of the same workload, full path (query embed + exact kNN via torch.cdist + compose + '
                      'networks), oracle/refpath.py on torch-CPU fp32, %d threads (best of 8/16/32/64 on a %d-core host), %.1f s'
                      % (n, best_threads, cores, el)}


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node %d for --gpus %d' % (args.gpus, args.gpus))
    assert torch.cuda.is_available(), 'bench.py needs the GPU (no CPU fallback for the hot path)'
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    knobs = sorted(k for k in os.environ if k.startswith('RFUSE_') and k != 'RFUSE_LIB')
    assert not knobs, 'refusing to measure with developer switches set: %s' % knobs
    force_dist = world == 1 and args.force_collectives and 'RANK' in os.environ        # dev: run the whole RCCL protocol with one rank
    if world > 1 or force_dist:
        dist.init_process_group('nccl', device_id=device)      # RCCL on ROCm

    from rfuse import configs, ops, synthetic
    from rfuse.database import PatchDatabase
    from rfuse.engine import RefinementEngine

    cfg = configs.get_config(args.config)
    n_patches = args.db or cfg['db_patches']
    B, K = args.batch, cfg['K']

    torch.manual_seed(0)
    emb, meta, vols = synthetic_database(cfg, n_patches, device)
    database = PatchDatabase(emb, meta, vols, device, rank, world)
    database.force_collectives = force_dist
    eng = RefinementEngine(cfg, device, database)
    # every rank refines its own B chunks (chunk-parallel replicas); inputs resident in HBM
    raws = np.stack([synthetic.make_chunk(10_000 + rank * B + b, cfg)['input_raw'] for b in range(B)])
    raw_dev = torch.from_numpy(raws).to(device)

    # dominant kernel: the 96->56 (nf=16) first conv of the retrieval backbone's last decoder, n = B*K*64 patches of 8^3
    # (32 skip channels at 8^3 + 64 channels upsampled from 4^3: the parity-split kernel rf_conv3d_up_k3_gn_relu)
    nf = cfg['nf']
    dom_cin, dom_cout = 6 * nf, (6 * nf + nf) // 2
    ops.conv_event_filter = lambda cin, cout, edge, n: cin == dom_cin and cout == dom_cout and edge == 8 and n == B * K * 64

    def barrier():
        if world > 1 or force_dist:
            dist.barrier()

    for _ in range(args.warmup):
        eng.refine(raw_dev)
    torch.cuda.synchronize()
    ops.conv_events.clear()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        df = eng.refine(raw_dev)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1 or force_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    assert torch.isfinite(df).all()

    ev = list(ops.conv_events)
    ops.conv_event_filter = None
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b, _ in ev])) if ev else float('nan')
    kern_flops = ev[0][2] if ev else 0.0
    achieved = kern_flops / (kern_ms * 1e-3) / 1e12 if ev else float('nan')

    if rank == 0:
        value = world * B * args.steps / elapsed
        # HBM traffic of the dominant kernel: PMC-measured (separate rocprofv3 --pmc passes, FETCH_SIZE x2 gfx950 correction
        # + WRITE_SIZE, committed under profiles/), scaled per patch to this launch; None when no summary is committed
        traffic = None
        pmc_file = REPO / 'profiles' / 'r01_dominant_kernel.json'
        if pmc_file.exists() and args.config in ('C1', 'C2', 'C3'):
            traffic = json.loads(pmc_file.read_text())['traffic_bytes_per_patch'] * B * K * 64
        out = {
            'metric': '64^3 TSDF chunks/sec (retrieve+attend+refine)', 'value': value, 'unit': 'chunks/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s: %s super-res/recon ->064, synthetic chunks, K=%d, DB=%d patches (exact L2 top-%d), '
                                   'random-init weights' % (args.config, cfg['dataset_train']['dataset_name'], K, n_patches, 2 * K),
                       'chunks_per_gpu_per_step': B, 'db_patches': n_patches,
                       'parallelism': 'chunk-parallel replicas x%d, DB embedding matrix sharded %d-way + RCCL all-gather of top-2K' % (world, world)},
            'roofline': {'bound': 'mfma',
                         'kernel': 'k_conv3_up<8^3 box, 8 waves = 8 output parities, MB4, NB4> (retrieval backbone decoder conv %d+%d->%d @8^3: '
                                   '%d skip channels x 27 taps + %d upsampled channels x 8 pre-summed low-res taps, z-border padding taps '
                                   'left out, %d patches)' % (2 * nf, 4 * nf, dom_cout, 2 * nf, 4 * nf, B * K * 64),
                         'achieved': achieved, 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / FP32_MFMA_PEAK_TFLOPS,
                         'traffic': traffic, 'traffic_unit': 'bytes/launch (PMC)', 'launch_ms': kern_ms,
                         'flops_per_launch': kern_flops,          # multiply-adds ISSUED (decoder form minus the skipped zero-padding taps)
                         'decoder_form_flops_per_launch': 2.0 * (27 * 2 * nf + 8 * 4 * nf) * dom_cout * 512 * B * K * 64,
                         'direct_form_flops_per_launch': 2.0 * 27 * dom_cin * dom_cout * 512 * B * K * 64,
                         'algorithmic_bytes_per_launch': 4.0 * B * K * 64 * (2 * nf * 512 + 4 * nf * 64 + dom_cout * 512)},
        }
        if args.feature_cache and world == 1 and n_patches <= 200_000:
            # Reported separately, NEVER as `value`: optional serving mode that fetches per-database-row retrieval-backbone
            # features (query independent) from a 32 KB/row HBM cache instead of recomputing them (skips 87 % of the FLOPs).
            database.build_feature_cache(eng.retrieval_backbone, cfg)
            for _ in range(2):
                eng.refine(raw_dev, use_feature_cache=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                eng.refine(raw_dev, use_feature_cache=True)
            torch.cuda.synchronize()
            out['feature_cache_mode'] = {'value': B * args.steps / (time.perf_counter() - t1), 'unit': 'chunks/s',
                                         'note': 'optional mode, work skipped (retrieval backbone replaced by a gather of cached '
                                                 'features); not the headline metric', 'cache_bytes': int(database.feature_cache.numel() * 4)}
        if world == 1 and not args.no_cpu_baseline:
            state = {n: {k: v.detach().cpu() for k, v in m.state_dict().items()} for n, m in eng.modules().items()}
            db_host = {'emb': emb.cpu().numpy(), 'meta': meta.numpy(), 'volumes': vols.cpu().numpy()}
            out['cpu_baseline'] = cpu_baseline(cfg, state, db_host, raws, n_chunks=args.cpu_chunks)
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out))
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
