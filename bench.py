"""bench.py -- chunks/sec of the online retrieve -> attend -> refine path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--config C2|C3|C4|C5] [--db PATCHES]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the whole hot path over a batch of B synthetic 64^3 chunks per GPU (steps are software-pipelined through
RefinementEngine.refine_stream: the front end of step i + 1 runs beside the back end of step i; `unpipelined` = the same steps one after the other):
query windows -> query encoder -> exact L2 top-2K over the patch database (sharded N ways, one RCCL all-gather of the
per-shard candidate keys when N > 1) -> same-scene demotion -> patch gather -> retrieval backbone (K*64 patches per chunk)
|| U-Net backbone -> patch attention -> decoder -> df.  Inputs, weights and the database are resident in HBM before
the timed region; weights are random-init (torch.manual_seed(0)), data synthetic (no datasets / checkpoints here).

Workload at N=1 (default): BASELINE.json configs[1] -- ShapeNetV2 super-res 008->064, synthetic batch, K=4, DB=50k patches.
Prints ONE JSON line (rank 0) with the bench contract fields plus
  roofline      the dominant kernel, timed live with HIP events on its launch stream inside the timed region
  cpu_baseline  the oracle (a CPU port of the reference path) timed on the host cores (N=1 only)
  recall_at_k   the exact search against a float64 brute force on a query subset of the bench batch (k = 1, 4, 8)
  parity_max_abs  df of chunks of the bench batch against the oracle (explicit Gumbel noise for the ShapeNet configs)
  kernels       per-entry-point time / achieved / peak of the five most expensive launches (HIP events, a separate serial pass)
  batch_sweep   chunks/s at B = 1, 8, 64 (B = 1 also replayed from a captured HIP graph)           (N=1 only)
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

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X dense fp32 MFMA (= packed-fp32 vector) peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0              # HBM3E, same guide
F16_MFMA_PEAK_TFLOPS = 2500.0      # dense f16/bf16 MFMA peak, same guide ("~2.5 PF dense"; 16x the fp32 MFMA rate)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=32, help='64^3 chunks per GPU per step')
    ap.add_argument('--config', default='C2')
    ap.add_argument('--db', type=int, default=0, help='database patches (default: the config\'s)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip recall / parity / kernel table / batch sweep (profiling runs)')
    ap.add_argument('--feature-cache', action='store_true', help='also time the optional cached-retrieval-features serving mode (reported separately)')
    ap.add_argument('--repeats', type=int, default=4, help='further blocks of K steps timed after the contract\'s K (reported as `blocks`: min / median; never `value`)')
    ap.add_argument('--cpu-chunks', type=int, default=0, help='chunks for the CPU baseline sample (0 = sized to ~15 s)')
    ap.add_argument('--kernels-top', type=int, default=8, help='rows of the serial per-kernel table')
    ap.add_argument('--force-collectives', action='store_true', help='dev: with one rank, still run the all-gather + all-to-all + merge protocol over RCCL')
    ap.add_argument('--half-store', action='store_true', help='insist on the float16 voxel store (the default already keeps it as float16 -- the reference\'s own scene precision -- whenever every voxel survives the round trip: same gathered bits, half the bytes)')
    ap.add_argument('--fp32-store', action='store_true', help='keep the replicated voxel store as float32')
    ap.add_argument('--isotropic-db', action='store_true', help='database embeddings as isotropic unit Gaussians (rounds 1-5) instead of laid where the query encoder puts synthetic chunks')
    ap.add_argument('--ranks-share-gpu', action='store_true', help='dev: every rank on cuda:0 with the collectives over gloo (RCCL refuses two ranks on one device): runs the whole multi-rank bench on a one-GPU box; the line says so and is no scaling measurement')
    ap.add_argument('--resident-batches', type=int, default=4, help='distinct resident input batches rotated through the timed loop')
    return ap.parse_args(argv)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_plan(args, argv, environ):
    """None when this process is a rank (or the plain one-GPU run); otherwise the command line that starts the ranks.

    `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment (the shape of the driver's one-GPU command) launches its own N ranks: one
    process per GPU under torch.distributed.run on 127.0.0.1 with a free port.  The ranks' stdout is this process's stdout, and only rank 0 writes the JSON
    line to it.  `--gpus 1 --force-collectives` takes the same route with one rank (the RCCL protocol needs a rendezvous)."""
    if 'WORLD_SIZE' in environ or 'RANK' in environ:
        return None
    if args.gpus == 1 and not args.force_collectives:
        return None
    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
            '--master-port', str(free_port()), str(Path(__file__).resolve())] + list(argv)


def synthetic_database(cfg, n_patches, device, seed=1234, anchors=None):
    """Seeded synthetic DB built on the device: unit embeddings, reference row semantics for meta (util/retrieval.py:32,39-45 + sentinel), U(0,trunc) fp16-rounded
    scene chunks as the voxel store.  Embeddings: with ``anchors`` [A, 64] (unit query embeddings of synthetic chunks under the engine's weights) row i is a random
    anchor plus Gaussian noise of the anchors' own nearest-neighbour spacing, re-normalised -- the database then lies where the queries lie, as a trained encoder pair's
    would, and a batch's neighbour lists name thousands of different rows.  Without anchors: isotropic unit Gaussians -- under a random-init query encoder, whose
    embeddings all sit within 0.1 of one direction, every query then prefers the same ~100 rows (the patch gather read 1.5 MB instead of 134 MB)."""
    from rfuse import configs, synthetic
    _, trunc_t = configs.truncations(cfg)
    g = torch.Generator(device=device).manual_seed(seed)
    emb = torch.randn(n_patches + 1, 64, generator=g, device=device, dtype=torch.float32)
    if anchors is not None:
        a = anchors.to(device, torch.float32)
        sub = a[torch.randperm(a.shape[0], generator=g, device=device)[:1024]]
        d2 = (2.0 - 2.0 * sub @ a.T).clamp_min(0)
        d2[d2 < 1e-9] = 9.0                                          # (itself and exact duplicates)
        sigma = d2.min(dim=1).values.median().sqrt().item() / 8.0       # per-dimension noise: |noise| ~ the anchors' nearest-neighbour distance
        pick = torch.randint(0, a.shape[0], (n_patches + 1,), generator=g, device=device)
        emb = a[pick] + sigma * emb
    emb = emb / emb.norm(dim=1, keepdim=True).clamp_min(1e-12)
    meta = torch.from_numpy(synthetic.make_database(seed, cfg, n_patches, with_volumes=False, with_embeddings=False)['meta'])
    n_scenes = (n_patches + 63) // 64
    vols = torch.empty(n_scenes, 64, 64, 64, device=device)
    for lo in range(0, n_scenes, 1024):                        # in pieces: the fp16 round trip of 1 M patches would need 3 x 16 GB at once
        hi = min(lo + 1024, n_scenes)
        vols[lo:hi] = (torch.rand(hi - lo, 64, 64, 64, generator=g, device=device) * trunc_t).half().float()
    return emb, meta, vols


def algorithmic_flops(name, a):
    """SURVEY 8(d)-style algorithmic flop of a launch that is not a padded 3^3 box conv (those: conv_shape): 2 * cin * k^3 * cout per output voxel for the patch encoders'
    valid convs, 2 * n_in * n_out per row and Linear layer for the attention encoder, the 3^3 direct form for the 2^3 / 1^3 GEMM level -- no operand-splitting factor,
    no tile padding.  None: no such figure for this entry (the caller keeps the issued work and says so)."""
    name = ENTRY_ALIAS.get(name, name)
    if name in ('rf_conv3d_valid_leaky_split', 'rf_conv3d_valid_leaky_mfma', 'rf_conv3d_valid_leaky_lds', 'rf_conv3d_valid_leaky_valu', 'rf_conv3d_valid_leaky_valu_ex', 'rf_conv3d_valid_leaky'):
        n, cin, s, cout, k, stride = a[:6]
        return 2.0 * cin * k ** 3 * cout * ((s - k) // stride + 1) ** 3 * n
    if name == 'rf_conv3d_valid_leaky_split_ex':
        _, n, cin, s, cout, k, stride = a[:7]
        return 2.0 * cin * k ** 3 * cout * ((s - k) // stride + 1) ** 3 * n
    if name in ('rf_attn_mlp_split_volume', 'rf_attn_mlp_volume'):
        b, kv, c, s, t = a[:5]
        return b * kv * (s // 2) ** 3 * 2.0 * (c * 8 * 128 + 2 * 128 * 128 + 128 * 32)
    if name == 'rf_conv3d_e2_split_k3_gn_relu':
        cin, n, edge, cout = a[:4]
        return 2.0 * 27 * cin * cout * edge ** 3 * n
    return None


def oracle_chunk(refpath, cfg, state, db_host, raw, noise, knn):
    """the oracle's version of one chunk of a step: query embed -> kNN -> demotion -> compose -> networks -> df"""
    from rfuse import configs, synthetic
    trunc_i, trunc_t = configs.truncations(cfg)
    d, K = cfg['dataset_train'], cfg['K']
    with torch.no_grad():
        q = refpath.embed_queries(refpath.extract_query_windows(raw, cfg, trunc_i), state['fenc_input'], cfg).numpy()
        idx, dd = knn(q, db_host['emb'], 2 * K)
        mapping = refpath.demote_same_scene(refpath.mapping_rows(idx, dd, db_host['meta']), np.full(q.shape[0], -1), K)
        retr = refpath.compose_retrieval(mapping, db_host['volumes'], K, trunc_t)[None]
        retr = ((retr - np.float32(d['target_mean'])) / np.float32(d['target_std'])).astype(np.float32)
        x_in = synthetic.normalise_input(cfg, raw)[None, None]
        return refpath.forward_full(state, cfg, torch.from_numpy(x_in), torch.from_numpy(retr), trunc_t, noise)


def cpu_baseline(cfg, eng_state, db_host, raws, target_s=15.0, n_chunks=0):
    """The oracle (CPU port of the reference path, oracle/refpath.py) on the host cores: same stages as one GPU step.
    kNN = torch.cdist + topk fp32; everything else the pinned oracle."""
    from oracle import refpath
    cores = os.cpu_count() or 1
    K = cfg['K']
    noise_gen = torch.Generator().manual_seed(0)

    def one_chunk(raw):
        noise = None
        if cfg['attn_retrieval_mode']:
            noise = -torch.empty(cfg['attn_num_patch'] ** 3, K).exponential_(generator=noise_gen).log()
        return oracle_chunk(refpath, cfg, eng_state, db_host, raw, noise, refpath.knn_cdist_f32)

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


# ------------------------------------------------------------------------------------------------------ evidence
def recall_at_k(eng, database, emb_full, raw_dev, K, n_sub=128):
    """the search the engine ran (exact fp32 scan, sharded or not) against a float64 brute force over the FULL embedding matrix
    on a subset of this rank's queries; recall@k = |top-k(gpu) & top-k(f64)| / k averaged over the subset"""
    q = eng.embed_queries(raw_dev)
    dist_g, idx_g = database.search(q, 2 * K)
    step = max(1, q.shape[0] // n_sub)
    sub = torch.arange(0, q.shape[0], step, device=q.device)
    ref = torch.empty((sub.numel(), 2 * K), dtype=torch.int64, device=q.device)
    e64 = emb_full.double()
    en = (e64 * e64).sum(1)[None]
    for lo in range(0, sub.numel(), 16):                        # float64 distances in row blocks: [16, N] doubles at a time
        qs = q[sub[lo:lo + 16]].double()
        d64 = (qs * qs).sum(1, keepdim=True) + en - 2.0 * qs @ e64.T
        ref[lo:lo + 16] = torch.topk(d64, 2 * K, dim=1, largest=False, sorted=True).indices
    got = idx_g[sub]
    out = {}
    for k in sorted({1, K, 2 * K}):
        hit = (got[:, :k, None] == ref[:, None, :k]).any(dim=2).float().mean().item()
        out['recall@%d' % k] = hit
    out['exact_order'] = (got == ref).all(dim=1).float().mean().item()
    out['queries'] = int(sub.numel())
    out['reference'] = 'float64 brute force over all %d rows' % emb_full.shape[0]
    # how many DISTINCT database rows the batch's lists name: with random-init encoder weights the 64 windows of a synthetic chunk embed almost alike, so a
    # step's B * 64 * 2K list entries name only a few dozen rows (trained encoders spread them).  Nothing in the engine exploits that -- every one of the
    # B * K * 64 retrieved patches goes through the retrieval backbone -- but the patch gather of this workload is cache-resident whatever the batch.
    out['distinct_rows_in_the_batch_lists'] = int(torch.unique(idx_g).numel())
    out['list_entries'] = int(idx_g.numel())
    return out


def parity_of_bench_batch(cfg, eng, state, db_host, raws, raw_dev, chunks=(0, 17)):
    """df of the bench batch itself against the oracle: the engine refines the whole batch once more with EXPLICIT Gumbel
    noise (scaled so that the hard arg-max cannot flip on fp32 differences; None for the softmax configs) and ``chunks`` of
    it are compared with the oracle run on the same chunk, noise and database (float64 exact kNN)."""
    from oracle import refpath
    B, K = raws.shape[0], cfg['K']
    rows = cfg['attn_num_patch'] ** 3
    noise = None
    if cfg['attn_retrieval_mode']:
        noise = -torch.empty(B * rows, K).exponential_(generator=torch.Generator().manual_seed(11)).log() * 4.0
    df = eng.refine(raw_dev, gumbel_noise=noise.to(raw_dev.device) if noise is not None else None)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    worst = 0.0
    for b in chunks:
        if b >= B:
            continue
        nb = noise[b * rows:(b + 1) * rows] if noise is not None else None
        ref = oracle_chunk(refpath, cfg, state, db_host, raws[b], nb, refpath.knn_exact)
        worst = max(worst, float((df[b:b + 1].cpu() - ref).abs().max()))
    from rfuse import configs
    _, trunc_t = configs.truncations(cfg)
    out = {'value': worst, 'chunks': [b for b in chunks if b < B], 'against': 'oracle/refpath.py (fp32 torch-CPU networks, float64 exact kNN)',
           'tanh_output_max_abs': worst * 2.0 / trunc_t}
    # Both figures, always: `met` is north_star's bar -- 1e-4 abs on df against the fp32 oracle.  Where the truncation is above 1 (Matterport3D, 11.25)
    # the fp32 reference itself is 4.1e-4 from its own float64 evaluation on df (tests/golden/truth_C4.npz), so `met` is expected False there for ANY
    # evaluation order but ATen's; `met_tanh` is the bar the GPU tests apply on that config (1e-4 on the network's own tanh output, df error * 2 / trunc,
    # beside "no further from the float64 truth than the reference", tests/test_network_gpu.py:assert_df_parity).
    out.update(bar=1e-4, bar_applies_to='df (north_star: 1e-4 abs on the reconstructed TSDF)', met=worst <= 1e-4,
               met_tanh=worst * 2.0 / trunc_t <= 1e-4, trunc=trunc_t)
    if trunc_t > 1.0:
        out['note'] = ('trunc %.4g: the fp32 reference is itself 4.1e-4 from its float64 evaluation on df here (tests/golden/truth_C4.npz); the bar the GPU tests '
                       'hold this config to is met_tanh plus no-further-from-the-truth-than-the-reference' % trunc_t)
    return out


# entry points that are another one's launch with a different hand-over layout: same argument list, same work
ENTRY_ALIAS = {'rf_conv3d_up_split_presplit_pm': 'rf_conv3d_up_split_presplit', 'rf_conv3d_split_pre_pm_k3_relu': 'rf_conv3d_split_pre_k3_relu',
               'rf_conv3d_valid_leaky_split_pg': 'rf_conv3d_valid_leaky_split'}       # (n, cin, s, cout, k, stride): the persistent grid form of the same layer


def kernel_bytes(name, a, nulls=()):
    """algorithmic HBM bytes of one launch -- every input element read once, every output element written once -- or None.
    ``nulls``: positions of the launch's null pointer arguments (an output that is not written is not counted)"""
    name = ENTRY_ALIAS.get(name, name)
    if name in ('rf_conv3d_k3_gn_relu', 'rf_conv3d_k3_gn_relu_stats', 'rf_conv3d_k3_gn_relu_pool', 'rf_conv3d_k3_gn_relu_direct', 'rf_conv3d_up_k3_gn_relu',
                'rf_conv3d_up_split_k3_gn_relu', 'rf_conv3d_up_split_presplit', 'rf_conv3d_up_split_k3_gn_relu_ch8'):
        c0, c1, n, edge, cout = a[:5]
        return 4.0 * n * (c0 * edge ** 3 + c1 * (edge // 2) ** 3 + cout * edge ** 3)
    if name in ('rf_conv3d_split_k3_gn_relu', 'rf_conv3d_split_pre_k3_relu', 'rf_conv3d_split_presplit'):
        cin, n, edge, cout = a[:4]
        out_pos, pool_pos = {'rf_conv3d_split_k3_gn_relu': (7, 9), 'rf_conv3d_split_pre_k3_relu': (6, 8), 'rf_conv3d_split_presplit': (None, None)}[name]
        written = (0.0 if out_pos in nulls else 1.0) + (0.125 if pool_pos is not None and pool_pos not in nulls else 0.0)      # full output / fused MaxPool3d(2) output
        return 4.0 * n * edge ** 3 * (cin + cout * written)
    if name in ('rf_conv3d_split_k3_gn_relu_pointwise_tanh', 'rf_conv3d_split_k3_gn_relu_pointwise_tanh_ch8'):
        cin, n, edge, cout = a[:4]
        return 4.0 * n * edge ** 3 * (cin + 1)
    if name == 'rf_conv3d_split_pre_k3_relu_pool_presplit':          # pre-split in, the pooled tensor pre-split out (no fp32 pooled tensor: a scratch slot per workgroup)
        cin, n, edge, cout = a[:4]
        return 4.0 * n * edge ** 3 * (cin + cout / 8.0)
    if name == 'rf_conv3d_split_pre_presplit':
        cin, n, edge, cout = a[:4]
        return 4.0 * n * edge ** 3 * (cin + cout)
    if name == 'rf_conv3d_cin1_presplit':
        n, edge, cout = a[:3]
        return 4.0 * n * edge ** 3 * (1 + cout)
    if name == 'rf_conv3d_valid_leaky_split_ex':
        _, n, cin, s, cout, k, stride = a[:7]
        return 4.0 * n * (cin * s ** 3 + cout * ((s - k) // stride + 1) ** 3)
    if name in ('rf_conv3d_valid_leaky_split', 'rf_conv3d_valid_leaky_mfma', 'rf_conv3d_valid_leaky_lds', 'rf_conv3d_valid_leaky_valu', 'rf_conv3d_valid_leaky_valu_ex', 'rf_conv3d_valid_leaky'):
        n, cin, s, cout, k, stride = a[:6]
        return 4.0 * n * (cin * s ** 3 + cout * ((s - k) // stride + 1) ** 3)
    if name in ('rf_l2_topk', 'rf_l2_topk_keys'):
        nq, dim, n = a[:3]
        return 4.0 * dim * (n + nq)
    return None


def conv_shape(name, a):
    """(c0, c1, n, edge, cout) of a 3x3x3 GroupNorm-conv launch, or None"""
    name = ENTRY_ALIAS.get(name, name)
    if name in ('rf_conv3d_k3_gn_relu', 'rf_conv3d_k3_gn_relu_stats', 'rf_conv3d_k3_gn_relu_pool', 'rf_conv3d_up_k3_gn_relu', 'rf_conv3d_up_split_k3_gn_relu', 'rf_conv3d_up_split_presplit',
                'rf_conv3d_up_split_k3_gn_relu_ch8'):
        return tuple(a[:5])
    if name in ('rf_conv3d_split_k3_gn_relu', 'rf_conv3d_split_pre_k3_relu', 'rf_conv3d_split_presplit', 'rf_conv3d_split_k3_gn_relu_pointwise_tanh', 'rf_conv3d_split_k3_gn_relu_pointwise_tanh_ch8',
                'rf_conv3d_split_pre_k3_relu_pool_presplit', 'rf_conv3d_split_pre_presplit'):
        cin, n, edge, cout = a[:4]
        return (cin, 0, n, edge, cout)
    return None


def kernel_work(name, a, cfg):
    """(bound, work per launch, unit) of one C-ABI launch from its integer arguments, or None when the launch is bookkeeping"""
    name = ENTRY_ALIAS.get(name, name)
    if name in ('rf_conv3d_k3_gn_relu', 'rf_conv3d_k3_gn_relu_stats', 'rf_conv3d_k3_gn_relu_pool', 'rf_conv3d_k3_gn_relu_direct'):
        c0, c1, n, edge, cout = a[:5]
        if c0 + c1 == 1:                                         # first layer: reads 4 B, writes 4*cout B per voxel
            return 'hbm', 4.0 * n * edge ** 3 * (1 + cout), 'bytes'
        return 'mfma', 2.0 * 27 * (c0 + c1) * cout * edge ** 3 * n, 'flop, direct form (the kernels leave out zero-padding taps of border voxels: issued work is lower)'
    if name == 'rf_conv3d_up_k3_gn_relu':
        c0, c1, n, edge, cout = a[:5]
        from rfuse import ops
        return 'mfma', ops.conv_up_issued_flops(c0, c1, n, edge, cout), 'flop ISSUED (decoder form, 8 pre-summed taps for the upsampled channels, minus the skipped padding taps)'
    if name in ('rf_conv3d_up_split_k3_gn_relu', 'rf_conv3d_up_split_presplit', 'rf_conv3d_up_split_k3_gn_relu_ch8'):
        c0, c1, n, edge, cout = a[:5]
        from rfuse import ops
        return 'mfma-f16', ops.conv_up_split_issued_flops(c0, c1, n, edge, cout), ('f16 flop ISSUED (operand splitting: 3 MFMAs per product tile; decoder form; 28 tap slots per 27 taps, couts padded to 16); '
                                                                                  'fp32-equivalent %.1f GFLOP' % (2.0 * (27 * c0 + 8 * c1) * cout * edge ** 3 * n / 1e9))
    if name == 'rf_conv3d_cin1_presplit':
        n, edge, cout = a[:3]
        return 'hbm', 4.0 * n * edge ** 3 * (1 + cout), 'bytes'
    if name in ('rf_conv3d_split_k3_gn_relu', 'rf_conv3d_split_pre_k3_relu', 'rf_conv3d_split_pre_k3_relu_pool_presplit', 'rf_conv3d_split_pre_presplit', 'rf_conv3d_split_presplit'):
        cin, n, edge, cout = a[:4]
        from rfuse import ops
        return 'mfma-f16', ops.conv_split_issued_flops(cin, n, edge, cout), ('f16 flop ISSUED (operand splitting: 3 MFMAs per product tile; 28 tap slots per 27 taps, couts padded to 16); '
                                                                            'fp32-equivalent %.1f GFLOP' % (2.0 * 27 * cin * cout * edge ** 3 * n / 1e9))
    if name == 'rf_attn_mlp_split_volume':
        b, kv, c, s, t = a[:5]
        return 'mfma-f16', b * kv * (s // 2) ** 3 * 2.0 * 3 * (-(-c * 8 // 32) * 32 * 128 + 2 * 128 * 128 + 128 * 32), 'f16 flop ISSUED (operand splitting: 3 MFMAs per product tile)'
    if name in ('rf_attn_mlp_volume',):
        b, kv, c, s, t = a[:5]
        return 'mfma', b * kv * (s // 2) ** 3 * 2.0 * (c * 8 * 128 + 2 * 128 * 128 + 128 * 32), 'flop'
    if name == 'rf_attn_mlp_rows':
        rows, n_in = a[:2]
        return 'mfma', rows * 2.0 * (n_in * 128 + 2 * 128 * 128 + 128 * 32), 'flop'
    if name in ('rf_l2_topk', 'rf_l2_topk_keys'):
        nq, dim, n = a[:3]
        algo = a[5] if len(a) > 5 else 0
        if algo == 0:                                            # TOPK_AUTO: VALU scan below 100 k rows per shard, f16-MFMA-filtered scan above (retrieval.hip)
            algo = 3 if n >= 100_000 else 1
        bound = {1: 'valu', 2: 'mfma', 3: 'mfma-f16'}[algo]
        return bound, 2.0 * nq * n * dim, 'flop (q.x per (query, row) pair; %s); the shard itself is %d B read once' % (
            {1: 'exact distances on the packed-fp32 VALU', 2: 'fp32-MFMA filter + exact re-check', 3: 'f16-MFMA filter + exact re-check'}[algo], 4 * n * dim)
    if name == 'rf_conv3d_valid_leaky_split':
        n, cin, s, cout, k, stride = a[:6]
        so = (s - k) // stride + 1
        return 'mfma-f16', 3.0 * 2.0 * cin * k ** 3 * cout * so ** 3 * n, ('f16 flop, 3 MFMAs per product tile (operand splitting; channel / cout padding of the tiles not counted); '
                                                                        'fp32-equivalent %.1f GFLOP' % (2.0 * cin * k ** 3 * cout * so ** 3 * n / 1e9))
    if name == 'rf_conv3d_valid_leaky_split_ex':                  # (in_split, n, cin, s, cout, k, stride, out_split)
        _, n, cin, s, cout, k, stride = a[:7]
        so = (s - k) // stride + 1
        return 'mfma-f16', 3.0 * 2.0 * cin * k ** 3 * cout * so ** 3 * n, ('f16 flop, 3 MFMAs per product tile (operand splitting; channel / cout padding of the tiles not counted); '
                                                                        'fp32-equivalent %.1f GFLOP; %.2f GB in + out' % (2.0 * cin * k ** 3 * cout * so ** 3 * n / 1e9, 4e-9 * n * (cin * s ** 3 + cout * so ** 3)))
    if name in ('rf_conv3d_valid_leaky_mfma', 'rf_conv3d_valid_leaky_lds', 'rf_conv3d_valid_leaky_valu', 'rf_conv3d_valid_leaky_valu_ex', 'rf_conv3d_valid_leaky'):
        n, cin, s, cout, k, stride = a[:6]
        so = (s - k) // stride + 1
        return 'mfma', 2.0 * cin * k ** 3 * cout * so ** 3 * n, 'flop'
    if name == 'rf_linear':
        rows, nin, nout = a[:3]
        return 'mfma', 2.0 * rows * nin * nout, 'flop'
    if name in ('rf_conv3d_split_k3_gn_relu_pointwise_tanh', 'rf_conv3d_split_k3_gn_relu_pointwise_tanh_ch8'):
        cin, n, edge, cout = a[:4]
        from rfuse import ops
        return 'mfma-f16', ops.conv_split_issued_flops(cin, n, edge, cout), 'f16 flop ISSUED (operand splitting; the pointwise head + tanh ride in the epilogue); fp32-equivalent %.1f GFLOP' % (2.0 * 27 * cin * cout * edge ** 3 * n / 1e9)
    if name == 'rf_conv3d_e2_split_k3_gn_relu':
        cin, n, edge, cout = a[:4]
        return 'mfma-f16', 3.0 * 2.0 * (cin * edge ** 3) * (cout * edge ** 3) * n, 'f16 flop ISSUED (dense GEMM form of a 2^3 / 1^3 level: K = cin * voxels, N = cout * voxels; 3 MFMAs per product tile)'
    if name == 'rf_conv1x1_tanh':
        n, c, vox = a[:3]
        return 'hbm', 4.0 * n * vox * (c + 1), 'bytes'
    if name == 'rf_attn_blend':
        b, k, c, s, t = a[:5]
        return 'hbm', 4.0 * b * c * s ** 3 * (2 + k), 'bytes'
    if name in ('rf_gather_patches', 'rf_gather_patches_f16'):
        n_scenes, chunks, K = a[:3]
        return 'hbm', (6.0 if name.endswith('f16') else 8.0) * chunks * K * 64 * 4096, 'bytes'
    if name == 'rf_query_windows':
        b, s, ps, ctx = a[:4]
        return 'hbm', 8.0 * b * (s // ps) ** 3 * (ps + 2 * ctx) ** 3, 'bytes'
    if name in ('rf_maxpool3d_2', 'rf_maxpool3d_2_stats'):
        n, c, edge = a[:3]
        return 'hbm', 4.0 * n * c * edge ** 3 * 1.125, 'bytes'
    return None


def kernel_table(eng, raw_dev, cfg, steps=3, top=8):
    """per entry point (and shape) HIP-event time over ``steps`` serial steps (backbone on the main stream: no overlap)"""
    from rfuse import _lib
    lib = _lib.load()
    eng.serial = True
    eng.refine(raw_dev)
    torch.cuda.synchronize()
    records = []
    lib.start_profile(records)
    try:
        for _ in range(steps):
            eng.refine(raw_dev)
        torch.cuda.synchronize()
    finally:
        lib.stop_profile()
        eng.serial = False
    agg = {}
    nulls_of = {}
    for name, ints, e0, e1, nulls in records:
        key = (name, ints[:8])
        t = agg.setdefault(key, [0.0, 0])
        t[0] += e0.elapsed_time(e1)
        t[1] += 1
        nulls_of[key] = nulls
    total = sum(v[0] for v in agg.values()) / steps
    rows = []
    dominant = None                                              # (ms per launch, name, ints): the single launch with the largest mean duration that has a roofline
    for (name, ints), (ms, calls) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        per_launch = ms / calls
        if kernel_work(name, ints, cfg) and (dominant is None or per_launch > dominant[0]):
            dominant = (per_launch, name, ints, nulls_of[(name, ints)])
        if len(rows) >= top:
            continue
        row = {'entry': name, 'args': list(ints), 'launches_per_step': calls / steps, 'ms_per_launch': per_launch, 'ms_per_step': ms / steps, 'share': (ms / steps) / total}
        nbytes = kernel_bytes(name, ints, nulls_of[(name, ints)])
        if nbytes:
            row.update(algorithmic_bytes=nbytes, hbm_frac=nbytes / (per_launch * 1e-3) / 1e9 / HBM_PEAK_GBS)
        w = kernel_work(name, ints, cfg)
        if w:
            bound, work, unit = w
            if bound in ('mfma', 'mfma-f16', 'valu'):
                peak = F16_MFMA_PEAK_TFLOPS if bound == 'mfma-f16' else FP32_MFMA_PEAK_TFLOPS      # packed-fp32 VALU peak == fp32 MFMA peak
                ach = work / (per_launch * 1e-3) / 1e12
                row.update(bound='valu' if bound == 'valu' else 'mfma', achieved=ach, peak=peak, unit='TFLOP/s', frac=ach / peak, work=unit)
            else:
                ach = work / (per_launch * 1e-3) / 1e9
                row.update(bound='hbm', achieved=ach, peak=HBM_PEAK_GBS, unit='GB/s', frac=ach / HBM_PEAK_GBS, work='algorithmic ' + unit)
        assert 'frac' not in row or row['frac'] <= 1.0, 'a roofline fraction above 1 is a bookkeeping error, not a result: %s' % (row,)
        rows.append(row)
    return {'serial_ms_per_step': total, 'top': rows,
            'note': 'HIP events around each C-ABI launch, backbone kept on the main stream for this pass (one step after the other, nothing overlapped); a launch = all '
                    'kernels of that entry point; hbm_frac = algorithmic bytes (inputs read once + full outputs written once) / time / 8 TB/s'}, dominant


def batch_sweep(cfg, database, state, device, sizes=(1, 8, 64), steps=10):
    from rfuse import synthetic
    from rfuse.engine import RefinementEngine
    eng = RefinementEngine(cfg, device, database)
    eng.load_state_dicts(state)
    out = {}
    for b in sizes:
        raws = torch.from_numpy(np.stack([synthetic.make_chunk(20_000 + i, cfg)['input_raw'] for i in range(b)])).to(device)
        n = steps if b > 1 else 5 * steps
        for _ in range(3):
            eng.refine(raws)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            eng.refine(raws)
        torch.cuda.synchronize()
        out['B=%d' % b] = b * n / (time.perf_counter() - t0)
        if b == 1:
            try:
                graph, _, _ = eng.capture_graph(raws)
                for _ in range(3):
                    graph.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    graph.replay()
                torch.cuda.synchronize()
                out['B=1 (HIP graph replay)'] = n / (time.perf_counter() - t0)
            except Exception as e:                                  # capture is an optimisation of the launch-bound case, not the metric
                out['B=1 (HIP graph replay)'] = 'capture failed: %s' % (str(e).splitlines()[0][:120],)
    out['unit'] = 'chunks/s'
    return out


def scene_driver_rate(eng, cfg, scenes=8, grid=(4, 4, 4)):
    """rfuse.scene.refine_scene (scene -> chunk grid -> batched refine -> float16 -> recomposition) on synthetic scenes of
    grid[0] x grid[1] x grid[2] chunks: scenes/s including the recomposition on the device and the transfer of the finished canvases.  A side line, never `value`."""
    from rfuse import configs, scene, synthetic
    trunc_i, _ = configs.truncations(cfg)
    s_in = cfg['dataset_train']['input_chunk_size']
    n = grid[0] * grid[1] * grid[2]
    base = np.stack([synthetic.make_chunk(30_000 + i, cfg)['input_raw'] for i in range(n)])
    if base.ndim != 4 or base.shape[1] != s_in:
        return None                                                  # point-cloud inputs (C5) are not tiled scenes
    low = base.reshape(grid + (s_in,) * 3).transpose(0, 3, 1, 4, 2, 5).reshape(grid[0] * s_in, grid[1] * s_in, grid[2] * s_in)
    names, chunks = scene.split_scene(low, s_in, 'bench', pad_value=trunc_i)
    for vols in scene.refine_scenes(eng, ((names, chunks) for _ in range(3)), batch=32):      # warm-up: three results are alive at a time, their pinned buffers come from torch's caching host allocator afterwards
        pass
    t0 = time.perf_counter()
    for vols in scene.refine_scenes(eng, ((names, chunks) for _ in range(scenes)), batch=32):
        pass
    el = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(3):
        scene.refine_scene(eng, names, chunks, batch=32)
    el_single = (time.perf_counter() - t0) / 3
    shape = list(next(iter(vols.values())).shape)
    per_call = len(vols)                                           # tiled datasets: one superscene; ShapeNet-style datasets: every chunk is a scene
    return {'value': scenes * per_call / el, 'unit': 'scenes/s', 'chunks_per_scene': n // per_call, 'scene_voxels': shape, 'chunks_per_s': scenes * n / el,
            'single_call': {'chunks_per_s': n / el_single, 'ms': 1e3 * el_single,
                            'note': 'one rfuse.scene.refine_scene call on its own: the pipeline fills and drains inside the call and the last regions cross PCIe before it returns'},
            'note': 'rfuse.scene.refine_scenes over %d scenes: chunk grid -> ONE RefinementEngine.refine_stream through all scenes in batches of 32 -> float16 rounding -> '
                    'float64 canvases assembled on the device -> regions to pinned host memory on a copy stream, a scene handed out one batch late (its last '
                    'regions travel under the next scene\'s first batch); everything up to the numpy arrays in the caller\'s hands is inside the timed region' % scenes}


def host_io_rate(eng, raws_host, device, steps=10):
    """The same step when the boundary hands over HOST buffers (the reference's DataLoader does): pinned input -> device on the
    compute stream, refined chunks -> pinned host buffers on a copy stream, double buffered so that the copy of step i runs under
    step i+1.  Reported beside `value`, never as `value` (bench contract: inputs resident in HBM)."""
    B = raws_host.shape[0]
    pin_in = raws_host.pin_memory()
    outs = [torch.empty((B, 1, 64, 64, 64), dtype=torch.float32).pin_memory() for _ in range(2)]
    copy = torch.cuda.Stream(device)
    done = [torch.cuda.Event(), torch.cuda.Event()]

    def run(n):
        for i in range(n):
            dev_in = pin_in.to(device, non_blocking=True)
            df = eng.refine(dev_in)
            ready = torch.cuda.Event()
            ready.record()
            done[i & 1].synchronize()                               # the host buffer of two steps ago has been written
            with torch.cuda.stream(copy):
                copy.wait_event(ready)
                outs[i & 1].copy_(df, non_blocking=True)
                df.record_stream(copy)
                done[i & 1].record()
        torch.cuda.synchronize()
    run(3)
    t0 = time.perf_counter()
    run(steps)
    return {'value': B * steps / (time.perf_counter() - t0), 'unit': 'chunks/s', 'bytes_in_per_step': int(pin_in.numel() * 4),
            'bytes_out_per_step': int(outs[0].numel() * 4), 'note': 'pinned host input -> HBM and refined chunks -> pinned host memory '
            'every step (copy stream, double buffered); measured, not the headline metric'}


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    plan = launch_plan(args, argv, os.environ)
    if plan is not None:
        import subprocess
        sys.stdout.flush()
        raise SystemExit(subprocess.call(plan))
    # The contract is ONE line on stdout.  RCCL and gloo print banners to the C-level stdout of every rank ("RCCL version : ...", "[Gloo] Rank 0 is connected
    # ..."): keep a private handle on the real stdout for the JSON line and point file descriptor 1 at stderr for everything else.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if args.gpus != world:
        raise SystemExit('--gpus %d but WORLD_SIZE is %d: start `python bench.py --gpus %d` without WORLD_SIZE / RANK in the environment (it launches its own '
                         'ranks) or under torch.distributed.run --nproc-per-node %d' % (args.gpus, world, args.gpus, args.gpus))
    assert torch.cuda.is_available(), 'bench.py needs the GPU (no CPU fallback for the hot path)'
    knobs = sorted(k for k in os.environ if k.startswith('RFUSE_') and k != 'RFUSE_LIB')
    assert not knobs, 'refusing to measure with developer switches set: %s' % knobs
    if args.ranks_share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    force_dist = world == 1 and args.force_collectives and 'RANK' in os.environ        # dev: run the whole RCCL protocol with one rank
    if world > 1 or force_dist:
        if args.ranks_share_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=device)      # RCCL on ROCm
        assert dist.get_world_size() == args.gpus or force_dist, '--gpus %d but the process group has %d ranks' % (args.gpus, dist.get_world_size())

    from rfuse import configs, ops, synthetic
    from rfuse.database import PatchDatabase
    from rfuse.engine import RefinementEngine

    cfg = configs.get_config(args.config)
    n_patches = args.db or cfg['db_patches']
    B, K = args.batch, cfg['K']

    torch.manual_seed(0)
    eng = RefinementEngine(cfg, device, None)
    # the database's embeddings are laid where the query encoder puts synthetic chunks (rank- and world-independent anchor chunks: every rank builds the same rows)
    anchors, anchors_note = None, 'isotropic unit Gaussians'
    if not args.isotropic_db:
        with torch.no_grad():
            anchors = torch.cat([eng.embed_queries(torch.from_numpy(np.stack([synthetic.make_chunk(10_000 + a0 + b, cfg)['input_raw'] for b in range(32)])).to(device))
                                 for a0 in range(0, 128, 32)])
            # one anchor per DISTINCT embedding: a synthetic chunk's empty-space windows all embed alike (the reference's datasets drop such patches by occupancy,
            # dataset/patched_scene_dataset.py:28-32), and a clump of thousands of database rows at one point is a tie-break benchmark, not a search
            keep = torch.ones(anchors.shape[0], dtype=torch.bool, device=device)
            for lo in range(0, anchors.shape[0], 1024):
                d2 = (2.0 - 2.0 * anchors[lo:lo + 1024] @ anchors.T).clamp_min(0)
                earlier = torch.arange(anchors.shape[0], device=device)[None, :] < torch.arange(lo, min(lo + 1024, anchors.shape[0]), device=device)[:, None]
                keep[lo:lo + 1024] = ~((d2 < 1e-4) & earlier).any(dim=1)
            anchors = anchors[keep]
            anchors_note = 'laid where the query encoder puts synthetic chunks: %d distinct anchors + noise of their nearest-neighbour spacing' % anchors.shape[0]
            if anchors.shape[0] < 1024:
                # a random-init encoder that maps every window to (nearly) one point -- PCPatch48 of C5: 73 %% of the windows exactly, the rest within 1e-7 -- gives
                # nothing to lay a database along
                anchors_note = 'isotropic unit Gaussians (the random-init query encoder of this config is collapsed: %d distinct embeddings among 8192 windows)' % anchors.shape[0]
                anchors = None
    emb, meta, vols = synthetic_database(cfg, n_patches, device, anchors=anchors)
    database = PatchDatabase(emb, meta, vols, device, rank, world, half_store=True if args.half_store else (False if args.fp32_store else None))
    if database.half_store:
        del vols
        vols = database.volumes
    database.force_collectives = force_dist
    collective_events = [] if (world > 1 or force_dist) else None
    eng.database = database
    # every rank refines its own B chunks (chunk-parallel replicas); inputs resident in HBM
    # R distinct batches stay resident and the loops rotate through them, so consecutive steps search for different queries and gather different database
    # patches (one batch replayed K times would fetch the same 8192 patches, 134 MB, out of the Infinity Cache every step).  Batch 0 of rank 0 is the batch
    # the parity / recall legs check against the oracle.
    R = max(1, args.resident_batches)
    raws_all = [np.stack([synthetic.make_chunk(10_000 + (r * world + rank) * B + b, cfg)['input_raw'] for b in range(B)]) for r in range(R)]
    raws = raws_all[0]
    batches = [torch.from_numpy(x).to(device) for x in raws_all]
    raw_dev = batches[0]

    def rotate(n):
        return (batches[i % R] for i in range(n))

    def barrier():
        if world > 1 or force_dist:
            dist.barrier()

    # The timed loop is RefinementEngine.refine_stream: K steps = K batches, software-pipelined -- the front end of batch i + 1 (query encoder,
    # exact top-k, patch gather, U-Net backbone) is issued on a helper stream beside the back end of batch i (retrieval backbone, attention,
    # decoder).  Every step does all of its work inside the timed region (the first front end and the last back end included); the same K steps
    # through refine() one after the other are timed right after it and reported as `unpipelined`.
    # Device spin-up ahead of the W warm-up steps (untimed, reported as `spinup_steps`): an idle MI355X needs a few hundred milliseconds of work to reach its
    # clocks, and the first passes also pack the weight images and grow both streams' allocator pools.  The timed region is exactly K steps after W warm-up steps.
    SPINUP = 12
    for df in eng.refine_stream(rotate(SPINUP)):
        pass
    # roofline: WHICH launch is the dominant one is decided by a serial pass (HIP events around every C-ABI launch, one step after the other, nothing
    # overlapped: the launch with the largest mean duration among those with a roofline) -- inside the pipelined step a launch shares the CUs with the other
    # streams' kernels and its in-step duration is partly contention (round 3's C5 line named a 15x inflated launch).  THAT entry point is then bracketed
    # with HIP events on its launch stream inside the timed region: `roofline.achieved` is its algorithmic work / its mean in-step duration (the contract's
    # definition), `frac_serial` the same against its duration in the serial pass.
    from rfuse import _lib
    lib = _lib.load()
    kernels, dominant = kernel_table(eng, raw_dev, cfg, top=args.kernels_top)
    for df in eng.refine_stream(rotate(args.warmup)):
        pass
    torch.cuda.synchronize()
    dom_records = []
    if dominant is not None:
        lib.start_profile(dom_records, only={dominant[1]})
    database.collective_events = collective_events
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for df in eng.refine_stream(rotate(args.steps)):
        pass
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    database.collective_events = None
    lib.stop_profile()
    # further blocks of the same K steps: box-to-box and run-to-run spread is a few percent (clocks follow the power budget), one block cannot show a 3 % change
    block_ms = [1e3 * elapsed / args.steps]
    for _ in range(max(0, args.repeats)):
        barrier()
        torch.cuda.synchronize()
        tb = time.perf_counter()
        for df in eng.refine_stream(rotate(args.steps)):
            pass
        torch.cuda.synchronize()
        barrier()
        block_ms.append(1e3 * (time.perf_counter() - tb) / args.steps)
    for _ in range(min(2, args.warmup)):                          # the front end's tensors now come from the main stream's allocator pool: let it grow
        eng.refine(raw_dev)
    barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for x in rotate(args.steps):
        eng.refine(x)
    torch.cuda.synchronize()
    barrier()
    elapsed_plain = time.perf_counter() - t1
    t = torch.tensor([elapsed] + [m * args.steps / 1e3 for m in block_ms[1:]], dtype=torch.float64, device=device)
    if world > 1 or force_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t[0].item())
    block_ms = [1e3 * float(v) / args.steps for v in t.tolist()]
    assert torch.isfinite(df).all()

    in_step = [e0.elapsed_time(e1) for name, ints, e0, e1, _ in dom_records if dominant is not None and ints[:8] == dominant[2]]
    per_rank_collectives = None
    if collective_events:
        mine = (float(np.mean([e[0].elapsed_time(e[1]) for e in collective_events])), float(np.mean([e[2].elapsed_time(e[3]) for e in collective_events])))
        per_rank_collectives = [None] * dist.get_world_size()
        dist.all_gather_object(per_rank_collectives, mine)

    # every rank computes its own recall (collective-free for world == 1; with shards the search itself is a collective)
    recall = None
    if not args.no_extras:
        recall = recall_at_k(eng, database, emb, raw_dev, K)

    if rank == 0:
        value = world * B * args.steps / elapsed
        roof = None
        if dominant is not None and in_step:
            serial_ms, entry, ints, nulls = dominant
            kern_ms = float(np.mean(in_step))
            bound, work, work_unit = kernel_work(entry, ints, cfg)
            nbytes = kernel_bytes(entry, ints, nulls)
            if bound == 'hbm':
                peak, unit, scale = HBM_PEAK_GBS, 'GB/s', 1e9
            else:
                peak, unit, scale = (F16_MFMA_PEAK_TFLOPS if bound == 'mfma-f16' else FP32_MFMA_PEAK_TFLOPS), 'TFLOP/s', 1e12
            issued = work / (kern_ms * 1e-3) / scale                # what the matrix pipe was fed (split operands, tap slots, cout padding)
            shape = conv_shape(entry, ints)
            direct = None
            if shape is not None and bound != 'hbm':
                c0_, c1_, n_, edge_, cout_ = shape
                direct = 2.0 * 27 * (c0_ + c1_) * cout_ * edge_ ** 3 * n_      # the layer as the reference evaluates it
            # `achieved` / `frac`: SURVEY 8(d)'s algorithmic work of the launch (a conv layer: 2 * 27 * cin * cout flop per output voxel, the reference's direct
            # form; an HBM-bound launch: its algorithmic bytes) / the launch's mean in-step duration / the peak of the pipe the launch runs on.
            # `issue_frac`: the flop actually ISSUED on that pipe / the same peak (pipe occupancy; round 4 printed this one as `frac`).
            if direct is None and bound != 'hbm':
                direct = algorithmic_flops(entry, ints)              # valid convs, attention MLP, 2^3 GEMM form: the reference's direct form, not the issued f16 flop
            algo_work = direct if direct is not None else work
            achieved = algo_work / (kern_ms * 1e-3) / scale
            # HBM traffic of the dominant kernel: OFFLINE PMC (separate rocprofv3 --pmc passes of this same command, FETCH_SIZE x2 gfx950 correction + WRITE_SIZE,
            # summary committed under profiles/), scaled per sample to this launch; None when the committed summary is of another kernel
            traffic, pmc_name = None, None
            other = REPO / 'profiles' / ('r06_dominant_kernel_%s.json' % args.config)      # a dominant launch that is not a padded 3^3 conv (C5): matched by entry + arguments
            if shape is None and other.exists():
                pmc = json.loads(other.read_text())
                if pmc.get('entry') == entry and pmc.get('args') == list(ints)[:len(pmc.get('args', []))]:
                    traffic, pmc_name = pmc['traffic_bytes_per_launch'], other.name
            for pmc_file in (REPO / 'profiles' / 'r06_dominant_kernel.json', REPO / 'profiles' / 'r05_dominant_kernel.json', REPO / 'profiles' / 'r04_dominant_kernel.json'):
                if not pmc_file.exists() or shape is None:
                    continue
                pmc = json.loads(pmc_file.read_text())
                same = pmc.get('entry') == entry or {pmc.get('entry'), entry} <= {'rf_conv3d_up_split_k3_gn_relu', 'rf_conv3d_up_split_presplit', 'rf_conv3d_up_split_presplit_pm'}
                if same and pmc.get('shape') == [shape[0], shape[1], shape[3], shape[4]]:
                    traffic, pmc_name = pmc['traffic_bytes_per_sample'] * shape[2], pmc_file.name
                    break
            pipe = {'hbm': 'HBM3E', 'mfma-f16': 'the dense f16 MFMA peak: the launch multiplies on v_mfma_f32_16x16x32_f16 (fp32 operands as two f16 pieces, 3 MFMAs per product '
                                                'tile, exact products, separate hi / lo fp32 accumulators), so that pipe is the one that bounds it -- against the fp32 MFMA '
                                                'peak (157.3 TFLOP/s) the same rate would read above 1',
                    'mfma': 'the dense fp32 MFMA peak (v_mfma_f32_16x16x4_f32)'}.get(bound, bound)
            roof = {'bound': 'hbm' if bound == 'hbm' else 'mfma',
                    'kernel': '%s %s' % (entry, list(ints)),
                    'achieved': achieved, 'peak': peak, 'unit': unit, 'frac': achieved / peak,
                    'work_per_launch': algo_work,
                    'work': (('SURVEY 8(d) algorithmic flop of the layer: 2 * 27 * cin * cout per output voxel (the reference\'s direct form)' if shape is not None else
                              'algorithmic flop of the launch in the reference\'s direct form (2 * cin * k^3 * cout per output voxel / 2 * n_in * n_out per row and layer; no operand-splitting factor, no tile padding)')
                             if direct is not None else work_unit)
                            + '; peak = ' + pipe,
                    'issue_frac': issued / peak, 'issued_per_launch': work, 'issued': work_unit,
                    'launch_ms': kern_ms, 'launches_timed': len(in_step),
                    'launch_ms_serial': serial_ms, 'frac_serial': algo_work / (serial_ms * 1e-3) / scale / peak, 'issue_frac_serial': work / (serial_ms * 1e-3) / scale / peak,
                    'chosen_by': 'largest mean duration per launch in the serial per-kernel pass (`kernels`); launch_ms = the same entry point bracketed with HIP events on its '
                                 'launch stream inside the timed (pipelined) region',
                    'traffic': traffic, 'traffic_unit': 'bytes/launch, OFFLINE PMC (%s), not measured in this run' % (pmc_name or 'no summary of this kernel committed'),
                    'algorithmic_bytes_per_launch': nbytes, 'hbm_frac': (nbytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if nbytes else None,
                    'heavy_launches_ms_serial': {'%s %s' % (r['entry'], r['args']): r['ms_per_launch'] for r in kernels['top']}}
            if direct is not None:
                roof.update(direct_form_flops_per_launch=direct)
                if shape is not None:
                    roof.update(fp32_equivalent_tflops=2.0 * (27 * c0_ + (8 if 'up' in entry else 27) * c1_) * cout_ * edge_ ** 3 * n_ / (kern_ms * 1e-3) / 1e12)
            assert roof['frac'] <= 1.0 and roof['issue_frac'] <= 1.0 and roof['issue_frac_serial'] <= 1.0, roof
        out = {
            'metric': '64^3 TSDF chunks/sec (retrieve+attend+refine)', 'value': value, 'unit': 'chunks/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'spinup_steps': SPINUP, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic (%d distinct resident batches rotated through the timed loop; database embeddings %s)' % (R, anchors_note),
            'arithmetic': 'fp32 tensors and fp32 accumulation throughout; the heavy 3x3x3 convolutions multiply on the F16 matrix cores with every fp32 operand '
                          'carried as two f16 pieces (x = h + l / 2^11, exact f16 x f16 products, separate hi / lo fp32 accumulators): measured error against '
                          'float64 is lower than that of the fp32 MFMA chain (tests/test_kernels_gpu.py, tools/micro/split_probe.hip)' if ops.CONV_ARITH == 'split' else 'fp32 (v_mfma_f32_16x16x4_f32)',
            'config': {'workload': '%s: %s super-res/recon ->064, synthetic chunks, K=%d, DB=%d patches (exact L2 top-%d), '
                                   'random-init weights' % (args.config, cfg['dataset_train']['dataset_name'], K, n_patches, 2 * K),
                       'chunks_per_gpu_per_step': B, 'db_patches': n_patches,
                       'parallelism': 'chunk-parallel replicas x%d, DB embedding matrix sharded %d-way: RCCL all-gather of the queries + all-to-all of the packed top-2K keys' % (world, world)},
            'schedule': 'RefinementEngine.refine_stream: steps software-pipelined (front end of batch i + 1 beside the back end of batch i), U-Net backbone beside the retrieval path',
            'blocks': {'ms_per_step': block_ms, 'min_ms': float(np.min(block_ms)), 'median_ms': float(np.median(block_ms)), 'value_at_median': world * B / (float(np.median(block_ms)) * 1e-3),
                       'note': '`value` is the FIRST block (the contract\'s K steps after W warm-up steps); the others are the same K steps timed again, back to back (max over ranks)'},
            'unpipelined': {'value': world * B * args.steps / elapsed_plain, 'ms_per_step': 1e3 * elapsed_plain / args.steps,
                            'note': 'the same K steps through RefinementEngine.refine() one after the other (rank-0 clock)'},
            'roofline': roof,
            'kernels': kernels,
            'recall_at_k': recall,
        }
        out['rccl_ranks'] = dist.get_world_size() if (world > 1 or force_dist) and not args.ranks_share_gpu else 0
        out['topk_scan'] = 'VALU scan: the shard probed as clumped (PatchDatabase.scan_algo)' if database.scan_algo else 'by size: VALU scan below 4e7 (query, row) pairs, split-f16-MFMA-filtered scan above'
        if args.ranks_share_gpu:
            out['ranks_share_gpu'] = 'dev run: %d ranks on ONE GPU, collectives over gloo (staged through the host) -- a functional run of the multi-rank path, not a scaling measurement' % world
        if collective_events:
            q_ms, k_ms = per_rank_collectives[0]
            out['collectives'] = {'per_rank_ms': [{'rank': r, 'all_gather_queries_ms': v[0], 'all_to_all_keys_ms': v[1]} for r, v in enumerate(per_rank_collectives)],
                                  ('gloo_ranks_on_one_gpu' if args.ranks_share_gpu else 'rccl_ranks'): dist.get_world_size(), 'per_step': 2, 'all_gather_queries_ms': q_ms, 'all_to_all_keys_ms': k_ms,
                                  'keys_bytes_received_per_rank': world * B * 64 * 2 * K * 8, 'step_ms': 1e3 * elapsed / args.steps,
                                  'note': 'HIP events on the issuing stream around each collective (includes waiting for the slowest rank); the U-Net backbone '
                                          'forked onto the side stream before the search keeps running while they are in flight'}
            if force_dist:
                # one rank, protocol forced: the same timed loop WITHOUT the collectives -> what the exchange adds to a step once overlap is counted
                database.force_collectives = False
                for df in eng.refine_stream(raw_dev for _ in range(2)):
                    pass
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for df in eng.refine_stream(rotate(args.steps)):
                    pass
                torch.cuda.synchronize()
                plain_ms = 1e3 * (time.perf_counter() - t1) / args.steps
                database.force_collectives = True
                out['collectives'].update(step_ms_without_collectives=plain_ms, exposed_ms_per_step=1e3 * elapsed / args.steps - plain_ms)
        state = None
        if world == 1 and not args.no_extras:
            state = {n: {k: v.detach().cpu() for k, v in m.state_dict().items()} for n, m in eng.modules().items()}
            out['batch_sweep'] = batch_sweep(cfg, database, state, device)
            out['host_io'] = host_io_rate(eng, torch.from_numpy(raws), device)
            out['scene_driver'] = scene_driver_rate(eng, cfg)
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
        host_db_ok = n_patches <= 200_000                        # the oracle needs the voxel store on the host (16 GB at 1 M patches)
        if world == 1 and host_db_ok and not (args.no_cpu_baseline and args.no_extras):
            state = state or {n: {k: v.detach().cpu() for k, v in m.state_dict().items()} for n, m in eng.modules().items()}
            db_host = {'emb': emb.cpu().numpy(), 'meta': meta.numpy(), 'volumes': vols.float().cpu().numpy()}
            if not args.no_extras:
                out['parity_max_abs'] = parity_of_bench_batch(cfg, eng, state, db_host, raws, raw_dev)
            out['cpu_baseline'] = None if args.no_cpu_baseline else cpu_baseline(cfg, state, db_host, raws, n_chunks=args.cpu_chunks)
        else:
            out['cpu_baseline'] = None
            if world == 1 and not host_db_ok:
                out['cpu_baseline_note'] = 'skipped: the oracle needs the %d-patch voxel store on the host' % n_patches
        print(json.dumps(out), file=json_out)
        json_out.flush()
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
