"""Seeded synthetic stand-ins for the reference's datasets (there is no dataset and no network here).

What is reproduced from the reference's data path (and nothing else):
  * unsigned truncated distance fields, trunc = 3 voxels rounded through float16  (dataset/scene.py:30-33)
  * values stored as float16 then widened to float32                              (dataset/scene.py:61,71)
  * normalisation (x - mean) / std with the per-dataset constants                 (dataset/patched_scene_dataset.py:127-133)
  * point cloud -> {0,1} occupancy grid scatter for surface reconstruction        (util/misc.py:73-78)
  * patch extent grid: non-overlapping 16^3 target boxes in a 64^3 chunk          (dataset/scene.py:152-160)

Everything is numpy ``default_rng(seed)`` so tests, golden vectors and bench inputs agree bit for bit.
"""
import numpy as np

from .configs import truncations


def _f16_round(a):
    return a.astype(np.float16).astype(np.float32)


def _shape_params(rng):
    n_shapes = int(rng.integers(1, 5))
    kinds = rng.integers(0, 2, size=n_shapes)               # 0 sphere, 1 box
    centers = rng.uniform(0.2, 0.8, size=(n_shapes, 3))     # in unit-cube coordinates
    sizes = rng.uniform(0.08, 0.3, size=(n_shapes, 3))
    return kinds, centers, sizes


def _udf_on_grid(res, kinds, centers, sizes):
    """Unsigned distance (unit-cube units) from voxel centres of a res^3 grid to the union of shapes."""
    t = (np.arange(res, dtype=np.float64) + 0.5) / res
    gx, gy, gz = np.meshgrid(t, t, t, indexing='ij')
    p = np.stack([gx, gy, gz], axis=-1)
    d = np.full((res, res, res), np.inf)
    for kind, c, s in zip(kinds, centers, sizes):
        if kind == 0:
            di = np.abs(np.linalg.norm(p - c, axis=-1) - s[0])
        else:
            q = np.abs(p - c) - s
            outside = np.linalg.norm(np.maximum(q, 0.0), axis=-1)
            inside = np.minimum(np.max(q, axis=-1), 0.0)
            di = np.abs(outside + inside)
        d = np.minimum(d, di)
    return d


def make_chunk(seed, config):
    """One synthetic 64^3 chunk: returns dict(input_raw, target_raw) un-normalised float32 fields.

    input_raw : [S,S,S] with S = input_chunk_size (8 / 16), or the [128]^3 {0,1} grid for surface reconstruction
    target_raw: [64,64,64]
    """
    rng = np.random.default_rng(seed)
    d = config['dataset_train']
    trunc_i, trunc_t = truncations(config)
    kinds, centers, sizes = _shape_params(rng)
    extent_world = 64 * d['voxel_size_target']
    tgt = _udf_on_grid(64, kinds, centers, sizes) * extent_world
    tgt = _f16_round(np.minimum(tgt, trunc_t).astype(np.float32))
    if config['task'] == 'surface_reconstruction':
        # util/misc.py:73-78 with grid_res=128, scale_factor = 128/64, pad=0; points live in 64^3 voxel coords
        n_pts = d['num_points']
        pts = _surface_points(rng, kinds, centers, sizes, n_pts) * 64.0
        grid = np.zeros((128, 128, 128), dtype=np.float32)
        pg = np.clip(pts * 2.0, 0, 127).astype(np.uint32)
        grid[pg[:, 0], pg[:, 1], pg[:, 2]] = 1
        inp = grid
    else:
        s_in = d['input_chunk_size']
        inp = _udf_on_grid(s_in, kinds, centers, sizes) * extent_world
        inp = _f16_round(np.minimum(inp, trunc_i).astype(np.float32))
    return {'input_raw': inp, 'target_raw': tgt}


def _surface_points(rng, kinds, centers, sizes, n):
    """n points on the surfaces of the shapes (unit-cube coordinates)."""
    which = rng.integers(0, len(kinds), size=n)
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    pts = np.empty((n, 3))
    for i in range(n):
        k = which[i]
        if kinds[k] == 0:
            pts[i] = centers[k] + u[i] * sizes[k][0]
        else:
            face = np.argmax(np.abs(u[i]))
            q = rng.uniform(-1, 1, size=3)
            q[face] = np.sign(u[i][face])
            pts[i] = centers[k] + q * sizes[k]
    return np.clip(pts, 0.0, 1.0 - 1e-6)


def normalise_input(config, x):
    d = config['dataset_train']
    return ((x - np.float32(d['input_mean'])) / np.float32(d['input_std'])).astype(np.float32)


def normalise_target(config, x):
    d = config['dataset_train']
    return ((x - np.float32(d['target_mean'])) / np.float32(d['target_std'])).astype(np.float32)


def uniform_stress_volume(seed, shape, trunc):
    """U(0, trunc) volume rounded through float16 -- the 'uniform-random stress' input of SURVEY 8(d)."""
    rng = np.random.default_rng(seed)
    return _f16_round((rng.random(size=shape) * trunc).astype(np.float32))


def patch_boxes_64():
    """[64,6] int32 (x0,x1,y0,y1,z0,z1): the 4x4x4 non-overlapping 16^3 boxes of a 64^3 chunk in the
    reference's enumeration order (meshgrid indexing='ij' flattened, dataset/scene.py:152-160)."""
    o = np.arange(0, 64, 16, dtype=np.int32)
    x, y, z = np.meshgrid(o, o, o, indexing='ij')
    x, y, z = x.ravel(), y.ravel(), z.ravel()
    return np.stack([x, x + 16, y, y + 16, z, z + 16], axis=1).astype(np.int32)


def make_database(seed, config, n_patches, latent_dim=64, with_volumes=True, with_embeddings=True):
    """Synthetic retrieval database with the reference's row semantics (util/retrieval.py:32,39-45):

      meta  [N+1,7] int32  (scene_idx, x0,x1,y0,y1,z0,z1)  un-padded 16^3 boxes; last row = sentinel (-1, 0,16,0,16,0,16)
      emb   [N+1,64] float32 unit vectors
      volumes [S,64,64,64] float32 raw (un-normalised, fp16-representable) scene chunks, S = ceil(N/64)

    Embeddings are seeded unit Gaussians (the DB-side Patch32 encoder is a 'next' row).
    """
    rng = np.random.default_rng(seed)
    n_scenes = (n_patches + 63) // 64
    boxes = patch_boxes_64()
    scene_idx = np.repeat(np.arange(n_scenes, dtype=np.int32), 64)[:n_patches]
    box_rows = np.tile(boxes, (n_scenes, 1))[:n_patches]
    meta = np.concatenate([scene_idx[:, None], box_rows], axis=1)
    sentinel = np.array([[-1, 0, 16, 0, 16, 0, 16]], dtype=np.int32)
    meta = np.concatenate([meta, sentinel], axis=0).astype(np.int32)
    out = {'meta': meta, 'n_scenes': n_scenes}
    if with_embeddings:
        emb = rng.standard_normal(size=(n_patches + 1, latent_dim)).astype(np.float32)
        emb /= np.maximum(np.linalg.norm(emb, axis=1, keepdims=True), 1e-12)
        out['emb'] = emb.astype(np.float32)
    if with_volumes:
        vols = np.empty((n_scenes, 64, 64, 64), dtype=np.float32)
        for s in range(n_scenes):
            vols[s] = make_chunk(seed * 1_000_003 + 17 + s, config)['target_raw']
        out['volumes'] = vols
    return out


def seeded_state_dict(shapes, seed):
    """Deterministic weights for golden vectors: ``shapes`` is an ordered {key: shape} mapping.

    conv / linear weights and biases: U(-1/sqrt(fan_in), +1/sqrt(fan_in)) (the torch default bound);
    GroupNorm weight 1 + 0.25 U(-1,1), bias 0.25 U(-1,1) so the affine part is exercised;
    sig_scale / sig_shift keep the reference's init 35 / -27 (model/attention.py:60-63).
    """
    rng = np.random.default_rng(seed)
    out = {}
    for key, shape in shapes.items():
        shape = tuple(int(s) for s in shape)
        if key.endswith('sig_scale'):
            out[key] = np.full(shape, 35.0, dtype=np.float32)
        elif key.endswith('sig_shift'):
            out[key] = np.full(shape, -27.0, dtype=np.float32)
        elif 'groupnorm' in key:
            u = rng.uniform(-1, 1, size=shape)
            out[key] = (1.0 + 0.25 * u if key.endswith('weight') else 0.25 * u).astype(np.float32)
        else:
            if key.endswith('bias'):
                # bias bound uses the fan_in of the companion weight, drawn right before it
                fan_in = out['__last_fan_in__']
            else:
                fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
                out['__last_fan_in__'] = fan_in
            b = 1.0 / np.sqrt(max(fan_in, 1))
            out[key] = rng.uniform(-b, b, size=shape).astype(np.float32)
    out.pop('__last_fan_in__', None)
    return out
