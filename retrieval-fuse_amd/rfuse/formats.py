"""On-disk formats of the reference's retrieval pipeline, so databases / mappings /
composed retrievals written by either code base are readable by the other.

  database.npy        (N+1) x 71 float32 rows [scene_idx, x0,x1,y0,y1,z0,z1, emb_0..emb_63], last row = sentinel
                      (reference util/retrieval.py:32,39-48)
  index.json          list of scene names, position = scene_idx                      (util/retrieval.py:48)
  map_{train,val}.npy pickled dict  patch_name -> [K,8] float32 rows [scene_idx, box6, dist]   (util/retrieval.py:99-100,235,238)
  compose/<scene>.npz 'arr_0' = [K,64,64,64] float32 retrieval volumes                (util/retrieval.py:248)
  patch names         '<scene>--x0_x1_y0_y1_z0_z1', 4-digit zero padded, PADDED extents (dataset/scene.py:169-177)

Host-side numpy only; nothing here is on the hot path.
"""
import json
from pathlib import Path

import numpy as np


def patch_name(scene, extent):
    """dataset/scene.py:169-171"""
    e = [int(v) for v in extent]
    return f"{scene}--{e[0]:04d}_{e[1]:04d}_{e[2]:04d}_{e[3]:04d}_{e[4]:04d}_{e[5]:04d}"


def parse_patch_name(identifier):
    """dataset/scene.py:173-177 -> (scene, [x0,x1,y0,y1,z0,z1])"""
    scene, rest = identifier.split('--')
    return scene, [int(r) for r in rest.split('_')]


def chunk_patch_names(scene, patch_size=16, context=8, chunk=64):
    """The 64 padded-extent patch names of one chunk in the reference's enumeration order (dataset/scene.py:152-160)."""
    o = range(0, chunk - patch_size + 1, patch_size)
    w = patch_size + 2 * context
    return [patch_name(scene, (x, x + w, y, y + w, z, z + w)) for x in o for y in o for z in o]


def database_to_rows(meta, emb):
    """(meta [N+1,7] int, emb [N+1,D] float32) -> database.npy array [N+1, 7+D] float32."""
    return np.concatenate([np.asarray(meta, dtype=np.float32), np.asarray(emb, dtype=np.float32)], axis=1)


def rows_to_database(rows):
    rows = np.asarray(rows)
    return rows[:, :7].astype(np.int32), np.ascontiguousarray(rows[:, 7:], dtype=np.float32)


def save_database(tree_path, meta, emb, scene_names):
    tree_path = Path(tree_path)
    tree_path.mkdir(parents=True, exist_ok=True)
    np.save(tree_path / 'database', database_to_rows(meta, emb))
    (tree_path / 'index.json').write_text(json.dumps(list(scene_names)))


def load_database(tree_path):
    tree_path = Path(tree_path)
    meta, emb = rows_to_database(np.load(tree_path / 'database.npy'))
    return meta, emb, json.loads((tree_path / 'index.json').read_text())


def mapping_to_dict(names, meta, dist):
    """names [P]; meta [P,K,7] int; dist [P,K] -> {name: [K,8] float32} exactly as flann_knn_worker stores it."""
    rows = np.concatenate([np.asarray(meta, dtype=np.float32), np.asarray(dist, dtype=np.float32)[..., None]], axis=-1)
    return {n: rows[i] for i, n in enumerate(names)}


def dict_to_mapping(mapping, names):
    rows = np.stack([np.asarray(mapping[n], dtype=np.float32) for n in names])
    return rows[..., :7].astype(np.int32), np.ascontiguousarray(rows[..., 7])


def save_mapping(path, mapping):
    np.save(path, mapping, allow_pickle=True)          # np.save of a dict, as the reference does (util/retrieval.py:235)


def load_mapping(path):
    return np.load(path, allow_pickle=True)[()]        # util/retrieval.py:245


def save_compose(retrievals_dir, scene, volumes):
    d = Path(retrievals_dir) / 'compose'
    d.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(d / f'{scene}.npz', np.asarray(volumes, dtype=np.float32))


def load_compose(retrievals_dir, scene):
    return np.load(Path(retrievals_dir) / 'compose' / f'{scene}.npz')['arr_0']
