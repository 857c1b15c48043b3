"""On-disk formats of the reference's retrieval pipeline, so databases / mappings /
composed retrievals written by either code base are readable by the other.

  database.npy        (N+1) x 71 float32 rows [scene_idx, x0,x1,y0,y1,z0,z1, emb_0..emb_63], last row = sentinel
                      (reference util/retrieval.py:32,39-48)
  index.json          list of scene names, position = scene_idx                      (util/retrieval.py:48)
  map_{train,val}.npy pickled dict  patch_name -> [K,8] float32 rows [scene_idx, box6, dist]   (util/retrieval.py:99-100,235,238)
  compose/<scene>.npz 'arr_0' = [K,64,64,64] float32 retrieval volumes                (util/retrieval.py:248)
  patch names         '<scene>--x0_x1_y0_y1_z0_z1', 4-digit zero padded, PADDED extents (dataset/scene.py:169-177)

The format helpers are host-side numpy.  ``create_dictionary`` / ``retrieval_mapping`` / ``compose_scene`` / ``retrievals_to_disk`` drive the DEVICE path
(query encoder, exact top-2K, demotion, patch gather: the same launches the engine's online step uses) and write the reference's files from it -- the
reference's offline ``util/retrieval.py --mode map`` / ``--mode compose`` (:210-248).
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


# ------------------------------------------------------------------------------------------------ the device path -> the reference's files
def create_dictionary(config, fenc_target, volumes, scene_names, tree_path, device, patch_mask=None, half_store=None):
    """The reference's ``create_dictionary`` (util/retrieval.py:29-48) on the device: the 64 target windows of every scene chunk ``volumes[s]`` ([S,64,64,64] raw)
    embedded by ``fenc_target`` + the sentinel row -> ``<tree_path>/database.npy`` and ``index.json`` (position in ``scene_names`` = scene_idx) -> the
    PatchDatabase over the same rows (resident in HBM, ready to be queried).  The FLANN index file the reference also writes (:49-55) has no counterpart:
    the search here is exact."""
    from .database import PatchDatabase, build_database_rows
    if len(scene_names) != len(volumes):
        raise ValueError('one scene name per scene chunk (%d names, %d volumes)' % (len(scene_names), len(volumes)))
    emb, meta = build_database_rows(config, fenc_target, volumes, device, patch_mask)
    save_database(tree_path, meta.cpu().numpy(), emb.cpu().numpy(), scene_names)
    return PatchDatabase(emb, meta, volumes, device, half_store=half_store)


def _per_chunk(scene_names, patch_size, context):
    return [n for scene in scene_names for n in chunk_patch_names(scene, patch_size, context)]


def retrieval_mapping(database, q, scene_names, K, index=None, ignore_patches_from_source=False, patch_mask=None, patch_size=16, context=8):
    """``RetrievalInterface.get_retrieval_mapping`` after the feature extraction (util/retrieval.py:184-187 -> :127-135 -> flann_knn_worker :87-100), on the
    device: ``q`` [n*64, latent] unit query embeddings of the 64 patches of each of the n chunks ``scene_names`` (device tensor, the engine's
    ``embed_queries`` order = the reference's extent enumeration) -> top-2K -> same-scene demotion when ``ignore_patches_from_source`` and the query's scene
    is in ``index`` (the database's index.json list; :94-97) -> first K -> {patch name: [K,8] float32 rows [scene_idx, box6, dist]}.
    ``patch_mask`` [n,64] bool: the dataset's occupancy filter -- dropped patches are no dataset items, so they are absent from the mapping
    (dataset/patched_scene_dataset.py:28-32)."""
    import torch
    n = len(scene_names)
    if q.shape[0] != n * 64:
        raise ValueError('%d query rows for %d chunks of 64 patches' % (q.shape[0], n))
    query_scene = None
    if ignore_patches_from_source:
        if index is None:
            raise ValueError('ignore_patches_from_source needs the database index (the list of index.json)')
        pos = {name: i for i, name in reversed(list(enumerate(index)))}          # list.index semantics: the first occurrence
        per_chunk = np.array([pos.get(s, -1) for s in scene_names], dtype=np.int32)
        query_scene = torch.from_numpy(np.repeat(per_chunk, 64)).to(q.device)
    keep = None
    if patch_mask is not None:
        keep = torch.as_tensor(patch_mask).reshape(-1).to(q.device, torch.bool).contiguous()
    meta, dist, _ = database.retrieve(q, K, query_scene, keep)
    database.check()
    names = _per_chunk(scene_names, patch_size, context)
    if len(set(names)) != len(names):
        raise ValueError('retrieval_mapping: a scene chunk is named twice (patch names are the keys of the mapping; ADVICE r5)')
    meta_h, dist_h = meta.cpu().numpy(), dist.cpu().numpy()
    if patch_mask is None:
        return mapping_to_dict(names, meta_h, dist_h)
    flags = np.asarray(torch.as_tensor(patch_mask).cpu()).reshape(-1).astype(bool)
    kept = [i for i in range(len(names)) if flags[i]]               # by index: the mask and the rows are aligned with `names`, not with a dict's order
    return mapping_to_dict([names[i] for i in kept], meta_h[kept], dist_h[kept])


def scene_patch_extents(size, patch_size=16, context=8, stride=16):
    """Padded target extents [P, 6] of a scene's patches in the reference's enumeration order (dataset/scene.py:152-160 get_extents_for_size)."""
    axes = [np.linspace(0, s - patch_size, (s - patch_size) // stride + 1).astype(np.int32) for s in size]
    x, y, z = np.meshgrid(*axes, indexing='ij')
    w = patch_size + 2 * context
    return np.stack([x.ravel(), x.ravel() + w, y.ravel(), y.ravel() + w, z.ravel(), z.ravel() + w], axis=1).astype(np.int32)


def compose_scene(database, mapping, scene, K, trunc_fill, trunc_ratio=1.0, patch_size=16, context=8, no_overlap=True, stride=None, size=(64, 64, 64),
                  patch_names=None):
    """``create_retrieval_from_mapping`` (util/retrieval.py:145-164) for one 64^3 scene chunk on the device: the K retrieval volumes [K,64,64,64] float32
    (numpy), raw values (times ``trunc_ratio`` = query trunc / database trunc, :159).  Patches absent from ``mapping`` (occupancy filter) and sentinel hits
    (scene_idx < 0, :157-158) keep the truncation fill (:148).

    ``no_overlap=False`` (a patch grid with ``stride`` < ``patch_size``, ``dataset.no_overlap`` False): the branch at :156 -- the scene's patches are visited in
    the order of ``patch_names`` (default: the reference's enumeration of the grid, restricted to the names ``mapping`` holds) and a patch overwrites its box of
    retrieval k only while the mean distance stored in that box is above its own (``rfuse.ops.compose_overlap``)."""
    import torch
    from . import ops
    if not no_overlap:
        stride = patch_size if stride is None else stride
        if patch_names is None:
            patch_names = [patch_name(scene, tuple(int(v) for v in e)) for e in scene_patch_extents(size, patch_size, context, stride)]
        names = [n for n in patch_names if n in mapping and mapping[n] is not None]
        if not names:
            return np.full((K,) + tuple(size), np.float32(trunc_fill), dtype=np.float32)
        rows = np.stack([np.asarray(mapping[n], dtype=np.float32)[:K, :8] for n in names])
        ext = np.array([[int(v) for v in n.split('--')[1].split('_')] for n in names], dtype=np.int32)
        boxes = ext.copy()
        boxes[:, 1::2] -= 2 * context                              # unpad (dataset/patched_scene_dataset.py:103-107): the padded extent's target box in the scene
        if (boxes[:, None, 1::2] - boxes[:, None, 0::2] != (rows[:, :, 2:7:2] - rows[:, :, 1:6:2]).astype(np.int32)).any():
            raise ValueError('compose_scene: a patch box and its database box differ in size')
        if (boxes[:, 0::2] < 0).any() or (boxes[:, 1::2] > np.asarray(size)[None, :]).any():
            raise ValueError('compose_scene: a patch box leaves the scene')
        out = ops.compose_overlap(database.volumes, torch.from_numpy(rows).to(database.device), torch.from_numpy(boxes).to(database.device), K, size, trunc_fill,
                                  trunc_ratio)
        return out.cpu().numpy()
    names = chunk_patch_names(scene, patch_size, context)
    rows = np.zeros((64, K, 7), dtype=np.int32)
    rows[:, :, 0] = -1
    for p, nm in enumerate(names):
        if nm in mapping and mapping[nm] is not None:
            rows[p] = np.asarray(mapping[nm])[:K, :7].astype(np.int32)
    meta = torch.from_numpy(rows).to(database.device)
    out = ops.gather_patches(database.volumes, meta, 1, K, trunc_fill, trunc_ratio, 0.0, 1.0, layout=0, no_overlap=no_overlap)
    vols = out[0].cpu().numpy()
    if trunc_ratio != 1.0:
        # patches the mapping does not hold were never visited by the reference's loop (:151): they keep the initial fill (:148), unscaled -- unlike sentinel
        # hits, which are scaled like any patch (:160-162)
        o = range(0, 64 - patch_size + 1, patch_size)
        for p, (xx, yy, zz) in enumerate((x, y, z) for x in o for y in o for z in o):
            if names[p] not in mapping or mapping[names[p]] is None:
                vols[:, xx:xx + patch_size, yy:yy + patch_size, zz:zz + patch_size] = np.float32(trunc_fill)
    return vols


def retrievals_to_disk(mode, engine, retrievals_dir, splits, index=None, batch=32, use_target_for_feats=False, fenc_target=None, target_chunks=None,
                       truncations_by_split=None, no_overlap=True, patch_stride=None):
    """The reference's ``retrievals_to_disk`` (util/retrieval.py:210-248) driven by the device path.

    ``splits``: {'train': (scene_names, input_chunks[, patch_mask]), 'val': (...)} -- ``input_chunks`` [n,S,S,S] raw low-resolution chunks (one per scene
    name: a dataset "scene" is a 64^3 chunk, dataset/scene.py), ``patch_mask`` [n,64] the occupancy filter or None.  ``engine``: a RefinementEngine whose
    ``fenc_input`` holds the retrieval checkpoint and whose ``database`` is the dictionary (``create_dictionary`` / ``PatchDatabase``); ``index``: the
    database's scene-name list (index.json), needed for the train split's same-scene demotion.

      mode 'map'      query embeddings -> ``retrieval_mapping`` -> ``map_train.npy`` with ignore_patches_from_source=True (:233-235), ``map_val.npy`` with
                      False (:236-238).  Queries: ``engine.embed_queries`` of the input chunks, or -- ``use_target_for_feats`` (:230-231: ``fenc_target`` /
                      ``extract_target_features``) -- the TARGET windows of ``target_chunks[split]`` ([n,64,64,64] raw) embedded by ``fenc_target``, i.e. the
                      rows ``create_dictionary`` would give those chunks
      mode 'compose'  reads the two map files back and writes ``compose/<scene>.npz`` for every scene of both splits (:239-248).  Fill value and scale follow the
                      reference (:148,159): a split's volumes are filled with THAT split's target truncation and database patches are scaled by split
                      truncation / train truncation -- ``truncations_by_split`` {'train': t, 'val': t} when the two datasets differ (default: the config's)

    ``no_overlap`` / ``patch_stride``: the dataset's ``no_overlap`` flag (dataset/patched_scene_dataset.py:113-115) and target patch stride for mode 'compose' -- with
    overlapping patches the branch at util/retrieval.py:156 decides patch by patch, in order (``compose_scene(no_overlap=False)``).

    -> the list of files written."""
    import torch
    cfg = engine.config
    K = cfg['K']
    g = cfg['query_geometry']
    ps, ctx = g['patch_size_target'], g['patch_context_target']
    retrievals_dir = Path(retrievals_dir)
    written = []
    if mode == 'map':
        retrievals_dir.mkdir(parents=True, exist_ok=True)
        for split, ignore in (('train', True), ('val', False)):
            if split not in splits:
                continue
            names, chunks = splits[split][0], np.asarray(splits[split][1], dtype=np.float32)
            mask = splits[split][2] if len(splits[split]) > 2 else None
            mapping = {}
            for lo in range(0, len(names), batch):
                hi = min(lo + batch, len(names))
                if use_target_for_feats:
                    if fenc_target is None or target_chunks is None or split not in target_chunks:
                        raise ValueError('use_target_for_feats needs fenc_target and target_chunks[%r] (the 64^3 target chunks of the split)' % split)
                    from .database import build_database_rows
                    q = build_database_rows(cfg, fenc_target, np.asarray(target_chunks[split], dtype=np.float32)[lo:hi], engine.device)[0][:-1]      # (no sentinel row)
                else:
                    q = engine.embed_queries(torch.from_numpy(chunks[lo:hi]).to(engine.device))
                mapping.update(retrieval_mapping(engine.database, q, names[lo:hi], K, index, ignore, None if mask is None else np.asarray(mask)[lo:hi], ps, ctx))
            save_mapping(retrievals_dir / ('map_%s.npy' % split), mapping)
            written.append(retrievals_dir / ('map_%s.npy' % split))
    elif mode == 'compose':
        from .configs import truncations
        _, trunc_t = truncations(cfg)
        tby = dict(truncations_by_split or {})
        trunc_train = float(tby.get('train', trunc_t))              # the database's volumes are the train split's targets (dataset_train.get_scene_target, :158)
        for split in ('train', 'val'):
            if split not in splits:
                continue
            trunc_split = float(tby.get(split, trunc_t))
            mapping = load_mapping(retrievals_dir / ('map_%s.npy' % split))
            for scene in splits[split][0]:
                save_compose(retrievals_dir, scene, compose_scene(engine.database, mapping, scene, K, trunc_split, trunc_split / trunc_train, ps, ctx, no_overlap=no_overlap,
                                                                   stride=patch_stride))
                written.append(retrievals_dir / 'compose' / ('%s.npz' % scene))
    else:
        raise ValueError("mode must be 'map' or 'compose' (the reference's 'evaluate' computes IoU / Chamfer metrics: out of scope)")
    return written
