"""Scene-level recomposition of refined chunks.

Behaviour of the reference's ``PatchedSceneDataset.combine_chunks`` / ``combine_inputs`` / ``combine_targets``
(dataset/patched_scene_dataset.py:153-180), pinned by tests/golden/combine_chunks.npz which was produced by running
that method itself: 3DFront / Matterport3D scenes are cut into 64^3 chunks whose names end in ``__<x>_<y>_<z>`` (origin
of the chunk in target voxels) and share the superscene prefix ``<scene>__<room>``; every other dataset's chunk is a
whole scene at the origin.  The canvas is float64, filled with the truncation value, as large as the farthest chunk
reaches; chunks are pasted in list order (a later chunk overwrites an earlier one at the same origin).

``refine_scene`` is the scene-level driver around the hot path: what the reference's inference loop does per visualisation dataset
(trainer/train_refinement.py:158-169: batches of chunks -> forward_full -> network_pred_to_df -> .cpu().half() -> combine_retrievals(.., 0)).

Host-side numpy scatter, not on the hot path.  Mesh export (util/visualization.py: marching cubes -> .obj) has no
counterpart here: it needs the ``marching_cubes`` / ``trimesh`` packages, which this image does not have.
"""
import numpy as np

_TILED_DATASETS = ('Matterport3D', '3DFront')


def superscene_and_position(chunk_name, dataset_name):
    """'<scene>__<room>__<x>_<y>_<z>' -> ('<scene>__<room>', int64[3]) for the tiled datasets, else (chunk_name, zeros)."""
    if not dataset_name.startswith(_TILED_DATASETS):
        return chunk_name, np.zeros(3, dtype=np.int64)
    fields = chunk_name.split('__')
    origin = np.array([int(t) for t in fields[-1].split('_')], dtype=np.int64)
    return '__'.join(fields[:2]), origin


def combine_chunks(chunk_names, chunk_volumes, dataset_name, scale_factor=1, chunk_size=64, trunc_val=0.0):
    """-> {superscene: float64 volume}.  ``chunk_volumes[i]`` is the cubic volume of ``chunk_names[i]`` (e.g. a refined df from
    RefinementEngine.refine squeezed to [64,64,64]); its origin is the name's position divided by ``scale_factor`` and
    truncated to int32 (so low-resolution inputs are combined with scale_factor = 64 / input_chunk_size, chunk_size =
    input_chunk_size)."""
    members = {}
    for i, name in enumerate(chunk_names):
        key, origin = superscene_and_position(name, dataset_name)
        members.setdefault(key, []).append((i, (origin / scale_factor).astype(np.int32)))
    scenes = {}
    for key, items in members.items():
        reach = np.max([o for _, o in items], axis=0) + chunk_size
        canvas = np.full(tuple(int(r) for r in reach), trunc_val, dtype=np.float64)
        for i, (x, y, z) in items:
            vol = np.asarray(chunk_volumes[i])
            e = vol.shape[0]
            canvas[x:x + e, y:y + e, z:z + e] = vol
        scenes[key] = canvas
    return scenes


def split_scene(scene_input, input_chunk_size, scene_name, room='room0', pad_value=0.0):
    """Cut a low-resolution scene volume [X, Y, Z] into the chunk grid the tiled datasets use: -> (names, chunks [n, s, s, s] float32).
    Names are '<scene>__<room>__<x>_<y>_<z>' with the chunk origin in TARGET voxels (64 per chunk), i.e. what ``superscene_and_position`` parses;
    a volume that is not a whole number of chunks is padded with ``pad_value`` (the input truncation) at its far faces."""
    vol = np.asarray(scene_input, dtype=np.float32)
    s = int(input_chunk_size)
    grid = [-(-d // s) for d in vol.shape]
    padded = np.full([g * s for g in grid], pad_value, dtype=np.float32)
    padded[:vol.shape[0], :vol.shape[1], :vol.shape[2]] = vol
    names, chunks = [], []
    for ix in range(grid[0]):
        for iy in range(grid[1]):
            for iz in range(grid[2]):
                names.append('%s__%s__%d_%d_%d' % (scene_name, room, ix * 64, iy * 64, iz * 64))
                chunks.append(padded[ix * s:(ix + 1) * s, iy * s:(iy + 1) * s, iz * s:(iz + 1) * s])
    return names, np.stack(chunks)


def combine_predictions(chunk_names, predictions, dataset_name, trunc_val):
    """The reference's ``combine_retrievals(predictions, 0)`` (dataset/patched_scene_dataset.py:182-186): ``predictions`` [n, K or 1, 64, 64, 64]
    in the order of ``chunk_names``; slot 0 of every chunk is pasted at the chunk's origin on a canvas filled with the target truncation."""
    preds = np.asarray(predictions)
    return combine_chunks(chunk_names, [preds[i, 0] for i in range(len(chunk_names))], dataset_name, 1, 64, trunc_val)


def refine_scene(engine, chunk_names, chunk_inputs, batch=32, query_scene=None, patch_mask=None, half=True):
    """Scene-level inference: low-resolution chunks of one or several superscenes -> {superscene: refined TSDF volume (float64)}.

    chunk_inputs [n, s, s, s] raw (un-normalised) low-resolution chunks, ``chunk_names[i]`` the dataset's chunk name (it carries the position).
    Chunks run through ``engine.refine`` in batches of ``batch`` (the last one ragged); predictions come back as float16 like the reference's
    ``network_pred_to_df(pred_shape).cpu().half()`` (``half=False`` keeps fp32) and are pasted by ``combine_predictions``.  The device -> host
    copy of batch i overlaps the refinement of batch i + 1 (pinned double buffer on a copy stream)."""
    import torch
    cfg = engine.config
    n = len(chunk_names)
    x = torch.as_tensor(np.asarray(chunk_inputs, dtype=np.float32))
    assert x.shape[0] == n, 'one input chunk per name'
    dev = engine.device
    out_dtype = torch.float16 if half else torch.float32
    host = torch.empty((n, 1, 64, 64, 64), dtype=out_dtype).pin_memory() if dev.type == 'cuda' else torch.empty((n, 1, 64, 64, 64), dtype=out_dtype)
    copy = torch.cuda.Stream(dev)
    main = torch.cuda.current_stream(dev)
    for lo in range(0, n, batch):
        hi = min(lo + batch, n)
        qs = query_scene[lo * 64:hi * 64] if query_scene is not None else None
        pm = patch_mask[lo:hi] if patch_mask is not None else None
        df = engine.refine(x[lo:hi].to(dev, non_blocking=True), query_scene=qs, patch_mask=pm)
        df = df.to(out_dtype)
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(copy):
            copy.wait_event(ready)
            host[lo:hi].copy_(df, non_blocking=True)
            df.record_stream(copy)
    copy.synchronize()
    return combine_predictions(chunk_names, host.numpy(), cfg['dataset_train']['dataset_name'], float(engine.target_trunc))
