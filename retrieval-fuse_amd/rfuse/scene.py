"""Scene-level recomposition of refined chunks.

Behaviour of the reference's ``PatchedSceneDataset.combine_chunks`` / ``combine_inputs`` / ``combine_targets``
(dataset/patched_scene_dataset.py:153-180), pinned by tests/golden/combine_chunks.npz which was produced by running
that method itself: 3DFront / Matterport3D scenes are cut into 64^3 chunks whose names end in ``__<x>_<y>_<z>`` (origin
of the chunk in target voxels) and share the superscene prefix ``<scene>__<room>``; every other dataset's chunk is a
whole scene at the origin.  The canvas is float64, filled with the truncation value, as large as the farthest chunk
reaches; chunks are pasted in list order (a later chunk overwrites an earlier one at the same origin).

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
