"""Scene-level recomposition of refined chunks.

Restates ``PatchedSceneDataset.combine_chunks`` and ``get_superscene_name_and_position_from_chunk``
(reference dataset/patched_scene_dataset.py:153-174): 3DFront / Matterport3D scenes are cut into 64^3 chunks named
``<scene>__<room>__<x>_<y>_<z>`` (position in target voxels); ShapeNet chunks are whole scenes at the origin.
Host-side numpy scatter, not on the hot path.
"""
from collections import defaultdict

import numpy as np


def superscene_and_position(chunk_name, dataset_name):
    """-> (superscene name, int position[3]) -- dataset/patched_scene_dataset.py:153-158"""
    if dataset_name.startswith('Matterport3D') or dataset_name.startswith('3DFront'):
        name = "__".join(chunk_name.split('__')[:2])
        position = [int(x) for x in chunk_name.split('__')[-1].split('_')]
        return name, np.array(position)
    return chunk_name, np.array([0, 0, 0])


def combine_chunks(chunk_names, chunk_volumes, dataset_name, scale_factor=1, chunk_size=64, trunc_val=0.0):
    """{superscene: float64 volume}: each chunk pasted at position/scale_factor into a trunc-filled canvas sized to the
    farthest chunk (dataset/patched_scene_dataset.py:160-174).  ``chunk_volumes[i]`` is the cubic volume of ``chunk_names[i]``
    (e.g. the refined df from RefinementEngine.refine, squeezed to [64,64,64])."""
    groups = defaultdict(list)
    for i, s in enumerate(chunk_names):
        name, position = superscene_and_position(s, dataset_name)
        groups[name].append((i, (position / scale_factor).astype(np.int32)))
    result = {}
    for ss, items in groups.items():
        positions = np.vstack([p for _, p in items])
        combined = np.ones([positions[:, 0].max() + chunk_size, positions[:, 1].max() + chunk_size, positions[:, 2].max() + chunk_size]) * trunc_val
        for i, p in items:
            v = np.asarray(chunk_volumes[i])
            e = v.shape[0]
            combined[p[0]:p[0] + e, p[1]:p[1] + e, p[2]:p[2] + e] = v
        result[ss] = combined
    return result
