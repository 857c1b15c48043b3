"""The five BASELINE.json workloads as plain dicts.

Key names are the reference's YAML keys (after its one-level ``inherit_from`` merge and the
``dataset:`` -> ``dataset_train`` fan-out, reference config/config_handler.py:5-31) so the factory
functions in ``model/__init__.py`` can index them exactly like the reference factories do
(reference model/__init__.py:41-61).  Only hot-path keys are kept; the values are copied from

  C1/C2  config/super_resolution/ShapeNetV2/refinement_008_064.yaml:3-32 (+ retrieval_008_064.yaml)
  C3     config/super_resolution/3DFront/refinement_008_064.yaml:3-29
  C4     config/super_resolution/Matterport3D/refinement_016_064.yaml:3-35, retrieval_016_064.yaml:3-40
  C5     config/surface_reconstruction/ShapeNetV2/refinement_128_064.yaml:3-27, retrieval_128_064.yaml:2-19
  base   config/base/refinement_superresolution.yaml, retrieval_superresolution.yaml,
         refinement_surface_reconstruction.yaml, retrieval_surface_reconstruction.yaml
"""
import copy

import numpy as np


def _f16(x):
    # reference dataset/scene.py:30-33: voxel sizes and truncation pass through float16
    return float(np.float16(x).astype(np.float32))


def _dataset(name, vin, vtgt, imean, istd, tmean, tstd, input_chunk_size=8, num_points=0):
    return {
        'dataset_name': name, 'voxel_size_input': vin, 'voxel_size_target': vtgt,
        'input_mean': imean, 'input_std': istd, 'target_mean': tmean, 'target_std': tstd,
        'input_chunk_size': input_chunk_size, 'target_chunk_size': 64, 'num_points': num_points,
    }


def _attn(retrieval_mode):
    return {
        'attn_normalize': True, 'attn_use_switching': True, 'attn_retrieval_mode': retrieval_mode,
        'attn_no_output_mapping': True, 'attn_blend': True, 'attn_patch_extent': 4, 'attn_num_patch': 16,
    }


_SHAPENET = _dataset('ShapeNetV2', 0.166667, 0.020834, 0.3095340441938771, 0.14730652990291243,
                     0.059954833543534335, 0.010110036361741626)
_FRONT = _dataset('3DFront', 0.43334, 0.054167, 0.8112343966484424, 0.5094238937427482,
                  0.15015658121788053, 0.03573221820637578)
_MATTERPORT = _dataset('Matterport3D16', 15.0, 3.75, 35.62394659115317, 14.58642912987053,
                       10.502049923464249, 2.3319665041587627, input_chunk_size=16)
_SHAPENET_PC = _dataset('ShapeNetV2', 0, 0.020834, 0.0, 1.0, 0.059954833543534335, 0.010110036361741626,
                        input_chunk_size=128, num_points=500)

_SR_RETRIEVAL = {'network_input': '2+1', 'network_target': '16+8', 'nf_input': 32, 'nf_target': 8, 'latent_dim': 64}
# query-side window geometry (reference config/base/retrieval_superresolution.yaml:9-14)
_SR_QUERY = {'patch_size_input': 2, 'patch_context_input': 1, 'patch_size_target': 16, 'patch_context_target': 8}


def _cfg(task, nf, unet_levels, K, dataset, retrieval_mode, retrieval_model, query_geom, db_patches):
    c = {
        'task': task, 'nf': nf, 'unet_num_level': unet_levels, 'layer_order': 'gcr',
        'retrieval_fmaps': nf, 'retrieval_num_level': 4, 'K': K,
        'dataset_train': dict(dataset), 'dataset_val': dict(dataset),
        'retrieval_model': dict(retrieval_model), 'query_geometry': dict(query_geom),
        'db_patches': db_patches,
    }
    c.update(_attn(retrieval_mode))
    return c


CONFIGS = {
    # BASELINE.json configs[0]/[1]: ShapeNetV2 008->064, K=4, gumbel-hard attention, DB = 50k patches
    'C1': _cfg('superresolution', 16, 4, 4, _SHAPENET, True, _SR_RETRIEVAL, _SR_QUERY, 50_000),
    'C2': _cfg('superresolution', 16, 4, 4, _SHAPENET, True, _SR_RETRIEVAL, _SR_QUERY, 50_000),
    # configs[2]: 3DFront 008->064, softmax attention, DB = 1M patches sharded 8 ways
    'C3': _cfg('superresolution', 16, 4, 4, _FRONT, False, _SR_RETRIEVAL, _SR_QUERY, 1_000_000),
    # configs[3]: Matterport3D 016->064, K=8 attention stress
    'C4': _cfg('superresolution', 16, 4, 8, _MATTERPORT, False,
               {'network_input': '4+2', 'network_target': '16+8', 'nf_input': 16, 'nf_target': 8, 'latent_dim': 64},
               {'patch_size_input': 4, 'patch_context_input': 2, 'patch_size_target': 16, 'patch_context_target': 8},
               50_000),
    # configs[4]: ShapeNetV2 surface reconstruction pc -> 064 (128^3 occupancy grid input)
    'C5': _cfg('surface_reconstruction', 12, 5, 4, _SHAPENET_PC, True,
               {'network_input': 'pc_32+8', 'network_target': '16+4V2', 'nf_input': 12, 'nf_target': 12, 'latent_dim': 64},
               {'patch_size_input': 32, 'patch_context_input': 8, 'patch_size_target': 16, 'patch_context_target': 4},
               50_000),
}


def get_config(name):
    return copy.deepcopy(CONFIGS[name])


def truncations(config):
    """(input_trunc, target_trunc): 3 voxels, rounded through float16 (reference dataset/scene.py:32-33)."""
    d = config['dataset_train']
    return _f16(d['voxel_size_input'] * 3), _f16(d['voxel_size_target'] * 3)
