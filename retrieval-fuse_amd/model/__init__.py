"""Drop-in ``model`` package: the five factory functions of the reference's model/__init__.py:6-61, returning the
MI355X-native modules.  Put ``retrieval-fuse_amd/`` ahead of the reference checkout on ``sys.path`` and
``from model import get_unet_backbone, ...`` (trainer/train_refinement.py:11, util/retrieval.py:14) resolves here.

Selection keys and constructor-argument order are the reference's; unknown keys return None exactly as its if/elif
chains fall through.
"""
from model.attention import AttentionBlock, PatchedAttentionBlock
from model.refinement import (Superresolution08UNetBackbone, SurfaceReconstructionUNetBackbone, Superresolution08FinalDecoder,
                              RetrievalUNetBackbone, Superresolution16UNetBackbone)
from model.retrieval import (Patch04, Patch08, Patch16, Patch24, Patch32, PCPatch32, PCPatch48, PCPatch64, Patch12, PatchNorm08,
                             PatchNorm32, Patch24V2, Patch04V2)

# reference model/__init__.py:8-23 (query side) and :24-37 (database side)
_INPUT_ENCODERS = {'2+1': Patch04, '2+1V2': Patch04V2, '4+2': Patch08, '4+2N': PatchNorm08, '16+4': Patch24,
                   'pc_16+8': PCPatch32, 'pc_32+8': PCPatch48, 'pc_32+16': PCPatch64}
_TARGET_ENCODERS = {'pc_32+16': PCPatch64, '8+2': Patch12, '8+4': Patch16, '16+4': Patch24, '16+4V2': Patch24V2,
                    '16+8': Patch32, '16+8N': PatchNorm32}


def get_retrieval_networks(model_config):
    enc_in = _INPUT_ENCODERS.get(model_config['network_input'])
    enc_tgt = _TARGET_ENCODERS.get(model_config['network_target'])
    fenc_input = enc_in(model_config['nf_input'], model_config['latent_dim']) if enc_in is not None else None
    fenc_target = enc_tgt(model_config['nf_target'], model_config['latent_dim']) if enc_tgt is not None else None
    return fenc_input, fenc_target


def get_unet_backbone(config):
    if config['task'] == 'superresolution':
        size = config['dataset_train']['input_chunk_size']
        cls = {8: Superresolution08UNetBackbone, 16: Superresolution16UNetBackbone}.get(size)
        if cls is not None:
            return cls(config['nf'], num_levels=config['unet_num_level'], layer_order=config['layer_order'])
    if config['task'] == 'surface_reconstruction':
        return SurfaceReconstructionUNetBackbone(config['nf'], num_levels=config['unet_num_level'], layer_order=config['layer_order'])
    return None


def get_decoder(config):
    return Superresolution08FinalDecoder(config['nf'], layer_order=config['layer_order'])


def get_retrieval_backbone(config):
    return RetrievalUNetBackbone(nf=config['nf'], f_maps=config['retrieval_fmaps'], num_levels=config['retrieval_num_level'],
                                 layer_order=config['layer_order'])


def get_attention_block(config):
    e = config['attn_patch_extent'] // 2
    block = AttentionBlock(config['nf'], e, config['K'], config['attn_normalize'], config['attn_use_switching'],
                           config['attn_retrieval_mode'], config['attn_no_output_mapping'], config['attn_blend'])
    return PatchedAttentionBlock(config['nf'], config['attn_num_patch'], e, config['K'], block)
