"""CPU: the drop-in boundary -- the C-ABI library loads and exports every symbol include/rfuse.h declares, the
Python class surface mirrors the reference's (factory keys, state_dict keys/shapes), and the product path refuses
to run without the GPU (no fallback)."""
import contextlib
import io
import re
from pathlib import Path

import pytest
import torch

from rfuse import configs as rf_configs

REPO = Path(__file__).resolve().parents[1]


def declared_symbols():
    text = (REPO / 'include' / 'rfuse.h').read_text()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(rf_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from rfuse import _lib
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/rfuse.h but not exported by librfuse_hip.so'
    assert set(names) == set(_lib.SIGNATURES), 'ctypes signature table out of sync with the header'
    assert lib.rf_abi_version() == 1
    assert lib.rf_conv3_packed_floats(56, 96) == 27 * 96 * 64
    assert lib.rf_linear_packed_floats(32, 126) == 128 * 32


def test_library_holds_no_unsafe_packed_fp32_instruction():
    """packed-fp32 VALU instructions with op_sel set on src1 / src2 return wrong results on gfx950 while another wave's F16 MFMA runs
    on the SIMD; hipcc emits them freely.  The shipped library must not contain one (the build refuses, this re-checks the file that ships) --
    and the detector itself must recognise the forms that were measured to fail and leave the measured-safe ones alone."""
    import sys
    sys.path.insert(0, str(REPO / 'retrieval-fuse_amd' / 'csrc'))
    import build as product_build
    bad = '''
        v_pk_fma_f32 v[60:61], v[40:41], v[58:59], v[58:59] op_sel:[0,0,1] op_sel_hi:[1,0,1]
        v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[0,1,0] op_sel_hi:[1,1,0]
        v_pk_add_f32 v[102:103], v[102:103], v[102:103] op_sel:[0,1] op_sel_hi:[1,0]
        v_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]'''
    good = '''
        v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7]
        v_pk_fma_f32 v[0:1], v[2:3], s[4:5], v[6:7] op_sel:[1,0,0]
        v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel_hi:[1,0,1]
        v_pk_add_f32 v[40:41], v[54:55], v[38:39] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]
        v_pk_fma_f16 v0, v1, v2, v3 op_sel:[0,1,0]'''
    assert len(product_build.unsafe_packed_fp32(bad)) == 4
    assert product_build.unsafe_packed_fp32(good) == []
    from rfuse import _lib
    assert product_build.check_isa(_lib.LIB_PATH) >= 10              # code objects scanned; raises on a finding


def build(cfg):
    import model
    with contextlib.redirect_stdout(io.StringIO()):
        return {'unet_backbone': model.get_unet_backbone(cfg), 'decoder': model.get_decoder(cfg),
                'retrieval_backbone': model.get_retrieval_backbone(cfg), 'patched_attention_block': model.get_attention_block(cfg)}


def test_state_dict_contract_c1():
    """SURVEY.md Appendix B (probed on the reference): tensor counts, parameter totals, key names."""
    m = build(rf_configs.get_config('C1'))
    counts = {k: (len(v.state_dict()), sum(t.numel() for t in v.state_dict().values())) for k, v in m.items()}
    assert counts == {'unet_backbone': (54, 1138858), 'decoder': (8, 13905), 'retrieval_backbone': (36, 1052122),
                      'patched_attention_block': (18, 107330)}
    sd = m['retrieval_backbone'].state_dict()
    assert tuple(sd['network.decoders.1.basic_module.SingleConv1.conv.weight'].shape) == (56, 96, 3, 3, 3)
    assert tuple(sd['network.decoders.1.basic_module.SingleConv2.conv.weight'].shape) == (16, 56, 3, 3, 3)
    assert 'network.0.decoders.2.basic_module.SingleConv2.groupnorm.bias' in m['unet_backbone'].state_dict()
    assert 'network.2.basic_module.SingleConv1.conv.weight' in m['unet_backbone'].state_dict()
    dsd = m['decoder'].state_dict()
    assert tuple(dsd['network.1.weight'].shape) == (1, 16, 1, 1, 1) and tuple(dsd['network.1.bias'].shape) == (1,)
    psd = m['patched_attention_block'].state_dict()
    assert list(psd)[:2] == ['attention_blocks_layer.sig_scale', 'attention_blocks_layer.sig_shift']
    assert tuple(psd['attention_blocks_layer.phi.encoder.6.weight'].shape) == (32, 128)
    assert m['retrieval_backbone'].nf == 16


def test_retrieval_network_factory_keys():
    import model
    fi, ft = model.get_retrieval_networks({'network_input': '2+1', 'network_target': '16+8', 'nf_input': 32, 'nf_target': 8, 'latent_dim': 64})
    assert type(fi).__name__ == 'Patch04' and type(ft).__name__ == 'Patch32'
    assert sum(p.numel() for p in fi.parameters()) == 320704 and sum(p.numel() for p in ft.parameters()) == 450720
    assert [tuple(v.shape) for k, v in ft.state_dict().items() if k.endswith('weight')] == \
        [(8, 1, 5, 5, 5), (16, 8, 3, 3, 3), (32, 16, 3, 3, 3), (64, 32, 3, 3, 3), (64, 64, 3, 3, 3), (64, 64, 4, 4, 4), (64, 64)]
    for key_in in ('2+1V2', '4+2', '4+2N', '16+4', 'pc_16+8', 'pc_32+8', 'pc_32+16'):
        assert model.get_retrieval_networks({'network_input': key_in, 'network_target': 'x', 'nf_input': 4, 'nf_target': 4, 'latent_dim': 8})[0] is not None
    for key_t in ('pc_32+16', '8+2', '8+4', '16+4', '16+4V2', '16+8', '16+8N'):
        assert model.get_retrieval_networks({'network_input': 'x', 'network_target': key_t, 'nf_input': 4, 'nf_target': 4, 'latent_dim': 8})[1] is not None
    bn = model.get_retrieval_networks({'network_input': '4+2N', 'network_target': 'x', 'nf_input': 4, 'nf_target': 4, 'latent_dim': 8})[0]
    assert 'layers.1.running_mean' in bn.state_dict() and 'layers.1.num_batches_tracked' in bn.state_dict()


@pytest.mark.parametrize('name', ['C3', 'C4', 'C5'])
def test_other_configs_construct(name):
    cfg = rf_configs.get_config(name)
    m = build(cfg)
    if name == 'C5':
        sd = m['unet_backbone'].state_dict()
        assert tuple(sd['network.decoders.1.basic_module.SingleConv1.conv.weight'].shape) == (78, 144, 3, 3, 3)
        assert tuple(sd['network.decoders.1.basic_module.SingleConv2.conv.weight'].shape) == (12, 78, 3, 3, 3)
        assert tuple(m['patched_attention_block'].state_dict()['attention_blocks_layer.theta.encoder.0.weight'].shape) == (128, 96)


def test_product_path_has_no_cpu_fallback():
    m = build(rf_configs.get_config('C1'))
    with torch.no_grad():
        with pytest.raises(RuntimeError, match='no CPU fallback'):
            m['decoder'](torch.zeros(1, 16, 32, 32, 32))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m['decoder'](torch.zeros(1, 16, 32, 32, 32))          # grad mode: the autograd functions run the same GPU-only kernels
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m['patched_attention_block'](torch.zeros(1, 16, 32, 32, 32), torch.zeros(4, 16, 32, 32, 32))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from rfuse import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', tmp_path / 'nope.so')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _lib.load()


def test_product_never_imports_oracle():
    pkg = REPO / 'retrieval-fuse_amd'
    for f in pkg.rglob('*.py'):
        text = f.read_text()
        assert 'import oracle' not in text and 'from oracle' not in text and '/root/reference' not in text, f
