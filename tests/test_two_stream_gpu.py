"""Bit-reproducibility of every kernel family beside a kernel of another stream that issues F16 MFMAs.

Root cause of round 2's "two-stream hazard" (tools/pkfma_probe.py, profiles/r03_pkfma_probe.log): on gfx950 a packed-fp32 VALU instruction
(v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) whose src1 takes its low-lane operand from the HIGH register of the pair (op_sel bit set) reads
that operand as ZERO in lanes 48..63 while another wave's F16 MFMA executes on the same SIMD.  hipcc had emitted such forms in the GroupNorm
apply of the fp32 conv kernels and in the final add of the exact top-k distance.  The build now refuses them (csrc/build.py:check_isa,
tests/test_boundary.py); this file is the run-time side: each kernel family runs 60 times on a side stream beside `rft_f16_mfma_load`
(tests/testkit: one 154-VGPR F16-MFMA workgroup per CU, so foreign waves always find room on its SIMDs) and must return its solo bits.
"""
import numpy as np
import pytest
import torch

import helpers
import testkit
from rfuse import configs as rf_configs
from rfuse import synthetic

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('needs the GPU')
    from rfuse import ops as o
    return o


def rnd(gen, *shape, scale=1.0):
    return torch.randn(*shape, generator=gen) * scale


def affine(ops, gen, x0, x1, groups):
    c = (x0.shape[1] if x0 is not None else 0) + (x1.shape[1] if x1 is not None else 0)
    gamma = (1.0 + 0.3 * rnd(gen, c)).to(DEV)
    beta = rnd(gen, c, scale=0.5).to(DEV)                         # a dropped shift is then an error of the size of the activations
    return ops.gn_affine(x0, x1, gamma, beta, groups)


def lib_supported(ops, fn, *args):
    from rfuse import _lib
    return getattr(_lib.load(), fn)(*args)


def victims(ops):
    """name -> zero-argument launch on the CURRENT stream returning the tensor(s) to compare"""
    g = torch.Generator().manual_seed(11)
    v = {}

    def conv_case(name, n, cin, edge, cout, arith):
        x = rnd(g, n, cin, edge, edge, edge).relu_().to(DEV)
        aff = affine(ops, g, x, None, 8 if cin >= 8 else 1)
        w = rnd(g, cout, cin, 3, 3, 3, scale=0.05).to(DEV)
        if arith == 'fp32':
            wp = ops.pack_conv3_weight(w)

            def run():
                saved, ops.CONV_ARITH = ops.CONV_ARITH, 'fp32'
                try:
                    return ops.conv3d_gn_relu(x, None, aff, wp, cout)
                finally:
                    ops.CONV_ARITH = saved
        else:
            ws = ops.pack_conv3_split_weight(w)
            assert ops.conv_split_supported(x, None, cout), name

            def run():
                return ops.conv3d_split_gn_relu(x, aff, ws, cout)
        v[name] = run

    conv_case('fp32 box conv, 128-voxel tiles (32->32 @16^3 x 8)', 8, 32, 16, 32, 'fp32')
    conv_case('fp32 box conv, 8^3 boxes (16->16 @8^3 x 2048)', 2048, 16, 8, 16, 'fp32')
    conv_case('fp32 box conv, whole 4^3 samples in box tiles (32->32 @4^3 x 64)', 64, 32, 4, 32, 'fp32')
    conv_case('fp32 position-major conv (32->64 @4^3 x 4096)', 4096, 32, 4, 64, 'fp32')
    conv_case('fp32 position-major conv (64->128 @2^3 x 8192)', 8192, 64, 2, 128, 'fp32')
    conv_case('cin = 1 first conv (1->8 @16^3 x 512)', 512, 1, 16, 8, 'fp32')
    conv_case('split box conv (16->16 @8^3 x 2048)', 2048, 16, 8, 16, 'split')
    conv_case('split 4^3 conv (32->64 @4^3 x 4096)', 4096, 32, 4, 64, 'split')

    # decoder-form convs: fp32 (k_conv3_up) and split (k_conv3_up_split)
    x0 = rnd(g, 256, 32, 8, 8, 8).relu_().to(DEV)
    x1 = rnd(g, 256, 64, 4, 4, 4).relu_().to(DEV)
    affu = affine(ops, g, x0, x1, 8)
    wu = rnd(g, 56, 96, 3, 3, 3, scale=0.05).to(DEV)
    wup, wups = ops.pack_conv3_up_weight(wu, 32), ops.pack_conv3_up_split_weight(wu, 32)
    assert ops.conv_up_supported(x0, x1, 56) and ops.conv_up_split_supported(x0, x1, 56)
    v['fp32 decoder-form conv (32+64->56 @8^3 x 256)'] = lambda: ops.conv3d_up_gn_relu(x0, x1, affu, wup, 56)
    v['split decoder-form conv (32+64->56 @8^3 x 256)'] = lambda: ops.conv3d_up_split_gn_relu(x0, x1, affu, wups, 56)
    # the 64^3 decoder's first conv: upsampled source only
    xl = rnd(g, 4, 16, 32, 32, 32).relu_().to(DEV)
    affl = affine(ops, g, None, xl, 8)
    wl = rnd(g, 16, 16, 3, 3, 3, scale=0.05).to(DEV)
    if ops.conv_up_supported(None, xl, 16):
        wlp = ops.pack_conv3_up_weight(wl, 0)
        v['fp32 decoder-form conv on an upsampled source (0+16->16 @64^3 x 4)'] = lambda: ops.conv3d_up_gn_relu(None, xl, affl, wlp, 16)
    if ops.conv_up_split_supported(None, xl, 16):
        wls = ops.pack_conv3_up_split_weight(wl, 0)
        v['split decoder-form conv on boxes of an upsampled source (0+16->16 @64^3 x 4)'] = lambda: ops.conv3d_up_split_gn_relu(None, xl, affl, wls, 16)
    # boxes with a skip source (C5's U-Net decoder shape class)
    xs0 = rnd(g, 16, 24, 32, 32, 32).relu_().to(DEV)
    xs1 = rnd(g, 16, 48, 16, 16, 16).relu_().to(DEV)
    affs = affine(ops, g, xs0, xs1, 6)
    wsk = rnd(g, 42, 72, 3, 3, 3, scale=0.05).to(DEV)
    if ops.conv_up_split_supported(xs0, xs1, 42):
        wsks = ops.pack_conv3_up_split_weight(wsk, 24)
        v['split decoder-form conv on boxes with a skip source (24+48->42 @32^3 x 16)'] = lambda: ops.conv3d_up_split_gn_relu(xs0, xs1, affs, wsks, 42)
    wlg = ops.pack_conv3_weight(wl)

    def generic_up():
        saved, ops.CONV_ARITH = ops.CONV_ARITH, 'fp32'
        try:
            return ops.conv3d_gn_relu(None, xl, affl, wlg, 16)
        finally:
            ops.CONV_ARITH = saved
    v['fp32 box conv reading an upsampled source (0+16->16 @64^3 x 4)'] = generic_up

    # GroupNorm statistics, max-pool, 1x1 conv + tanh
    xg = rnd(g, 64, 32, 16, 16, 16).to(DEV)
    gam, bet = (1 + 0.2 * rnd(g, 32)).to(DEV), rnd(g, 32).to(DEV)
    v['GroupNorm statistics (64 x 32 x 16^3)'] = lambda: ops.gn_affine(xg, None, gam, bet, 8)
    v['max-pool (64 x 32 x 16^3)'] = lambda: ops.maxpool2(xg)
    w1, b1 = rnd(g, 1, 32, 1, 1, 1).to(DEV), rnd(g, 1).to(DEV)
    v['1x1 conv + tanh (64 x 32 x 16^3)'] = lambda: ops.conv1x1_tanh(xg, w1, b1, 1.0, 0.5)

    # exact top-k: VALU scan (its final add was one of the unsafe forms) and the two MFMA-filtered scans
    emb = torch.nn.functional.normalize(rnd(g, 50001, 64), dim=1).to(DEV)
    dbp = ops.db_pack_embeddings(emb)
    q = torch.nn.functional.normalize(rnd(g, 512, 64), dim=1).to(DEV)
    for algo, nm in ((ops.TOPK_VALU_SCAN, 'VALU scan'), (ops.TOPK_MFMA_SCAN, 'fp32-MFMA-filtered scan'), (ops.TOPK_MFMA16_SCAN, 'f16-MFMA-filtered scan')):
        v['exact top-8 of 50 001 rows, %s' % nm] = (lambda a: (lambda: torch.cat([t.double() for t in ops.l2_topk(q, dbp, 50001, 0, 8, algo=a)], dim=1)))(algo)

    # patch encoders' valid convs
    xv = rnd(g, 64, 12, 24, 24, 24).to(DEV)
    wv = rnd(g, 24, 12, 3, 3, 3, scale=0.07).to(DEV)
    bv = rnd(g, 24).to(DEV)
    if ops.conv_valid_valu_supported(xv, 24, 3, 1):
        wvt = ops.pack_convv_valu_weight(wv)
        v['packed-fp32 VALU valid conv (12->24 k3 @24^3 x 64)'] = lambda: ops.conv3d_valid_leaky_valu(xv, wvt, bv, 1, 0.2)
    xv2 = rnd(g, 64, 24, 22, 22, 22).to(DEV)
    wv2 = rnd(g, 24, 24, 3, 3, 3, scale=0.07).to(DEV)
    if ops.conv_valid_lds_supported(xv2, 24, 3, 2):
        wv2l = ops.pack_convv_lds_weight(wv2)
        v['fp32 MFMA valid conv, LDS-staged (24->24 k3 s2 @22^3 x 64)'] = lambda: ops.conv3d_valid_leaky_lds(xv2, wv2l, bv, 24, 3, 2, 0.2)
    wv2g = ops.pack_convv_weight(wv2)
    v['fp32 MFMA valid conv, gather form (24->24 k3 s2 @22^3 x 64)'] = lambda: ops.conv3d_valid_leaky_mfma(xv2, wv2g, bv, 24, 3, 2, 0.2)
    if ops.conv_valid_split_supported(xv2, 24, 3, 2):
        wv2s = ops.pack_convv_split_weight(wv2, 22, 2)
        v['split valid conv (24->24 k3 s2 @22^3 x 64)'] = lambda: ops.conv3d_valid_leaky_split(xv2, wv2s, bv, 24, 3, 2, 0.2)

    # the persistent two-team grid form (packed-fp32 epilogue of one team beside the other team's -- and the load's -- F16 MFMAs)
    if ops.conv_valid_split_pg_supported((2, 12, 64), 24, 3, 1):
        x_pg = rnd(g, 2, 12, 64, 64, 64).to(DEV)
        t_pg = (x_pg * (1.0 / 16)).clamp(-65504.0, 65504.0)
        h_pg = t_pg.half()
        l_pg = ((t_pg - h_pg.float()) * 2048.0).half()
        xs_pg = ops.SplitActs(torch.stack([h_pg, l_pg], 0).view(2, 2, 3, 4, 64 ** 3).permute(1, 2, 0, 4, 3).contiguous().view(torch.float32).view(2, 12, 64, 64, 64))
        wpg = ops.pack_convv_split_pg_weight(wv, 64, 1)
        v['persistent grid valid conv (12->24 k3 @64^3 x 2, split form)'] = lambda: ops.conv3d_valid_leaky_split_pg(xs_pg, wpg, bv, 24, 3, 1, 0.2).data

    # linear layers and the attention kernels
    xr = rnd(g, 4096, 128).to(DEV)
    wlin, blin = rnd(g, 128, 128, scale=0.1).to(DEV), rnd(g, 128).to(DEV)
    wlp = ops.pack_linear_weight(wlin)
    v['Linear 128->128 + LeakyReLU x 4096 rows'] = lambda: ops.linear(xr, wlp, blin, 128, ops.ACT_LEAKY, 0.01)
    params = []
    for (o_, i_) in ((128, 128), (128, 128), (128, 128), (32, 128)):
        params += [rnd(g, o_, i_, scale=0.1).to(DEV), rnd(g, o_, scale=0.1).to(DEV)]
    packed = ops.pack_attn_mlp(params)
    for arith in ('fp32', 'split'):
        def mlp(a=arith):
            saved, ops.CONV_ARITH = ops.CONV_ARITH, a
            try:
                return ops.attn_mlp_rows(xr, packed)
            finally:
                ops.CONV_ARITH = saved
        v['attention feature encoder, %s form (4096 rows)' % arith] = mlp
    K = 4
    xf, pf = rnd(g, 4096, 32).to(DEV), rnd(g, 4096 * K, 32).to(DEV)
    noise = (-torch.empty(4096, K).exponential_(generator=g).log()).to(DEV)
    def weights_case(mode, sharp, nz):
        def run():
            w_, sw_ = ops.attn_weights(xf, pf, nz, K, mode, sharp)
            return torch.cat([w_, sw_[:, None]], dim=1)
        return run
    v['attention weights, Gumbel-hard (4096 rows, K = 4)'] = weights_case(ops.ATTN_GUMBEL_HARD, 25.0, noise)
    v['attention weights, softmax (4096 rows, K = 4)'] = weights_case(ops.ATTN_SOFTMAX, 1024.0, None)
    # the training slice's backward kernels (rfuse/autograd.py): data / weight gradient on the split forms, GroupNorm backward, pooling ops
    from rfuse import autograd as rfa
    xt = rnd(g, 128, 16, 16, 16, 16).relu_().to(DEV)
    yt = rnd(g, 128, 24, 16, 16, 16).relu_().to(DEV)
    dyt = rnd(g, 128, 24, 16, 16, 16, scale=1e-3).to(DEV)
    gt, bt = (1 + 0.2 * rnd(g, 16)).to(DEV), rnd(g, 16, scale=0.2).to(DEV)
    wt_ = rnd(g, 24, 16, 3, 3, 3, scale=0.05).to(DEV)
    afft = ops.gn_affine(xt, None, gt, bt, 8)

    def backward_case(which):
        def run():
            dz, amax = rfa.relu_backward_amax(dyt, yt)
            ident, scales = rfa.dz_scale(amax, 128, 24)
            if which == 'dgrad':
                return rfa.dgrad_split(dz, ident, wt_, 16)
            if which == 'wgrad':
                return rfa.conv3d_wgrad_split(xt, afft, dz, scales, 24)
            return dz
        return run
    v['ReLU backward + max |dz| (128 x 24 x 16^3)'] = backward_case('relu')
    if bool(lib_supported(ops, 'rf_conv3d_split_k3_gn_supported', 24, 128, 16, 16)):
        v['split data-gradient conv (24->16 @16^3 x 128, scaled dz)'] = backward_case('dgrad')
    v['split weight gradient (16->24 @16^3 x 128)'] = backward_case('wgrad')
    v['fp32 weight gradient (16->24 @16^3 x 128)'] = lambda: rfa.conv3d_wgrad(xt, afft, rfa.relu_backward(dyt, yt), 24)
    dxn_t = rnd(g, 128, 16, 16, 16, 16).to(DEV)
    v['GroupNorm backward (128 x 16 x 16^3)'] = lambda: torch.cat([t.reshape(-1).double() for t in rfa.gn_backward(xt, dxn_t, gt, 8, 1e-5)])
    v['max-pool backward (128 x 16 x 16^3)'] = lambda: rfa.maxpool2_backward(xt, dyt[:, :16, ::2, ::2, ::2].contiguous())
    v['nearest x2 upsample / 2^3 sum pool (128 x 16 x 16^3)'] = lambda: rfa.sumpool2(rfa.upsample2(xt)) + rfa.sumpool2(xt).mean()
    xb = rnd(g, 2, 16, 32, 32, 32).to(DEV)
    rb = rnd(g, 2 * K, 16, 32, 32, 32).to(DEV)
    wts = torch.softmax(rnd(g, 2 * 4096, K), dim=1).to(DEV)
    sw = torch.rand(2 * 4096, generator=g).to(DEV)
    v['attention blend (2 x 16 x 32^3, K = 4)'] = lambda: ops.attn_blend(xb, rb, K, 32, wts, sw)
    return v


def test_every_kernel_family_keeps_its_bits_beside_f16_mfma(ops):
    fams = victims(ops)
    main, side = torch.cuda.current_stream(), torch.cuda.Stream(DEV)
    scratch = torch.empty(256 * 256, device=DEV)
    report, failed = [], []
    for name, run in fams.items():
        ref = run().clone()
        torch.cuda.synchronize()
        assert torch.isfinite(ref).all(), name
        bad = 0
        for _ in range(3):
            outs = []
            side.wait_stream(main)
            testkit.f16_mfma_load(main, scratch)
            with torch.cuda.stream(side):
                for _ in range(20):
                    outs.append(run())
            torch.cuda.synchronize()
            bad += sum(0 if torch.equal(o, ref) else 1 for o in outs)
        report.append('%3d/60  %s' % (bad, name))
        if bad:
            failed.append(name)
    print('\nlaunches with different bits beside the F16-MFMA load:\n  ' + '\n  '.join(report))
    assert not failed, failed


ALL_CONFIGS = [('C1', 8), ('C2', 32), ('C3', 8), ('C4', 4), ('C5', 4)]


@pytest.mark.parametrize('cfg_name,B', ALL_CONFIGS)
def test_engine_two_stream_schedule_is_bit_equal_to_serial_everywhere(cfg_name, B):
    """The engine's default schedule (U-Net backbone on a second stream beside the retrieval path) against the one-stream schedule: 20 steps of
    each config must give the serial bits -- also with the F16-MFMA load kernel of the testkit running beside half of them on a third stream."""
    if not torch.cuda.is_available():
        pytest.skip('needs the GPU')
    from rfuse.database import PatchDatabase
    from rfuse.engine import RefinementEngine
    cfg = rf_configs.get_config(cfg_name)
    db = synthetic.make_database(6, cfg, 64 * 30)
    eng = RefinementEngine(cfg, DEV, PatchDatabase(db['emb'], db['meta'], db['volumes'], DEV))
    sds = {}
    for name, m in eng.modules().items():
        sds[name] = helpers.seeded_sd({k: tuple(v.shape) for k, v in m.state_dict().items()}, 91 + len(name))
    eng.load_state_dicts(sds)
    raws = np.stack([synthetic.make_chunk(5000 + b, cfg)['input_raw'] for b in range(B)])
    x = torch.from_numpy(raws).to(DEV)
    noise = None
    if cfg['attn_retrieval_mode']:
        rows = B * cfg['attn_num_patch'] ** 3
        noise = (-torch.empty(rows, cfg['K']).exponential_(generator=torch.Generator().manual_seed(2)).log()).to(DEV)
    eng.serial = True
    ref = eng.refine(x, gumbel_noise=noise).clone()
    eng.serial = False
    third = torch.cuda.Stream(DEV)
    scratch = torch.empty(256 * 256, device=DEV)
    torch.cuda.synchronize()
    bad = 0
    for it in range(20):
        if it % 2:
            testkit.f16_mfma_load(third, scratch)
        out = eng.refine(x, gumbel_noise=noise)
        torch.cuda.synchronize()
        bad += 0 if torch.equal(out, ref) else 1
    assert bad == 0, '%d of 20 two-stream steps differ from the serial schedule' % bad
