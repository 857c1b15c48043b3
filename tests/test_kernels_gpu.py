"""GPU parity tests, kernel by kernel, through the C ABI (rfuse.ops -> librfuse_hip.so) against the oracle /
plain fp32 PyTorch-CPU references of the same op.  Run with ``-m gpu`` on an MI355X."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import refpath

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('no GPU visible')
    from rfuse import ops as _ops
    return _ops


def rnd(gen, *shape, scale=1.0):
    return (torch.randn(*shape, generator=gen) * scale).float()


def close(got, ref, tol=2e-5, what=''):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = (got - ref).abs().max().item()
    bound = tol * max(1.0, ref.abs().max().item())
    assert err <= bound, f'{what}: max abs err {err:.3e} > {bound:.3e}'


def same_affine(a, b, what=''):
    """two GroupNorm affine triples [n, C, 4] describe the same map y = (x - center) * scale + shift (to 1e-6): the scales
    agree and the two-term shifts  shift - center * scale  agree (the centre itself may differ by an fp32 rounding flip)"""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    close(a[..., 1], b[..., 1], 1e-6, what + ' (scale)')
    close(a[..., 2] - a[..., 0] * a[..., 1], b[..., 2] - b[..., 0] * b[..., 1], 1e-6, what + ' (shift)')


def ref_gcr(src0, src1, gamma, beta, groups, w):
    parts = []
    if src0 is not None:
        parts.append(src0)
    if src1 is not None:
        parts.append(F.interpolate(src1, scale_factor=2, mode='nearest'))
    x = torch.cat(parts, dim=1)
    g = 1 if x.shape[1] < groups else groups
    x = F.group_norm(x, g, gamma, beta, eps=1e-5)
    return F.relu(F.conv3d(x, w, None, padding=1))


CONV_CASES = [
    # (n, c0, c1, edge, cout, groups)
    (2, 1, 0, 8, 8, 8),        # first layer: cin=1 (< groups -> one group), cout < 16
    (3, 8, 0, 8, 16, 8),
    (1, 16, 0, 16, 32, 8),     # several 8^3 tiles per sample
    (2, 32, 64, 8, 56, 8),     # decoder read: skip + upsampled, cout=56 (StepDown), groups straddle the two sources
    (1, 0, 32, 16, 32, 8),     # DecoderNoJoining: upsampled source only
    (9, 16, 0, 4, 32, 8),      # 4^3 volumes: 8 samples per workgroup, partial last workgroup
    (5, 64, 128, 4, 64, 8),    # 4^3 decoder read
    (70, 64, 0, 2, 128, 8),    # 2^3 volumes: 64 samples per workgroup, cout block 2
    (3, 64, 128, 2, 64, 8),
    (4, 64, 0, 1, 128, 8),     # 1^3 -> direct path
    (1, 6, 0, 8, 12, 6),       # nf=12 family: cin not a multiple of 4
    (1, 78, 0, 8, 12, 6),
    (1, 16, 0, 32, 16, 8),
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv3d_gn_relu(ops, case):
    n, c0, c1, edge, cout, groups = case
    gen = torch.Generator().manual_seed(hash(case) & 0xffff)
    src0 = rnd(gen, n, c0, edge, edge, edge).relu_() if c0 else None          # inputs are post-ReLU in the network
    src1 = rnd(gen, n, c1, edge // 2, edge // 2, edge // 2).relu_() if c1 else None
    cin = c0 + c1
    gamma, beta = 1 + 0.2 * rnd(gen, cin), 0.2 * rnd(gen, cin)
    w = rnd(gen, cout, cin, 3, 3, 3, scale=1.0 / np.sqrt(27 * cin))
    ref = ref_gcr(src0, src1, gamma, beta, groups, w)
    d0 = src0.to(DEV) if src0 is not None else None
    d1 = src1.to(DEV) if src1 is not None else None
    aff = ops.gn_affine(d0, d1, gamma.to(DEV), beta.to(DEV), groups)
    # GroupNorm folding itself: (center, scale, shift) with y = (x - center) * scale + shift
    g = 1 if cin < groups else groups
    xcat = torch.cat([t for t in (src0, F.interpolate(src1, scale_factor=2, mode='nearest') if src1 is not None else None) if t is not None], 1)
    xg = xcat.double().reshape(n, g, -1)
    mean, var = xg.mean(-1), xg.var(-1, unbiased=False)
    rstd = 1 / torch.sqrt(var + 1e-5)
    cpg = cin // g
    sc_ref = gamma.double()[None] * rstd.repeat_interleave(cpg, 1)
    sh_ref = beta.double()[None] - mean.repeat_interleave(cpg, 1) * sc_ref
    mean_c = mean.repeat_interleave(cpg, 1)
    a64 = aff.cpu().double()
    close(aff[..., 1], sc_ref, 1e-6, 'gn scale')
    assert torch.equal(aff[..., 0].cpu(), mean_c.float()), 'centre = fp32-rounded float64 mean'
    close(a64[..., 2] - a64[..., 0] * a64[..., 1], sh_ref, 1e-6, 'gn shift (folded back to the two-term form)')
    # what fl32(mean) loses must be carried by `shift`: the composite equals GroupNorm to ~eps*|y|, not eps*|mean*scale|
    close(a64[..., 2], beta.double()[None] - (mean_c - mean_c.float().double()) * sc_ref, 1e-7, 'gn shift')
    wd = w.to(DEV)
    direct = ops.conv3d_gn_relu(d0, d1, aff, None, cout, direct_weight=wd)
    close(direct, ref, 2e-5, 'direct conv vs torch')
    if edge >= 2:
        mfma = ops.conv3d_gn_relu(d0, d1, aff, ops.pack_conv3_weight(wd), cout)
        close(mfma, ref, 2e-5, 'mfma conv vs torch')
        close(mfma, direct, 2e-5, 'mfma conv vs direct conv')


UP_CASES = [
    # (n, c0, c1, edge, cout, groups): shapes rf_conv3d_up_supported accepts (>= 256 boxes)
    (256, 32, 64, 8, 56, 8),    # retrieval backbone dec1 (dominant layer), cout 56 -> NB 4 with a half-empty block
    (300, 0, 16, 8, 16, 8),     # DecoderNoJoining: no skip source
    (1027, 64, 128, 4, 64, 8),  # 4^3 decoder read, four samples per workgroup, ragged last workgroup
    (2, 0, 16, 64, 16, 8),      # final decoder: 8^3 boxes of a 64^3 volume (halo crosses box borders)
    (40, 6, 12, 16, 12, 6),     # nf = 12 family: c0 % 4 != 0, c1 % 8 != 0
    (300, 8, 8, 8, 72, 8),      # cout16 = 80: two cout blocks
    (260, 4, 8, 8, 24, 4),      # NB = 2
    (1100, 16, 8, 4, 24, 8),
    # many 8^3 samples (bench-size launches of the parity-split box kernel, ragged sample counts)
    (600, 32, 64, 8, 56, 8),
    (530, 0, 16, 8, 16, 8),
    (520, 6, 12, 8, 12, 6),
    (513, 8, 8, 8, 72, 8),
    # couts 48..55 on the 4x4x1 MFMA form (3 n-blocks + 8): partial group, several boxes per sample (x halos), padded channels
    (40, 8, 16, 16, 52, 4),
    (300, 6, 12, 8, 49, 6),
    # many 4^3 samples: the position-major decoder form (conv3d_small.hip, UP instances)
    (4100, 64, 128, 4, 64, 8),
    (4099, 16, 8, 4, 24, 8),
    (8200, 0, 16, 4, 16, 8),
    (4104, 6, 12, 4, 12, 6),
]


@pytest.mark.parametrize('case', UP_CASES)
def test_conv3d_up_parity_split_decoder_kernel(ops, case):
    """rf_conv3d_up_k3_gn_relu (upsampled channels convolved in low resolution, pre-summed taps) vs float64 torch on the
    materialised upsample + concat (model/unet.py:297-308, 19-76) and vs the generic kernel; fused statistics too."""
    n, c0, c1, edge, cout, groups = case
    gen = torch.Generator().manual_seed(sum(case))
    src0 = rnd(gen, n, c0, edge, edge, edge).relu_() if c0 else None
    src1 = rnd(gen, n, c1, edge // 2, edge // 2, edge // 2).relu_()
    cin = c0 + c1
    gamma, beta = 1 + 0.2 * rnd(gen, cin), 0.2 * rnd(gen, cin)
    w = rnd(gen, cout, cin, 3, 3, 3, scale=1.0 / np.sqrt(27 * cin))
    d0 = src0.to(DEV) if src0 is not None else None
    d1 = src1.to(DEV)
    assert ops.conv_up_supported(d0, d1, cout)
    aff = ops.gn_affine(d0, d1, gamma.to(DEV), beta.to(DEV), groups)
    wd = w.to(DEV)
    got = ops.conv3d_up_gn_relu(d0, d1, aff, ops.pack_conv3_up_weight(wd, c0), cout)
    generic = ops.conv3d_gn_relu(d0, d1, aff, ops.pack_conv3_weight(wd), cout)
    close(got, generic, 1e-5, 'parity-split vs generic kernel')
    sub = slice(0, min(n, 24))                                 # float64 reference on a slice (CPU time)
    sub_tail = slice(max(0, n - 5), n)
    for sl in (sub, sub_tail):
        ref = ref_gcr(src0[sl].double() if c0 else None, src1[sl].double(), gamma.double(), beta.double(), groups, w.double())
        close(got[sl], ref.float(), 1e-5, 'parity-split vs float64 torch')
    g2, b2 = (1 + 0.2 * rnd(gen, cout)).to(DEV), (0.2 * rnd(gen, cout)).to(DEV)
    g = groups if cout % groups == 0 else 1
    assert getattr(got, '_rf_stats', None) is not None
    fused = ops.gn_affine(got, None, g2, b2, g)
    plain = ops.gn_affine(got.clone(), None, g2, b2, g)
    same_affine(fused, plain, 'scale from fused stats')


SPLIT_BOX_CASES = [
    # (n, cin, edge, cout, groups): shapes rf_conv3d_split_supported accepts (cin in eights, <= 32 couts, >= 1024 boxes of 8^3)
    (1030, 8, 8, 16, 8),       # one chunk, whole 8^3 volumes (zero-padding halo), ragged sample count
    (130, 8, 16, 16, 8),       # retrieval backbone enc0 second conv: 8 boxes per sample, halos from the neighbouring boxes
    (1100, 16, 8, 32, 8),      # NB 2, two chunks
    (1025, 56, 8, 16, 8),      # seven chunks (odd), GroupNorm groups of 7 channels
    (3, 16, 64, 16, 8),        # final decoder: 512 boxes per sample
    (1040, 24, 8, 12, 6),      # nf = 12 family: cout < 16
    (2100, 24, 8, 12, 6),      # ... with enough boxes for the persistent z-column form (k_conv3_split_zcm): three chunks, 12 couts
    (2049, 16, 8, 16, 8),      # ... two chunks, ragged box count over the workgroups
    (5, 16, 64, 16, 8),        # ... 2560 boxes of five 64^3 samples: halos from the neighbouring boxes, statistics per box
    (40, 56, 32, 16, 8),       # ... seven chunks, 2560 boxes
    (2100, 16, 8, 32, 8),      # ... 32 couts: two cout blocks of the persistent form
    (2050, 12, 8, 12, 6),      # ... cin not a multiple of 8 (nf = 12): the last chunk's missing channels as zero slots
    (2060, 42, 8, 24, 6),      # ... 42 -> 48 slots, six chunks, two cout blocks
    (5, 12, 64, 12, 6),        # ... C5's final decoder at a batch that gives 2560 boxes
    (6, 32, 64, 24, 8),        # ... 24 couts (the second block half empty), four chunks, halos
    (33, 32, 32, 24, 8),
    (1040, 12, 8, 12, 6),      # cin not a multiple of 8: the last chunk's missing channels are zero slots (nf = 12: C5's U-Net)
    (1030, 42, 8, 12, 6),      # 42 -> 48 slots, six chunks
    (3, 12, 64, 12, 6),        # C5's final decoder
    (140, 20, 16, 24, 4),
    (3, 6, 128, 12, 3),        # C5's U-Net, second conv of level 0: six of eight slots real, one chunk
    (1030, 6, 16, 12, 6),      # ... of its retrieval backbone
    (20, 12, 64, 24, 6),       # 12 -> 24: zero slots and two n-blocks in one workgroup
    (1030, 32, 4, 64, 8),      # whole 4^3 samples, 8 per workgroup (k_conv3_split_s4): ragged sample count, four cout blocks
    (1024, 64, 4, 64, 8),
    (2050, 8, 4, 12, 4),       # one chunk, cout < 16
    (1100, 32, 4, 32, 8),
    (1030, 24, 4, 48, 6),      # nf = 12 (C5): 48 couts in one workgroup (NB 3)
    (1026, 48, 4, 96, 6),      # ... 96 couts: two workgroups of 48
    (1025, 16, 4, 44, 4),      # padded couts in the NB 3 instance
    (16, 24, 32, 48, 6),       # more than 32 couts (round 6; the deep levels of C5's U-Net at 16 chunks): three 16-cout blocks on grid.y, fused max-pool + statistics
    (16, 48, 16, 96, 6),       # ... 128 boxes x three 32-cout blocks
    (16, 96, 16, 96, 6),       # ... twelve chunks
    (32, 32, 16, 64, 8),       # ... the nf = 16 U-Nets' 32 -> 64 @16^3 at B = 32
    (9, 16, 32, 40, 4),        # ... a padded last block (40 -> 48)
]


@pytest.mark.parametrize('case', SPLIT_BOX_CASES)
def test_conv3d_split_operand_box_kernel(ops, case):
    """rf_conv3d_split_k3_gn_relu (conv3d_split.hip) vs float64 torch (model/unet.py:19-76) and vs the fp32-MFMA box kernel: same bar, no
    further from float64 than it; the fused MaxPool3d(2) outputs equal the stand-alone pool of its own output bit for bit; statistics."""
    n, cin, edge, cout, groups = case
    gen = torch.Generator().manual_seed(sum(case) + 11)
    src = rnd(gen, n, cin, edge, edge, edge).relu_()
    gamma, beta = 1 + 0.2 * rnd(gen, cin), 0.2 * rnd(gen, cin)
    w = rnd(gen, cout, cin, 3, 3, 3, scale=1.0 / np.sqrt(27 * cin))
    x = src.to(DEV)
    assert ops.conv_split_supported(x, None, cout)
    aff = ops.gn_affine(x, None, gamma.to(DEV), beta.to(DEV), groups)
    wd = w.to(DEV)
    ws = ops.pack_conv3_split_weight(wd)
    got = ops.conv3d_split_gn_relu(x, aff, ws, cout)
    fp32 = ops.conv3d_gn_relu(x, None, aff, ops.pack_conv3_weight(wd), cout)
    close(got, fp32, 1e-5, 'split-operand vs fp32-MFMA box kernel')
    nref = max(1, min(n, 16384 // edge ** 3 * 2))
    e_split, e_fp32 = [], []
    for sl in (slice(0, nref), slice(max(0, n - 3), n)):
        ref = ref_gcr(src[sl].double(), None, gamma.double(), beta.double(), groups, w.double())
        close(got[sl], ref.float(), 1e-5, 'split-operand vs float64 torch')
        e_split.append((got[sl].cpu().double() - ref).flatten())
        e_fp32.append((fp32[sl].cpu().double() - ref).flatten())
    e_split, e_fp32 = torch.cat(e_split), torch.cat(e_fp32)
    rms_s, rms_f = e_split.pow(2).mean().sqrt().item(), e_fp32.pow(2).mean().sqrt().item()
    print(f'\n{case}: error vs float64  split rms {rms_s:.3e} max {e_split.abs().max().item():.3e} | fp32 MFMA rms {rms_f:.3e} max {e_fp32.abs().max().item():.3e}')
    assert rms_s <= 1.05 * rms_f and e_split.abs().max().item() <= 1.25 * e_fp32.abs().max().item()
    want_pool = ops.maxpool2(got)
    full, pooled = ops.conv3d_split_gn_relu(x, aff, ws, cout, pool='also')
    assert torch.equal(full, got) and torch.equal(pooled, want_pool)
    none, pooled_only = ops.conv3d_split_gn_relu(x, aff, ws, cout, pool='only')
    assert none is None and torch.equal(pooled_only, want_pool)
    g = groups if cout % groups == 0 else 1
    g2, b2 = (1 + 0.2 * rnd(gen, cout)).to(DEV), (0.2 * rnd(gen, cout)).to(DEV)
    for t in (got, pooled, pooled_only, full):
        assert getattr(t, '_rf_stats', None) is not None
        same_affine(ops.gn_affine(t, None, g2, b2, g), ops.gn_affine(t.clone(), None, g2, b2, g), 'scale from fused stats')


E2_CASES = [
    # (n, cin, cout, groups, edge): whole 2^3 / 1^3 volumes as one dense GEMM (rf_conv3d_e2_split_k3_gn_relu)
    (8192, 64, 64, 8, 2),      # retrieval backbone enc3, first conv (C1-C4, B = 32)
    (1030, 64, 128, 8, 2),     # ... second conv: four n-chunks, ragged sample count
    (300, 48, 96, 6, 2),       # nf = 12 (C5): 768 columns = three n-chunks
    (260, 8, 6, 4, 2),         # two k-steps, one partial n-chunk (48 of 256 columns)
    (513, 20, 40, 4, 2),       # cin a multiple of 4 only, partial second chunk
    (32, 32, 64, 8, 2),        # the U-Net backbone's 2^3 level at B = 32: half a workgroup of samples
    (32, 64, 128, 8, 1),       # ... its 1^3 level: K = cin, N = cout, centre tap
    (300, 72, 40, 8, 1),       # cin not a multiple of 32: zero-filled k-step tail
    (16, 128, 300, 8, 1),      # two n-chunks
]


@pytest.mark.parametrize('case', E2_CASES)
def test_conv3d_e2_split_gemm_form(ops, case):
    """rf_conv3d_e2_split_k3_gn_relu (csrc/conv3d_e2_split.hip: SingleConv 'gcr' of model/unet.py:19-76 on whole 2^3 / 1^3 volumes as one dense GEMM on
    the F16 matrix cores) vs float64 torch and vs the fp32 kernels (position-major / direct): the same bar, and no further from float64 than they; fused
    statistics."""
    n, cin, cout, groups, edge = case
    gen = torch.Generator().manual_seed(sum(case) + 5)
    src = rnd(gen, n, cin, edge, edge, edge).relu_()
    gamma, beta = 1 + 0.2 * rnd(gen, cin), 0.2 * rnd(gen, cin)
    w = rnd(gen, cout, cin, 3, 3, 3, scale=1.0 / np.sqrt(edge ** 3 * cin))
    x = src.to(DEV)
    assert ops.conv_e2_split_supported(x, cout)
    aff = ops.gn_affine(x, None, gamma.to(DEV), beta.to(DEV), groups)
    wd = w.to(DEV)
    got = ops.conv3d_e2_split_gn_relu(x, aff, ops.pack_conv3_e2_split_weight(wd, edge), cout)
    saved, ops.CONV_ARITH = ops.CONV_ARITH, 'fp32'
    try:
        fp32 = ops.conv3d_gn_relu(x, None, aff, ops.pack_conv3_weight(wd) if edge > 1 else None, cout, direct_weight=wd if edge == 1 else None)
    finally:
        ops.CONV_ARITH = saved
    close(got, fp32, 1e-5, 'GEMM form vs the fp32 kernel')
    ref = ref_gcr(src.double(), None, gamma.double(), beta.double(), groups, w.double())
    close(got, ref.float(), 1e-5, 'GEMM form vs float64 torch')
    e_split, e_fp32 = (got.cpu().double() - ref).flatten(), (fp32.cpu().double() - ref).flatten()
    rms_s, rms_f = e_split.pow(2).mean().sqrt().item(), e_fp32.pow(2).mean().sqrt().item()
    print(f'\n{case}: error vs float64  split rms {rms_s:.3e} max {e_split.abs().max().item():.3e} | fp32 rms {rms_f:.3e} max {e_fp32.abs().max().item():.3e}')
    assert rms_s <= 1.05 * rms_f and e_split.abs().max().item() <= 1.25 * e_fp32.abs().max().item()
    g = groups if cout % groups == 0 else 1
    g2, b2 = (1 + 0.2 * rnd(gen, cout)).to(DEV), (0.2 * rnd(gen, cout)).to(DEV)
    assert getattr(got, '_rf_stats', None) is not None
    same_affine(ops.gn_affine(got, None, g2, b2, g), ops.gn_affine(got.clone(), None, g2, b2, g), 'scale from fused stats')


@pytest.mark.parametrize('cin,cout', [(16, 16), (16, 32), (8, 16)])
def test_split_box_kernel_leaves_concurrent_kernels_alone(ops, cin, cout):
    """Two-stream regression (the engine runs the U-Net backbone on a side stream): small fp32 convs on a second stream must return
    their solo results bit for bit while split-operand box convs run on the main stream -- including the two-n-block instance (32 couts,
    158 VGPRs) that leaves room for foreign waves on its SIMDs.  Round 2 saw the fp32 kernels' bits move there; the cause was an unsafe
    packed-fp32 instruction form in the fp32 kernels (tests/test_two_stream_gpu.py), not this kernel."""
    gen = torch.Generator().manual_seed(3)
    x = rnd(gen, 2048, cin, 8, 8, 8).relu_().to(DEV)
    aff = ops.gn_affine(x, None, torch.ones(cin, device=DEV), torch.zeros(cin, device=DEV), 8)
    ws = ops.pack_conv3_split_weight(rnd(gen, cout, cin, 3, 3, 3, scale=0.05).to(DEV))
    xs = rnd(gen, 8, 32, 16, 16, 16).relu_().to(DEV)
    affs = ops.gn_affine(xs, None, torch.ones(32, device=DEV), torch.zeros(32, device=DEV), 8)
    wps = ops.pack_conv3_weight(rnd(gen, 32, 32, 3, 3, 3, scale=0.05).to(DEV))
    solo = ops.conv3d_gn_relu(xs, None, affs, wps, 32).clone()
    main_ref = ops.conv3d_split_gn_relu(x, aff, ws, cout).clone()
    side = torch.cuda.Stream(DEV)
    torch.cuda.synchronize()
    wrong = 0
    for _ in range(6):
        outs = []
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(30):
                outs.append(ops.conv3d_gn_relu(xs, None, affs, wps, 32))
        for _ in range(6):
            o = ops.conv3d_split_gn_relu(x, aff, ws, cout)
        torch.cuda.synchronize()
        wrong += sum(0 if torch.equal(v, solo) else 1 for v in outs) + (0 if torch.equal(o, main_ref) else 1)
    assert wrong == 0, f'{wrong} launches returned different bits when run concurrently'


SPLIT_UP_CASES = [
    # (n, c0, c1, edge, cout, groups): shapes rf_conv3d_up_split_supported accepts (whole 8^3 samples, channels in eights, 33..64 couts)
    (256, 32, 64, 8, 56, 8),    # retrieval backbone dec1 of C1-C4 (the dominant launch): 4 + 8 chunks, NB 4 with 8 padded couts
    (300, 24, 48, 8, 42, 6),    # the same layer at nf = 12 (C5): odd number of skip chunks, NB 3
    (257, 0, 16, 8, 64, 8),     # no skip source
    (260, 8, 8, 8, 33, 4),      # one chunk each
    (515, 40, 64, 8, 64, 8),    # five skip chunks, full couts
    (1030, 64, 128, 4, 64, 8),  # whole 4^3 samples (k_conv3_up_split_s4): retrieval backbone dec0, ragged sample count, four groups of low-res chunks
    (1024, 0, 16, 4, 32, 8),    # no skip source, one partial group
    (2050, 8, 8, 4, 60, 4),     # one chunk each, padded couts
    (1100, 32, 72, 4, 32, 8),   # nine low-res chunks: groups of 4 + 4 + 1
    (1030, 48, 96, 4, 48, 6),   # C5's dec0: 48 couts in one workgroup (NB 3 wide instance)
    (4, 0, 16, 64, 16, 8),      # 8^3 boxes of a large volume, no skip source (k_conv3_up_split_box): the final decoder's first conv, 512 boxes per sample
    (33, 0, 32, 32, 16, 8),     # the U-Net backbone's 32 -> 16 @32^3: four channel groups, ragged sample count
    (130, 0, 8, 16, 32, 4),     # one channel group, two n-blocks
    (3, 0, 24, 64, 20, 4),      # three channel groups, padded couts
    (16, 48, 96, 32, 78, 6),    # boxes WITH a skip source (k_conv3_up_split_boxskip): C5's U-Net decoder 48 + 96 -> 78 @32^3: six + twelve chunks, 5 n-blocks as 3 + 2 (one re-read)
    (70, 8, 8, 16, 20, 4),      # one chunk each, two n-blocks in one group, ragged sample count
    (2, 24, 48, 64, 24, 6),     # 512 boxes per sample, halos from 26 neighbours
    (9, 16, 40, 32, 16, 8),     # one n-block
]


@pytest.mark.parametrize('case', SPLIT_UP_CASES)
def test_conv3d_up_split_operand_kernel(ops, case):
    """rf_conv3d_up_split_k3_gn_relu (fp32 operands as two f16 pieces on the F16 matrix cores, hi/lo fp32 accumulators) vs float64
    torch on the materialised upsample + concat (model/unet.py:297-308, 19-76) and vs the fp32-MFMA decoder kernel: the same bar as
    that kernel, AND no further from the float64 result than it (the split form is the more accurate one); fused statistics too."""
    n, c0, c1, edge, cout, groups = case
    gen = torch.Generator().manual_seed(sum(case) + 7)
    src0 = rnd(gen, n, c0, edge, edge, edge).relu_() if c0 else None
    src1 = rnd(gen, n, c1, edge // 2, edge // 2, edge // 2).relu_()
    cin = c0 + c1
    gamma, beta = 1 + 0.2 * rnd(gen, cin), 0.2 * rnd(gen, cin)
    w = rnd(gen, cout, cin, 3, 3, 3, scale=1.0 / np.sqrt(27 * cin))
    d0 = src0.to(DEV) if src0 is not None else None
    d1 = src1.to(DEV)
    assert ops.conv_up_split_supported(d0, d1, cout)
    aff = ops.gn_affine(d0, d1, gamma.to(DEV), beta.to(DEV), groups)
    wd = w.to(DEV)
    got = ops.conv3d_up_split_gn_relu(d0, d1, aff, ops.pack_conv3_up_split_weight(wd, c0), cout)
    if ops.conv_up_supported(d0, d1, cout):
        fp32 = ops.conv3d_up_gn_relu(d0, d1, aff, ops.pack_conv3_up_weight(wd, c0), cout)
    else:                                                            # no fp32 decoder-form instance for this shape: the generic fp32 kernel on the virtual upsample
        saved, ops.CONV_ARITH = ops.CONV_ARITH, 'fp32'
        try:
            fp32 = ops.conv3d_gn_relu(d0, d1, aff, ops.pack_conv3_weight(wd), cout)
        finally:
            ops.CONV_ARITH = saved
    close(got, fp32, 1e-5, 'split-operand vs fp32-MFMA decoder kernel')
    e_split, e_fp32 = [], []
    for sl in (slice(0, min(n, 24 if edge <= 8 else 2)), slice(max(0, n - (5 if edge <= 8 else 1)), n)):       # float64 reference on slices (CPU time)
        ref = ref_gcr(src0[sl].double() if c0 else None, src1[sl].double(), gamma.double(), beta.double(), groups, w.double())
        close(got[sl], ref.float(), 1e-5, 'split-operand vs float64 torch')
        e_split.append((got[sl].cpu().double() - ref).flatten())
        e_fp32.append((fp32[sl].cpu().double() - ref).flatten())
    e_split, e_fp32 = torch.cat(e_split), torch.cat(e_fp32)
    rms_s, rms_f = e_split.pow(2).mean().sqrt().item(), e_fp32.pow(2).mean().sqrt().item()
    print(f'\n{case}: error vs float64  split rms {rms_s:.3e} max {e_split.abs().max().item():.3e} | fp32 MFMA rms {rms_f:.3e} max {e_fp32.abs().max().item():.3e}')
    assert rms_s <= 1.05 * rms_f and e_split.abs().max().item() <= 1.25 * e_fp32.abs().max().item()
    g2, b2 = (1 + 0.2 * rnd(gen, cout)).to(DEV), (0.2 * rnd(gen, cout)).to(DEV)
    g = groups if cout % groups == 0 else 1
    assert getattr(got, '_rf_stats', None) is not None
    fused = ops.gn_affine(got, None, g2, b2, g)
    plain = ops.gn_affine(got.clone(), None, g2, b2, g)
    same_affine(fused, plain, 'scale from fused stats')


UP_CH8_CASES = [
    # (n, c1, edge, cout): rf_conv3d_up_split_k3_gn_relu_ch8 -- from 2048 boxes on the persistent kernel (k_conv3_up_split_boxp), below that one workgroup per box
    (5, 16, 64, 16),      # the final decoder's first conv: 2560 boxes, five per workgroup
    (3, 16, 64, 16),      # 1536 boxes: one workgroup per box
    (300, 8, 16, 8),      # one channel group in, one out: 2400 boxes = 480 workgroups x 5, a run crosses samples every 8 boxes
    (37, 16, 32, 8),      # 2368 boxes: the last workgroup's run is short
    (41, 8, 32, 16),      # one group in, two out
]


@pytest.mark.parametrize('case', UP_CH8_CASES)
def test_conv3d_up_split_channel_interleaved_output(ops, case):
    """The channel-interleaved ([n][cout / 8][voxel][8]) form of the decoder box conv (model/unet.py:297-308 without a skip source) holds exactly the
    values and the statistics of the NCDHW form: the persistent kernel changes the schedule of a box, not its arithmetic."""
    n, c1, edge, cout = case
    gen = torch.Generator().manual_seed(sum(case) + 11)
    src1 = rnd(gen, n, c1, edge // 2, edge // 2, edge // 2).relu_().to(DEV)
    gamma, beta = (1 + 0.2 * rnd(gen, c1)).to(DEV), (0.2 * rnd(gen, c1)).to(DEV)
    w = rnd(gen, cout, c1, 3, 3, 3, scale=1.0 / np.sqrt(27 * c1)).to(DEV)
    aff = ops.gn_affine(None, src1, gamma, beta, 8 if c1 % 8 == 0 else 1)
    wp = ops.pack_conv3_up_split_weight(w, 0)
    from rfuse import _lib
    assert _lib.load().rf_conv3d_up_split_ch8_supported(0, c1, n, edge, cout)
    plain = ops.conv3d_up_split_gn_relu(None, src1, aff, wp, cout)
    got, stats, tiles = ops.conv3d_up_split_gn_relu_ch8(src1, aff, wp, cout)
    back = got.permute(0, 1, 5, 2, 3, 4).reshape(n, cout, edge, edge, edge)
    assert torch.equal(back, plain), 'ch8 output differs from the NCDHW output: max %.3e' % (back - plain).abs().max().item()
    pst, ptiles = plain._rf_stats[:2]
    assert tiles == ptiles and torch.equal(stats, pst), 'per-box statistics differ'
    ref = ref_gcr(None, src1[:1].cpu().double(), gamma.cpu().double(), beta.cpu().double(), 8 if c1 % 8 == 0 else 1, w.cpu().double())
    close(back[:1], ref.float(), 1e-5, 'ch8 vs float64 torch')


def test_conv3d_up_split_saturates_instead_of_overflowing(ops):
    """activations beyond the f16 range of the split (|GroupNorm output| > 65504 * 16) saturate; nothing becomes inf / nan"""
    n, c0, c1, cout = 256, 8, 8, 40
    gen = torch.Generator().manual_seed(5)
    d0 = rnd(gen, n, c0, 8, 8, 8).to(DEV)
    d1 = rnd(gen, n, c1, 4, 4, 4).to(DEV)
    aff = torch.zeros(n, c0 + c1, 4, device=DEV)
    aff[..., 1] = 1.0
    aff[0, :, 1] = 1e7                                                # sample 0: far out of range
    w = rnd(gen, cout, c0 + c1, 3, 3, 3, scale=0.05).to(DEV)
    got = ops.conv3d_up_split_gn_relu(d0, d1, aff, ops.pack_conv3_up_split_weight(w, c0), cout)
    assert torch.isfinite(got).all()
    fp32 = ops.conv3d_up_gn_relu(d0, d1, aff, ops.pack_conv3_up_weight(w, c0), cout)
    close(got[1:], fp32[1:], 1e-5, 'in-range samples unaffected')


@pytest.mark.parametrize('case', [(300, 8, 16, 16, 8), (1100, 16, 8, 32, 8), (2, 16, 64, 16, 8), (130, 6, 16, 12, 6), (520, 16, 8, 72, 8),
                                  (4100, 32, 4, 64, 8), (2100, 16, 4, 16, 8), (2050, 6, 4, 12, 6)])
def test_conv_with_fused_maxpool_epilogue(ops, case):
    """rf_conv3d_k3_gn_relu_pool == conv followed by the stand-alone MaxPool3d(2) kernel, bit for bit, with and without the
    full-resolution output; the pooled tensor's fused statistics give the same GroupNorm fold as re-reading it."""
    n, cin, edge, cout, groups = case
    gen = torch.Generator().manual_seed(sum(case))
    x = rnd(gen, n, cin, edge, edge, edge).relu_().to(DEV)
    gamma, beta = (1 + 0.2 * rnd(gen, cin)).to(DEV), (0.2 * rnd(gen, cin)).to(DEV)
    w = rnd(gen, cout, cin, 3, 3, 3, scale=1.0 / np.sqrt(27 * cin)).to(DEV)
    aff = ops.gn_affine(x, None, gamma, beta, groups)
    wp = ops.pack_conv3_weight(w)
    assert ops.conv_pool_supported(x, None, cout)
    plain = ops.conv3d_gn_relu(x, None, aff, wp, cout)
    want_pool = ops.maxpool2(plain)
    full, pooled = ops.conv3d_gn_relu_pool(x, None, aff, wp, cout, keep_full=True)
    assert torch.equal(full, plain) and torch.equal(pooled, want_pool)
    none, pooled_only = ops.conv3d_gn_relu_pool(x, None, aff, wp, cout, keep_full=False)
    assert none is None and torch.equal(pooled_only, want_pool)
    g = groups if cout % groups == 0 else 1
    g2, b2 = (1 + 0.2 * rnd(gen, cout)).to(DEV), (0.2 * rnd(gen, cout)).to(DEV)
    for t in (pooled, pooled_only, full):
        assert getattr(t, '_rf_stats', None) is not None
        fused = ops.gn_affine(t, None, g2, b2, g)
        reread = ops.gn_affine(t.clone(), None, g2, b2, g)
        same_affine(fused, reread, 'scale from fused stats')


@pytest.mark.parametrize('case', [(4100, 32, 4, 64, 8), (4099, 16, 4, 16, 8), (4104, 6, 4, 12, 6), (33000, 64, 2, 128, 8), (32770, 64, 2, 16, 8),
                                  (8192, 64, 4, 72, 8), (600, 16, 8, 32, 8), (530, 56, 8, 16, 8), (515, 6, 8, 12, 6), (1030, 16, 8, 72, 8)])
def test_conv_small_volume_position_major_kernel(ops, case):
    """Whole 4^3 / 2^3 volumes, many samples: the position-major kernel (conv3d_small.hip: every zero-padding tap left out)
    must equal the generic kernel bit for bit (the generic one is reached by calling with few samples at a time), statistics
    included; a slice is also checked against float64 torch."""
    n, cin, edge, cout, groups = case
    gen = torch.Generator().manual_seed(sum(case))
    x = rnd(gen, n, cin, edge, edge, edge).relu_().to(DEV)
    gamma, beta = (1 + 0.2 * rnd(gen, cin)).to(DEV), (0.2 * rnd(gen, cin)).to(DEV)
    w = rnd(gen, cout, cin, 3, 3, 3, scale=1.0 / np.sqrt(27 * cin))
    wp = ops.pack_conv3_weight(w.to(DEV))
    aff = ops.gn_affine(x, None, gamma, beta, groups)
    got = ops.conv3d_gn_relu(x, None, aff, wp, cout)
    step = {8: 500, 4: 500, 2: 3000}[edge]                      # few enough samples (< 128 workgroups) for the box-tiled kernel
    parts, stat_parts = [], []
    for i in range(0, n, step):
        y = ops.conv3d_gn_relu(x[i:i + step].contiguous(), None, aff[i:i + step].contiguous(), wp, cout)
        parts.append(y)
        stat_parts.append(y._rf_stats[0])
    want = torch.cat(parts)
    assert torch.equal(got, want)
    close(got._rf_stats[0].sum(dim=2), torch.cat(stat_parts).sum(dim=2), 1e-12, 'fused statistics')
    sl = slice(n - 20, n)
    ref = ref_gcr(x[sl].cpu().double(), None, gamma.cpu().double(), beta.cpu().double(), groups, w.double())
    close(got[sl], ref.float(), 2e-5, 'vs float64 torch')


def test_conv_identity_weight_is_transpose_detecting(ops):
    """centre-tap identity on an asymmetric ramp: catches swapped voxel axes / channel transposes exactly."""
    n, c, edge = 1, 16, 8
    x = torch.arange(n * c * edge ** 3, dtype=torch.float32).reshape(n, c, edge, edge, edge) / 100.0
    w = torch.zeros(c, c, 3, 3, 3)
    for i in range(c):
        w[(i * 5) % c, i, 1, 1, 1] = 1.0                  # a channel permutation, not the identity
    aff = torch.zeros(n, c, 4, device=DEV)
    aff[..., 1] = 1.0                                     # centre 0, scale 1, shift 0: identity
    out = ops.conv3d_gn_relu(x.to(DEV), None, aff, ops.pack_conv3_weight(w.to(DEV)), c)
    ref = F.conv3d(x, w, padding=1)
    assert torch.equal(out.cpu(), ref)


def test_maxpool_and_conv1x1_tanh(ops):
    gen = torch.Generator().manual_seed(3)
    x = rnd(gen, 3, 5, 8, 8, 8)
    assert torch.equal(ops.maxpool2(x.to(DEV)).cpu(), F.max_pool3d(x, 2))
    for n, c, e in ((67, 5, 2), (130, 24, 4), (33, 7, 8), (3, 2, 16)):             # the small-volume kernel (edge <= 8: several planes per wave), ragged plane counts
        x = rnd(gen, n, c, e, e, e)
        got = ops.maxpool2(x.to(DEV))
        ref = F.max_pool3d(x, 2)
        assert torch.equal(got.cpu(), ref)
        st = got._rf_stats[0].sum(dim=2).cpu()                                     # [n, c, 2] float64 (sum, sum of squares) of the pooled planes
        want = torch.stack([ref.double().sum(dim=(2, 3, 4)), (ref.double() ** 2).sum(dim=(2, 3, 4))], dim=-1)
        assert torch.allclose(st, want, rtol=1e-12, atol=1e-12)
    x = rnd(gen, 2, 16, 16, 16, 16)
    w, b = rnd(gen, 1, 16, 1, 1, 1, scale=0.3), rnd(gen, 1)
    ref = torch.tanh(F.conv3d(x, w, b))
    close(ops.conv1x1_tanh(x.to(DEV), w.to(DEV), b.to(DEV)), ref, 1e-6, 'conv1x1+tanh')
    close(ops.conv1x1_tanh(x.to(DEV), w.to(DEV), b.to(DEV), 1.0, 0.0625 / 2), (ref + 1) * 0.0625 / 2, 1e-6, 'df epilogue')


@pytest.mark.parametrize('shape', [(2, 3, 8, 2), (1, 16, 32, 2), (4, 1, 64, 16), (2, 16, 32, 8)])
def test_unfold_fold(ops, shape):
    b, c, s, e = shape
    x = torch.arange(b * c * s ** 3, dtype=torch.float32).reshape(b, c, s, s, s)
    rows = ops.unfold3d(x.to(DEV), e)
    assert torch.equal(rows.cpu(), refpath.unfold3d(x, e))
    assert torch.equal(ops.fold3d(rows, s // e, e, c).cpu(), x)


@pytest.mark.parametrize('case', [(4096, 128, 128, 2), (1000, 128, 32, 0), (64, 64, 128, 1), (130, 512, 256, 1), (64, 256, 64, 0),
                                  (37, 96, 128, 2), (5, 7, 3, 0)])
def test_linear(ops, case):
    rows, nin, nout, act = case
    gen = torch.Generator().manual_seed(rows + nin)
    x, w, b = rnd(gen, rows, nin), rnd(gen, nout, nin, scale=1 / np.sqrt(nin)), rnd(gen, nout)
    ref = F.linear(x, w, b)
    ref = F.relu(ref) if act == 1 else (F.leaky_relu(ref, 0.01) if act == 2 else ref)
    got = ops.linear(x.to(DEV), ops.pack_linear_weight(w.to(DEV)), b.to(DEV), nout, act, 0.01)
    close(got, ref, 1e-5, 'linear')
    eye = torch.eye(nin)[:min(nin, nout)] if nout <= nin else None
    if eye is not None:      # asymmetric, transpose-detecting: y = first rows of x
        got = ops.linear(x.to(DEV), ops.pack_linear_weight(eye.contiguous().to(DEV)), None, eye.shape[0])
        assert torch.equal(got.cpu(), x[:, :eye.shape[0]])


def test_l2_normalize(ops):
    gen = torch.Generator().manual_seed(5)
    x = rnd(gen, 77, 64)
    x[3] = 0
    got = ops.l2_normalize_rows_(x.clone().to(DEV))
    close(got, F.normalize(x, dim=1), 1e-6, 'normalize')


@pytest.mark.parametrize('mode,K,c', [(0, 4, 16), (1, 4, 16), (0, 8, 16), (1, 4, 12)])
def test_attention_block(ops, mode, K, c):
    """AttentionBlock.forward through the HIP path vs the oracle restatement (model/attention.py:84-113)."""
    import contextlib, io
    from model.attention import AttentionBlock
    gen = torch.Generator().manual_seed(11 + K + c + mode)
    b, e = 600, 2
    with contextlib.redirect_stdout(io.StringIO()):
        blk = AttentionBlock(c, e, K, True, True, bool(mode), True, True)
    sd = {k: rnd(gen, *v.shape, scale=0.15) for k, v in blk.state_dict().items()}
    blk.load_state_dict(sd)
    blk.to(DEV)
    x = rnd(gen, b, c, e, e, e).relu_()
    p = rnd(gen, b, K, c, e, e, e).relu_()
    p[:, 0] = x + 0.05 * rnd(gen, b, c, e, e, e)             # one close candidate -> sharp softmax with real mixing
    noise = -torch.empty(b, K).exponential_(generator=gen).log() if mode else None
    det = {}
    with torch.no_grad():
        ref = refpath.attention_block(x, p, {'a.' + k: v for k, v in sd.items()}, 'a', bool(mode), noise, det)
        dbg = {}
        got = blk(x.to(DEV), p.to(DEV), noise.to(DEV) if noise is not None else None, dbg)
    close(dbg['scores'], det['scores'], 2e-6, 'scores')
    if mode:
        top2 = torch.topk(det['scores'] * 25 + noise, 2, dim=1).values
        safe = (top2[:, 0] - top2[:, 1]) > 1e-4              # rows whose hard arg-max cannot flip by rounding
        assert safe.float().mean() > 0.99
        close(got.cpu()[safe], ref[safe], 1e-5, 'gumbel-hard attention')
        assert torch.equal(dbg['weights'].cpu()[safe].round(), det['weights'][safe].round())
    else:
        # sharpness 1024 amplifies score rounding: weights agree to ~1e-3 relative, outputs accordingly
        close(dbg['weights'], det['weights'], 2e-3, 'softmax weights')
        close(got, ref, 2e-3, 'softmax attention')


def test_attn_gather_layouts(ops):
    gen = torch.Generator().manual_seed(2)
    b, K, c, s, e, t = 2, 4, 3, 16, 2, 8
    vols = rnd(gen, b * K, c, s, s, s)
    r = s // e
    ref = refpath.unfold3d(vols, e).reshape(b, K, r, r, r, c, e, e, e).permute(0, 2, 3, 4, 1, 5, 6, 7, 8).reshape(-1, K, c, e, e, e)
    got0 = ops.attn_gather_retrieved(vols.to(DEV), 0, b, K, c, s, e)
    assert torch.equal(got0.cpu(), ref)
    patch_major = refpath.unfold3d(vols, t)                   # what the retrieval backbone emits: [(b*K*q^3), c, t^3]
    got1 = ops.attn_gather_retrieved(patch_major.to(DEV), 1, b, K, c, s, e, t)
    assert torch.equal(got1.cpu(), ref)


def _seeded_encoder(gen, n_in_channels, e, scale=0.15):
    import contextlib, io
    from model.attention import AttentionFeatureEncoder
    with contextlib.redirect_stdout(io.StringIO()):
        enc = AttentionFeatureEncoder(n_in_channels, 32, e)
    sd = {k: rnd(gen, *v.shape, scale=scale) for k, v in enc.state_dict().items()}
    enc.load_state_dict(sd)
    return enc.to(DEV), sd


@pytest.mark.parametrize('rows,c', [(1000, 16), (16, 12), (5000, 12), (1, 2), (257, 16)])
def test_attn_mlp_rows_matches_float64(ops, rows, c):
    """Fused 4-layer AttentionFeatureEncoder (rf_attn_mlp_rows) vs float64 (model/attention.py:36-46)."""
    gen = torch.Generator().manual_seed(rows + c)
    enc, sd = _seeded_encoder(gen, c, 2)
    assert enc.fusable()
    x = rnd(gen, rows, c * 8)
    h = x.double()
    for i in (0, 2, 4, 6):
        h = h @ sd['encoder.%d.weight' % i].double().t() + sd['encoder.%d.bias' % i].double()
        if i < 6:
            h = F.leaky_relu(h, 0.01)
    with torch.no_grad():
        got = enc(x.to(DEV))                                       # the split-operand form (ops.CONV_ARITH == 'split')
        ops.CONV_ARITH = 'fp32'
        try:
            got32 = enc(x.to(DEV))
        finally:
            ops.CONV_ARITH = 'split'
        ops.USE_FUSED_ATTN_MLP = False
        try:
            layered = enc(x.to(DEV))
        finally:
            ops.USE_FUSED_ATTN_MLP = True
    close(got, h.float(), 2e-6, 'fused MLP, split-operand form')
    close(got32, h.float(), 2e-6, 'fused MLP, fp32 MFMA form')
    close(got, layered, 2e-6, 'fused vs per-layer rf_linear')
    e_s = (got.cpu().double() - h).pow(2).mean().sqrt().item()
    e_f = (got32.cpu().double() - h).pow(2).mean().sqrt().item()
    print(f'\nrows {rows} c {c}: rms error vs float64  split {e_s:.3e} | fp32 MFMA {e_f:.3e}')
    assert e_s <= 1.05 * e_f + 1e-9


@pytest.mark.parametrize('b,kv,c,s,t', [(2, 1, 16, 32, 32), (2, 4, 16, 32, 8), (1, 8, 12, 32, 8), (3, 2, 4, 8, 4), (1, 1, 2, 4, 2)])
def test_attn_mlp_volume_equals_rows_on_unfolded(ops, b, kv, c, s, t):
    """Reading the 2^3 attention patches in place (NCDHW or patch-major) == the same encoder on the materialised rows, bit for bit."""
    gen = torch.Generator().manual_seed(b * 7 + kv + c + s + t)
    enc, _ = _seeded_encoder(gen, c, 2)
    vols = rnd(gen, b * kv, c, s, s, s)
    r3 = (s // 2) ** 3
    rows = refpath.unfold3d(vols, 2).reshape(b, kv, r3, c * 8).permute(0, 2, 1, 3).reshape(-1, c * 8).contiguous()
    src = vols if t == s else refpath.unfold3d(vols, t)
    with torch.no_grad():
        want = ops.attn_mlp_rows(rows.to(DEV), enc.packed_fused())
        got = ops.attn_mlp_volume(src.to(DEV), b, kv, c, s, t, enc.packed_fused())
    assert got.shape == want.shape
    assert torch.equal(got, want)


@pytest.mark.parametrize('mode,K,c,s,t', [(1, 4, 16, 32, 8), (0, 4, 16, 32, 8), (0, 8, 16, 32, 32), (1, 4, 12, 16, 8), (0, 3, 2, 4, 2)])
def test_attn_weights_and_blend_equal_row_domain_kernels(ops, mode, K, c, s, t):
    """rf_attn_weights + rf_attn_blend (folded layout) == unfold -> regroup -> rf_attn_fuse -> fold on the same features, bit for bit."""
    gen = torch.Generator().manual_seed(mode + K + c + s + t)
    b, e = 2, 2
    r = s // e
    rows = b * r ** 3
    x = rnd(gen, b, c, s, s, s).relu_()
    vols = rnd(gen, b * K, c, s, s, s).relu_()
    xf, pf = rnd(gen, rows, 32), rnd(gen, rows * K, 32)
    pf[::K] = xf + 0.02 * rnd(gen, rows, 32)                 # candidate 0 close to the query: switch > 0, real mixing
    noise = (-torch.empty(rows, K).exponential_(generator=gen).log()).to(DEV) if mode else None
    sharp = 25.0 if mode else 1024.0
    src = (vols if t == s else refpath.unfold3d(vols, t)).to(DEV)
    xd, xfd, pfd = x.to(DEV), xf.to(DEV), pf.to(DEV)
    w, sw, sc = ops.attn_weights(xfd, pfd, noise, K, mode, sharp, debug=True)
    got = ops.attn_blend(xd, src, K, t, w, sw)
    x_rows = ops.unfold3d(xd, e)
    p_rows = ops.attn_gather_retrieved(src, 0 if t == s else 1, b, K, c, s, e, t)
    out_rows, sc_old, w_old = ops.attn_fuse(x_rows, p_rows, xfd, pfd, noise, mode, sharp, debug=True)
    want = ops.fold3d(out_rows, r, e, c)
    assert torch.equal(sc, sc_old) and torch.equal(w, w_old)
    assert (sw > 0).float().mean() > 0.5
    assert torch.equal(got, want)


def test_patched_attention_volume_route_vs_row_route(ops):
    """PatchedAttentionBlock.forward_patch_major: volume-domain route vs the materialised-rows route (softmax mode)."""
    import contextlib, io
    from model.attention import AttentionBlock, PatchedAttentionBlock
    gen = torch.Generator().manual_seed(77)
    b, K, c, s, t = 2, 4, 16, 32, 8
    with contextlib.redirect_stdout(io.StringIO()):
        pab = PatchedAttentionBlock(c, 16, 2, K, AttentionBlock(c, 2, K, True, True, False, True, True))
    pab.load_state_dict({k: rnd(gen, *v.shape, scale=0.15) for k, v in pab.state_dict().items()})
    pab.to(DEV)
    x = rnd(gen, b, c, s, s, s).relu_()
    vols = rnd(gen, b * K, c, s, s, s).relu_()
    vols[::K] = x + 0.05 * rnd(gen, b, c, s, s, s)
    feats = refpath.unfold3d(vols, t).to(DEV)
    with torch.no_grad():
        assert pab.attention_blocks_layer.volume_route_ok()
        got = pab.forward_patch_major(x.to(DEV), feats, t)
        ops.USE_FUSED_ATTN_MLP = False
        try:
            assert not pab.attention_blocks_layer.volume_route_ok()
            want = pab.forward_patch_major(x.to(DEV), feats, t)
        finally:
            ops.USE_FUSED_ATTN_MLP = True
    close(got, want, 2e-3, 'volume route vs row route')     # sharpness 1024 amplifies the rounding of the two MLP forms


@pytest.mark.parametrize('cfg_name', ['C1', 'C4', 'C5'])
def test_query_windows_bit_exact(ops, cfg_name):
    from rfuse import configs, synthetic
    cfg = configs.get_config(cfg_name)
    trunc_i, _ = configs.truncations(cfg)
    raws = np.stack([synthetic.make_chunk(50 + i, cfg)['input_raw'] for i in range(2)])
    ref = np.concatenate([refpath.extract_query_windows(r, cfg, trunc_i) for r in raws])
    g, d = cfg['query_geometry'], cfg['dataset_train']
    pad = 0.0 if cfg['task'] == 'surface_reconstruction' else trunc_i
    got = ops.query_windows(torch.from_numpy(raws).to(DEV), g['patch_size_input'], g['patch_context_input'], pad, d['input_mean'], d['input_std'])
    assert np.array_equal(got.cpu().numpy(), ref)


def check_topk(idx_got, dist_got, q, emb, k2, row_base=0):
    """bit-exact indices vs the float64 oracle; where the float64 gap to the next neighbour is < 1e-6 either order is
    allowed (SURVEY.md section 7 'hard parts')."""
    idx_ref, dist_ref = refpath.knn_exact(q, emb, min(k2 + 1, emb.shape[0]))
    idx_got = idx_got.cpu().numpy() - row_base
    nq = q.shape[0]
    n_valid = min(k2, emb.shape[0])
    exact = 0
    for i in range(nq):
        for j in range(n_valid):
            if idx_got[i, j] == idx_ref[i, j]:
                exact += 1
                continue
            d_here = ((q[i].astype(np.float64) - emb[idx_got[i, j]].astype(np.float64)) ** 2).sum()
            assert abs(d_here - dist_ref[i, j]) < 1e-6, f'query {i} rank {j}: got row {idx_got[i, j]} (d={d_here}), want {idx_ref[i, j]} (d={dist_ref[i, j]})'
    np.testing.assert_allclose(dist_got.cpu().numpy()[:, :n_valid], dist_ref[:, :n_valid], rtol=1e-5, atol=1e-6)
    return exact / (nq * n_valid)


@pytest.mark.parametrize('algo', [1, 2, 3])
@pytest.mark.parametrize('n,nq,k2', [(1000, 64, 8), (50_001, 128, 8), (4097, 70, 16), (5, 3, 8), (64, 1, 8), (131, 300, 8)])
def test_l2_topk(ops, n, nq, k2, algo):
    """the scans (1 = VALU, every pair exact; 2 / 3 = fp32- / f16-MFMA dot-product filter + exact re-check) against the float64 oracle"""
    rng = np.random.default_rng(n + nq)
    emb = rng.standard_normal((n, 64)).astype(np.float32)
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    q = rng.standard_normal((nq, 64)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[0] = emb[min(7, n - 1)]                                   # an exact hit (distance 0)
    if n > 10:
        emb[9] = emb[4]                                         # an exact tie: lower row id must win
        q[1 % nq] = emb[4]
    packed = ops.db_pack_embeddings(torch.from_numpy(emb).to(DEV))
    dist, idx = ops.l2_topk(torch.from_numpy(q).to(DEV), packed, n, 0, k2, algo)
    frac = check_topk(idx, dist, q, emb, k2)
    assert frac > 0.999
    if n < k2:
        assert (idx.cpu()[:, n:] == -1).all() and torch.isinf(dist.cpu()[:, n:]).all()
    if n > 10:
        row = idx.cpu().numpy()[1 % nq]
        assert row[0] == 4 and row[1] == 9


@pytest.mark.parametrize('mfma_algo', [2, 3])
@pytest.mark.parametrize('n,nq,k2,kind', [(20_000, 200, 8, 'unit'), (70_001, 64, 16, 'unit'), (30_000, 100, 8, 'dups'), (9_000, 130, 8, 'big'),
                                          (300_000, 512, 8, 'unit'), (12_000, 90, 8, 'huge'), (120_000, 300, 8, 'clustered'), (40_000, 130, 16, 'clustered')])
def test_mfma_filtered_scan_equals_exact_scan_bit_for_bit(ops, n, nq, k2, kind, mfma_algo):
    """The matrix-core dot product only FILTERS; every survivor is re-evaluated with the one exact distance of the path, so the
    MFMA scan must return the same bits (distances and row ids) as the VALU scan that evaluates every pair exactly -- on unit
    vectors, on a database full of duplicate rows (ties at the list threshold, lower row id wins), and on rows of very
    different norms (the filter margin scales with |q|^2 + |x|^2; 'huge': some rows and queries beyond the f16 range, which the
    f16 filter (algo 3) must hand to the exact re-check unfiltered).  Also through the packed-key outputs."""
    rng = np.random.default_rng(n + nq)
    emb = rng.standard_normal((n, 64)).astype(np.float32)
    q = rng.standard_normal((nq, 64)).astype(np.float32)
    if kind == 'clustered':
        # a database that lies where the queries lie (what a trained encoder pair gives, and bench.py's construction): queries within ~0.3 of one direction,
        # rows = a random query + noise of the queries' nearest-neighbour spacing -- squared distances to the nearest rows ~1e-2, where a filter on f16-rounded
        # operands alone (rounds 3-5: slack 4e-3) passed every row of the neighbourhood
        q = (rng.standard_normal((1, 64)) + 0.04 * rng.standard_normal((nq, 64))).astype(np.float32)
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        emb = (q[rng.integers(0, nq, size=n)] + 0.011 * rng.standard_normal((n, 64))).astype(np.float32)
        emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    elif kind == 'huge':
        emb *= (10.0 ** rng.uniform(0, 5.5, size=(n, 1))).astype(np.float32)
        q *= (10.0 ** rng.uniform(2, 5.5, size=(nq, 1))).astype(np.float32)
    elif kind != 'big':
        emb /= np.linalg.norm(emb, axis=1, keepdims=True)
        q /= np.linalg.norm(q, axis=1, keepdims=True)
    else:
        emb *= (10.0 ** rng.uniform(-2, 2, size=(n, 1))).astype(np.float32)
        q *= (10.0 ** rng.uniform(-1, 1, size=(nq, 1))).astype(np.float32)
    if kind == 'dups':
        emb[rng.integers(0, n, size=n // 2)] = emb[17]          # half the database is one row: massive ties
        emb[n // 3: n // 3 + 3000] = emb[5]
        q[0], q[1] = emb[17], emb[5]
        q[2] = emb[17] + 1e-4 * rng.standard_normal(64).astype(np.float32)
    qd = torch.from_numpy(q).to(DEV)
    packed = ops.db_pack_embeddings(torch.from_numpy(emb).to(DEV))
    d1, i1 = ops.l2_topk(qd, packed, n, 1000, k2, ops.TOPK_VALU_SCAN)
    d2, i2 = ops.l2_topk(qd, packed, n, 1000, k2, mfma_algo)
    assert torch.equal(i1, i2), f'{(i1 != i2).sum().item()} row ids differ'
    assert torch.equal(d1, d2)
    keys = ops.l2_topk_keys(qd, packed, n, 1000, k2, mfma_algo)
    assert torch.equal(keys & 0xffffffff, i1) and torch.equal((keys >> 32).to(torch.int32).view(torch.float32), d1)
    dm, im = ops.topk_merge_keys(keys[None].contiguous())
    assert torch.equal(im, i1) and torch.equal(dm, d1)
    if kind == 'dups':
        dup_rows = np.sort(np.where((emb == emb[17]).all(axis=1))[0])[:k2] + 1000
        assert i1[0].cpu().tolist() == dup_rows.tolist()         # ties resolved towards the lower row id


@pytest.mark.parametrize('algo', [1, 2, 3])
def test_sharded_topk_merge_equals_single_scan(ops, algo):
    rng = np.random.default_rng(1)
    n, nq, k2, shards = 10_000, 96, 8, 4
    emb = rng.standard_normal((n, 64)).astype(np.float32)
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    q = rng.standard_normal((nq, 64)).astype(np.float32)
    qd = torch.from_numpy(q).to(DEV)
    full_d, full_i = ops.l2_topk(qd, ops.db_pack_embeddings(torch.from_numpy(emb).to(DEV)), n, 0, k2, algo)
    from rfuse.database import shard_bounds
    ds, is_, ks = [], [], []
    for r in range(shards):
        lo, hi = shard_bounds(n, r, shards)
        packed = ops.db_pack_embeddings(torch.from_numpy(emb[lo:hi]).to(DEV))
        d, i = ops.l2_topk(qd, packed, hi - lo, lo, k2, {1: 3, 2: 1, 3: 2}[algo])        # the shards with ANOTHER scan: same bits
        ds.append(d), is_.append(i), ks.append(ops.l2_topk_keys(qd, packed, hi - lo, lo, k2, algo))
    md, mi = ops.topk_merge(torch.stack(ds), torch.stack(is_))
    assert torch.equal(mi, full_i) and torch.equal(md, full_d)
    md, mi = ops.topk_merge_keys(torch.stack(ks))
    assert torch.equal(mi, full_i) and torch.equal(md, full_d)


def test_topk_million_row_database(ops):
    """BASELINE config 3's database size on one device: 1 000 001 rows, 2048 queries (a 32-chunk batch).  A 256-query subset is
    checked against the float64 oracle; the single scan must equal the merge of 8 shard scans (what 8 GPUs exchange)."""
    n, nq, k2 = 1_000_001, 2048, 8
    g = torch.Generator(device=DEV).manual_seed(7)
    emb = torch.randn(n, 64, generator=g, device=DEV)
    emb = emb / emb.norm(dim=1, keepdim=True)
    q = torch.randn(nq, 64, generator=g, device=DEV)
    q = q / q.norm(dim=1, keepdim=True)
    q[3] = emb[n - 1]                                            # the very last row is reachable
    packed = ops.db_pack_embeddings(emb)
    dist, idx = ops.l2_topk(q, packed, n, 0, k2)
    assert idx[3, 0].item() == n - 1 and dist[3, 0].item() == 0.0
    sub = torch.arange(0, nq, 8, device=DEV)
    d64 = torch.cdist(q[sub].double(), emb.double()) ** 2        # float64 brute force on the device (oracle arithmetic, knn_exact)
    ref_d, ref_i = torch.topk(d64, k2, dim=1, largest=False, sorted=True)
    got_i, got_d = idx[sub], dist[sub]
    same = got_i == ref_i
    # where the order differs the float64 gap must be below fp32 resolution (near-tie), as in check_topk
    gap = (torch.gather(d64, 1, got_i) - ref_d).abs()
    assert (same | (gap < 1e-6)).all()
    assert same.float().mean().item() > 0.999
    assert (got_d.double() - ref_d).abs().max().item() < 1e-5
    from rfuse.database import shard_bounds
    ks = []
    for r in range(8):
        lo, hi = shard_bounds(n, r, 8)
        ks.append(ops.l2_topk_keys(q, ops.db_pack_embeddings(emb[lo:hi].contiguous()), hi - lo, lo, k2))
    md, mi = ops.topk_merge_keys(torch.stack(ks))
    assert torch.equal(mi, idx) and torch.equal(md, dist)


def test_demote_and_gather_match_oracle(ops):
    from rfuse import configs, synthetic
    cfg = configs.get_config('C1')
    _, trunc_t = configs.truncations(cfg)
    K = cfg['K']
    db = synthetic.make_database(5, cfg, 64 * 12)
    rng = np.random.default_rng(8)
    q = rng.standard_normal((128, 64)).astype(np.float32)
    own = np.where(db['meta'][:, 0] == 3)[0]
    for i in range(0, 64, 2):
        q[i] = db['emb'][own[i]] + 0.05 * rng.standard_normal(64).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    qscene = np.concatenate([np.full(64, 3), np.full(64, -1)]).astype(np.int32)      # chunk 0 demotes, chunk 1 does not
    idx, dist = refpath.knn_exact(q, db['emb'], 2 * K)
    rows = refpath.mapping_rows(idx, dist, db['meta'])
    ref_map = refpath.demote_same_scene(rows, qscene, K)
    meta_d = torch.from_numpy(db['meta']).to(DEV)
    m, d, i = ops.demote_same_scene(torch.from_numpy(dist.astype(np.float32)).to(DEV), torch.from_numpy(idx).to(DEV), meta_d,
                                    torch.from_numpy(qscene).to(DEV), K)
    np.testing.assert_array_equal(m.cpu().numpy(), ref_map[..., :7].astype(np.int32))
    np.testing.assert_array_equal(d.cpu().numpy(), ref_map[..., 7])
    # a sentinel hit exercises the trunc fill
    m[5, 1] = torch.tensor([-1, 0, 16, 0, 16, 0, 16], dtype=torch.int32)
    ref_map[5, 1, :7] = [-1, 0, 16, 0, 16, 0, 16]
    dd = cfg['dataset_train']
    vols = torch.from_numpy(db['volumes']).to(DEV)
    composed = ops.gather_patches(vols, m, 2, K, trunc_t, 1.0, dd['target_mean'], dd['target_std'], layout=0).cpu().numpy()
    rows16 = ops.gather_patches(vols, m, 2, K, trunc_t, 1.0, dd['target_mean'], dd['target_std'], layout=1).cpu()
    for c in range(2):
        ref = refpath.compose_retrieval(ref_map[c * 64:(c + 1) * 64], db['volumes'], K, trunc_t)
        ref = ((ref - np.float32(dd['target_mean'])) / np.float32(dd['target_std'])).astype(np.float32)    # patched_scene_dataset.py:133
        assert np.array_equal(composed[c], ref)
    assert torch.equal(rows16, refpath.unfold3d(torch.from_numpy(composed).reshape(2 * K, 1, 64, 64, 64), 16))
    # database boxes OFF the 16-voxel grid (any extent is legal in a mapping row, util/retrieval.py:152): origins with z0 % 4 != 0 take the element-wise route, the others the
    # 16-byte route -- both against the restatement, with the fp32 and the float16 voxel store
    m2 = m.clone()
    rng2 = np.random.default_rng(9)
    o = rng2.integers(0, 49, size=(m2.shape[0], m2.shape[1], 3))
    o[::3, :, 2] = (o[::3, :, 2] // 4) * 4                                                 # a third of them 16-byte aligned in z
    m2[:, :, 1] = torch.from_numpy(o[..., 0]).int(); m2[:, :, 2] = m2[:, :, 1] + 16
    m2[:, :, 3] = torch.from_numpy(o[..., 1]).int(); m2[:, :, 4] = m2[:, :, 3] + 16
    m2[:, :, 5] = torch.from_numpy(o[..., 2]).int(); m2[:, :, 6] = m2[:, :, 5] + 16
    ref_map2 = ref_map.copy()
    ref_map2[..., :7] = m2.cpu().numpy()
    for store in (vols, vols.half()):
        got2 = ops.gather_patches(store, m2, 2, K, trunc_t, 1.0, dd['target_mean'], dd['target_std'], layout=0).cpu().numpy()
        for c in range(2):
            ref = refpath.compose_retrieval(ref_map2[c * 64:(c + 1) * 64], db['volumes'], K, trunc_t)
            assert np.array_equal(got2[c], ((ref - np.float32(dd['target_mean'])) / np.float32(dd['target_std'])).astype(np.float32))


@pytest.mark.parametrize('spec', [(3, 1, 8, 16, 3, 1), (2, 4, 12, 8, 3, 2), (2, 3, 6, 5, 2, 1), (1, 1, 20, 6, 5, 1), (2, 8, 4, 8, 4, 1)])
def test_conv3d_valid_leaky(ops, spec):
    n, cin, s, cout, k, stride = spec
    gen = torch.Generator().manual_seed(sum(spec))
    x, w, b = rnd(gen, n, cin, s, s, s), rnd(gen, cout, cin, k, k, k, scale=1 / np.sqrt(cin * k ** 3)), rnd(gen, cout)
    ref = F.leaky_relu(F.conv3d(x, w, b, stride=stride), 0.2)
    close(ops.conv3d_valid_leaky(x.to(DEV), w.to(DEV), b.to(DEV), stride, 0.2), ref, 1e-5, 'valid conv')


@pytest.mark.parametrize('spec', [(3, 1, 8, 16, 3, 1), (2, 4, 12, 8, 3, 2), (2, 3, 6, 5, 2, 1), (1, 1, 20, 6, 5, 1), (2, 8, 4, 8, 4, 1), (2, 12, 13, 24, 3, 1),
                                  (3, 24, 11, 48, 3, 2), (5, 96, 2, 96, 2, 1), (1, 1, 48, 12, 5, 1)])
def test_conv3d_valid_leaky_mfma(ops, spec):
    """matrix-core form of the patch encoders' layers vs torch (odd edges, strides, cin = 1, cout not a multiple of 16)"""
    n, cin, s, cout, k, stride = spec
    gen = torch.Generator().manual_seed(sum(spec) + 1)
    x, w, b = rnd(gen, n, cin, s, s, s), rnd(gen, cout, cin, k, k, k, scale=1 / np.sqrt(cin * k ** 3)), rnd(gen, cout)
    ref = F.leaky_relu(F.conv3d(x, w, b, stride=stride), 0.2)
    got = ops.conv3d_valid_leaky_mfma(x.to(DEV), ops.pack_convv_weight(w.to(DEV)), b.to(DEV), cout, k, stride, 0.2)
    close(got, ref, 1e-5, 'valid conv (mfma)')


@pytest.mark.parametrize('spec', [(3, 1, 48, 12, 5, 1), (2, 12, 44, 24, 3, 1), (2, 24, 42, 48, 3, 2), (3, 48, 20, 48, 3, 2), (2, 1, 32, 8, 5, 1),
                                  (2, 8, 28, 16, 3, 1), (3, 16, 26, 32, 3, 2), (2, 32, 12, 64, 3, 1), (2, 12, 24, 24, 3, 1), (1, 1, 24, 12, 3, 1),
                                  (2, 24, 22, 24, 3, 2), (5, 16, 11, 20, 2, 1), (2, 4, 13, 24, 4, 1), (3, 8, 17, 96, 3, 1)])
def test_conv3d_valid_leaky_lds(ops, spec):
    """LDS-staged form of the patch encoders' large layers (every layer shape of PCPatch48 / Patch32 / Patch24V2 with an output
    edge >= 8, plus odd edges, k = 2 / 4, ragged last tiles, cout not a multiple of 16, two cout blocks) vs torch and vs the
    gather form"""
    n, cin, s, cout, k, stride = spec
    gen = torch.Generator().manual_seed(sum(spec) + 2)
    x, w, b = rnd(gen, n, cin, s, s, s), rnd(gen, cout, cin, k, k, k, scale=1 / np.sqrt(cin * k ** 3)), rnd(gen, cout)
    xd = x.to(DEV)
    assert ops.conv_valid_lds_supported(xd, cout, k, stride)
    ref = F.leaky_relu(F.conv3d(x.double(), w.double(), b.double(), stride=stride), 0.2).float()
    got = ops.conv3d_valid_leaky_lds(xd, ops.pack_convv_lds_weight(w.to(DEV)), b.to(DEV), cout, k, stride, 0.2)
    close(got, ref, 1e-5, 'valid conv (lds)')
    gather = ops.conv3d_valid_leaky_mfma(xd, ops.pack_convv_weight(w.to(DEV)), b.to(DEV), cout, k, stride, 0.2)
    close(got, gather, 1e-5, 'lds form vs gather form')


@pytest.mark.parametrize('spec', [(3, 1, 48, 12, 5), (2, 12, 44, 24, 3), (3, 1, 32, 8, 5), (2, 8, 28, 16, 3), (2, 1, 24, 12, 3), (3, 12, 22, 24, 3),
                                  (2, 1, 16, 16, 3), (5, 1, 13, 8, 3), (2, 12, 11, 24, 3)])
def test_conv3d_valid_leaky_valu(ops, spec):
    """packed-fp32 VALU form of the patch encoders' first layers (PCPatch48 / Patch32 / Patch24V2 / Patch16 shapes, odd edges and
    ragged last tiles) vs float64 torch and vs the gather-form MFMA kernel"""
    n, cin, s, cout, k = spec
    gen = torch.Generator().manual_seed(sum(spec) + 3)
    x, w, b = rnd(gen, n, cin, s, s, s), rnd(gen, cout, cin, k, k, k, scale=1 / np.sqrt(cin * k ** 3)), rnd(gen, cout)
    xd = x.to(DEV)
    assert ops.conv_valid_valu_supported(xd, cout, k, 1)
    ref = F.leaky_relu(F.conv3d(x.double(), w.double(), b.double()), 0.2).float()
    got = ops.conv3d_valid_leaky_valu(xd, ops.pack_convv_valu_weight(w.to(DEV)), b.to(DEV), 1, 0.2)
    close(got, ref, 1e-5, 'valid conv (valu)')
    gather = ops.conv3d_valid_leaky_mfma(xd, ops.pack_convv_weight(w.to(DEV)), b.to(DEV), cout, k, 1, 0.2)
    close(got, gather, 1e-5, 'valu form vs gather form')


@pytest.mark.parametrize('spec', [(2, 12, 44, 24, 3, 1), (2, 24, 42, 48, 3, 2), (3, 48, 20, 48, 3, 2), (2, 8, 28, 16, 3, 1), (3, 16, 26, 32, 3, 2),
                                  (2, 32, 12, 64, 3, 1), (2, 12, 24, 24, 3, 1), (2, 24, 22, 24, 3, 2), (5, 16, 11, 20, 2, 1), (2, 4, 13, 24, 4, 1),
                                  (3, 8, 17, 96, 3, 1), (1, 4, 16, 8, 5, 1), (2, 20, 15, 40, 3, 1)])
def test_conv3d_valid_leaky_split(ops, spec):
    """split-operand F16-MFMA form of the patch encoders' large layers (PCPatch48 / Patch32 / Patch24V2 shapes with cin a multiple
    of 4, plus odd edges, k = 2 / 4 / 5, ragged last tiles, several chunks, cout not a multiple of 16, two cout-block groups) vs
    float64 torch: no further from it than the fp32-MFMA LDS form"""
    n, cin, s, cout, k, stride = spec
    gen = torch.Generator().manual_seed(sum(spec) + 4)
    x, w, b = rnd(gen, n, cin, s, s, s), rnd(gen, cout, cin, k, k, k, scale=1 / np.sqrt(cin * k ** 3)), rnd(gen, cout)
    xd = x.to(DEV)
    assert ops.conv_valid_split_supported(xd, cout, k, stride)
    ref = F.leaky_relu(F.conv3d(x.double(), w.double(), b.double(), stride=stride), 0.2)
    got = ops.conv3d_valid_leaky_split(xd, ops.pack_convv_split_weight(w.to(DEV), s, stride), b.to(DEV), cout, k, stride, 0.2)
    close(got, ref.float(), 1e-5, 'valid conv (split)')
    if ops.conv_valid_lds_supported(xd, cout, k, stride):
        fp32 = ops.conv3d_valid_leaky_lds(xd, ops.pack_convv_lds_weight(w.to(DEV)), b.to(DEV), cout, k, stride, 0.2)
    else:
        fp32 = ops.conv3d_valid_leaky_mfma(xd, ops.pack_convv_weight(w.to(DEV)), b.to(DEV), cout, k, stride, 0.2)
    e_split, e_fp32 = (got.cpu().double() - ref).abs(), (fp32.cpu().double() - ref).abs()
    print(f'valid split {spec}: rms {e_split.pow(2).mean().sqrt():.3e} (fp32 form {e_fp32.pow(2).mean().sqrt():.3e}), '
          f'max {e_split.max():.3e} ({e_fp32.max():.3e})')
    assert e_split.pow(2).mean().sqrt() <= 1.05 * e_fp32.pow(2).mean().sqrt()
    assert e_split.max() <= 1.25 * e_fp32.max()


@pytest.mark.parametrize('spec', [(1, 12, 140, 24, 3, 1), (1, 24, 138, 48, 3, 2), (2, 48, 68, 48, 3, 2), (1, 8, 76, 16, 3, 1), (2, 16, 74, 32, 3, 2), (1, 32, 36, 64, 3, 1),
                                  (1, 8, 70, 16, 3, 1)])
def test_conv3d_valid_leaky_split_on_big_volumes(ops, spec):
    """the x-tiled form of the split valid conv -- the layers of PCPatch48 / Patch32 on a whole padded chunk (140^3 ... 36^3, model/retrieval.py
    forward_grid) and an edge that leaves ragged tiles in x, y and z -- vs float64 torch, and vs the same kernel run on windows of the volume
    (the per-output arithmetic does not depend on the tile)"""
    n, cin, s, cout, k, stride = spec
    gen = torch.Generator().manual_seed(sum(spec) + 5)
    x, w, b = rnd(gen, n, cin, s, s, s), rnd(gen, cout, cin, k, k, k, scale=1 / np.sqrt(cin * k ** 3)), rnd(gen, cout)
    xd = x.to(DEV)
    assert ops.conv_valid_split_supported(xd, cout, k, stride)
    got = ops.conv3d_valid_leaky_split(xd, ops.pack_convv_split_weight(w.to(DEV), s, stride), b.to(DEV), cout, k, stride, 0.2)
    ref = F.leaky_relu(F.conv3d(x.double(), w.double(), b.double(), stride=stride), 0.2)
    err = (got.cpu().double() - ref).abs()
    print(f'valid split on {s}^3 {spec}: rms {err.pow(2).mean().sqrt():.3e} max {err.max():.3e}')
    close(got, ref.float(), 1e-5, 'valid conv (split, big volume)')
    # a 44-voxel window of the same volume through the whole-row plan of that size: same products, same order within a chunk of channels
    sw = 44 if s > 60 else 20
    o0 = 8 * stride
    xw = xd[:, :, o0:o0 + sw, o0:o0 + sw, o0:o0 + sw].contiguous()
    if ops.conv_valid_split_supported(xw, cout, k, stride):
        gw = ops.conv3d_valid_leaky_split(xw, ops.pack_convv_split_weight(w.to(DEV), sw, stride), b.to(DEV), cout, k, stride, 0.2)
        so_w = gw.shape[-1]
        sub = got[:, :, 8:8 + so_w, 8:8 + so_w, 8:8 + so_w]
        close(sub, gw.cpu(), 2e-6, 'grid vs window evaluation')


@pytest.mark.parametrize('spec', [(2, 144, 12, 5), (1, 80, 8, 5), (1, 72, 16, 3), (1, 68, 12, 3)])
def test_conv3d_valid_leaky_valu_on_big_volumes(ops, spec):
    """the x-tiled VALU form (a patch encoder's first layer on a whole padded chunk) vs float64 torch; BIT-equal to the window kernel on a window"""
    n, s, cout, k = spec
    gen = torch.Generator().manual_seed(sum(spec) + 6)
    x, w, b = rnd(gen, n, 1, s, s, s), rnd(gen, cout, 1, k, k, k, scale=1 / np.sqrt(k ** 3)), rnd(gen, cout)
    xd = x.to(DEV)
    assert ops.conv_valid_valu_supported(xd, cout, k, 1)
    wt = ops.pack_convv_valu_weight(w.to(DEV))
    got = ops.conv3d_valid_leaky_valu(xd, wt, b.to(DEV), 1, 0.2)
    ref = F.leaky_relu(F.conv3d(x.double(), w.double(), b.double()), 0.2).float()
    close(got, ref, 1e-5, 'valid conv (valu, big volume)')
    for o0 in (0, s - 48, 16):
        xw = xd[:, :, o0:o0 + 48, o0:o0 + 48, o0:o0 + 48].contiguous()
        gw = ops.conv3d_valid_leaky_valu(xw, wt, b.to(DEV), 1, 0.2)
        so_w = gw.shape[-1]
        assert torch.equal(got[:, :, o0:o0 + so_w, o0:o0 + so_w, o0:o0 + so_w], gw)


def test_gather_windows(ops):
    gen = torch.Generator().manual_seed(11)
    grid = rnd(gen, 2, 5, 33, 33, 33)
    got = ops.gather_windows(grid.to(DEV), 9, 8, 4).cpu()
    ref = grid.unfold(2, 9, 8).unfold(3, 9, 8).unfold(4, 9, 8).permute(0, 2, 3, 4, 1, 5, 6, 7).reshape(2 * 64, 5, 9, 9, 9)
    assert torch.equal(got, ref)
    with pytest.raises(RuntimeError, match='last window'):
        ops.gather_windows(grid.to(DEV), 9, 9, 4)


@pytest.mark.parametrize('enc', [('PCPatch48', 12, 48, 32), ('Patch32', 8, 32, 16), ('Patch24V2', 8, 24, 16), ('Patch16', 8, 16, 16)])
def test_patch_encoder_on_the_grid_equals_the_encoder_on_the_windows(ops, enc):
    """forward_grid (leading layers once on the padded chunk, windows cut out of the feature grid) against forward on the 64 windows: the same
    embeddings (the VALU layers bit for bit; the split layers may pick another channel chunking for the big volume: <= 2e-6 of the scale)"""
    import model as rf_model
    name, nf, window, step = enc
    torch.manual_seed(5)
    m = getattr(rf_model, name)(nf, 64).to(DEV).eval()
    g = 3 * step + window
    gen = torch.Generator().manual_seed(9)
    grid = rnd(gen, 2, 1, g, g, g).to(DEV)
    with torch.no_grad():
        win = grid[:, 0].unfold(1, window, step).unfold(2, window, step).unfold(3, window, step).reshape(2 * 64, 1, window, window, window).contiguous()
        ref = m(win)
        got = m.forward_grid(grid, window, step)
    on_grid = m.grid_plan(window, step, 4)[0]
    print(f'{name}: {on_grid} layers on the grid, max |grid - windows| = {(got - ref).abs().max():.3e} of {ref.abs().max():.3e}')
    assert got.shape == ref.shape
    assert (got - ref).abs().max() <= 2e-6 * max(1.0, float(ref.abs().max()))
    if on_grid == 0:
        assert torch.equal(got, ref)


@pytest.mark.parametrize('enc', [('PCPatch48', 12, 48, 32), ('Patch32', 8, 32, 16), ('Patch24V2', 8, 24, 16), ('Patch16', 8, 16, 16), ('PCPatch32', 12, 32, 32)])
def test_split_form_between_the_encoder_layers_changes_no_bit(ops, enc):
    """activations kept in split form between valid-conv layers (the producer scales / clamps / splits once, the consumer's staging is a copy):
    the same embeddings bit for bit as with fp32 tensors between the layers -- on the windows and on the grid.  (The persistent grid form of a layer is another
    kernel with another MFMA shape -- equal to tolerance, test_patch_encoder_grid_layer_takes_the_persistent_form -- and is switched off here: this test is
    about the hand-over format.)"""
    import model as rf_model
    name, nf, window, step = enc
    torch.manual_seed(6)
    m = getattr(rf_model, name)(nf, 64).to(DEV).eval()
    gen = torch.Generator().manual_seed(10)
    g = 3 * step + window
    grid = rnd(gen, 2, 1, g, g, g).to(DEV)
    win = rnd(gen, 70, 1, window, window, window).to(DEV)
    res = {}
    ops.USE_CONVV_PG = False
    try:
        with torch.no_grad():
            for flag in (False, True):
                ops.USE_SPLIT_CHAIN = flag
                res[flag] = (m(win), m.forward_grid(grid, window, step))
    finally:
        ops.USE_SPLIT_CHAIN = True
        ops.USE_CONVV_PG = True
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])


def _to_split(ops, x):
    """fp32 [n, c, s, s, s] -> ops.SplitActs as include/rfuse.h defines it: [n][c/4][h | l][voxel][4 halves], h = f16(x/16), l = f16((x/16 - h) * 2^11)"""
    n, c, s = x.shape[0], x.shape[1], x.shape[2]
    t = (x * (1.0 / 16)).clamp(-65504.0, 65504.0)
    h = t.half()
    l = ((t - h.float()) * 2048.0).half()
    hl = torch.stack([h, l], 0).view(2, n, c // 4, 4, s * s * s).permute(1, 2, 0, 4, 3).contiguous()
    return ops.SplitActs(hl.view(torch.float32).view(n, c, s, s, s))


def _from_split(xs):
    n, c, s = xs.shape[0], xs.shape[1], xs.shape[2]
    hl = xs.data.view(torch.float16).view(n, c // 4, 2, s * s * s, 4).float()
    return ((hl[:, :, 0] + hl[:, :, 1] * (1.0 / 2048.0)) * 16.0).permute(0, 1, 3, 2).reshape(n, c, s, s, s)


@pytest.mark.parametrize('spec', [(2, 70, 24), (1, 92, 24), (1, 140, 24), (3, 64, 20), (1, 66, 24)])
def test_conv3d_valid_leaky_split_pg(ops, spec):
    """the persistent two-team form of PCPatch48's 12 -> 24 k3 layer on a whole padded chunk (model/retrieval.py:222 of the reference evaluated fully
    convolutionally; csrc/conv_valid_split_pg.hip), split form in and out: vs float64 torch on the values the split input stands for, and beside the
    tile-per-workgroup kernel (same operands, v_mfma 32x32x16 instead of 16x16x32: equal to within the last bits of an fp32 sum).  Edges that leave ragged
    tiles in every dimension (68 = 4 * 16 + 4 = 17 * 4; 62 ...), several samples (tiles of two samples in one workgroup's walk), cout = 20 (a padded group)."""
    n, s, cout = spec
    cin, k = 12, 3
    gen = torch.Generator().manual_seed(sum(spec) + 6)
    x, w, b = rnd(gen, n, cin, s, s, s), rnd(gen, cout, cin, k, k, k, scale=1 / np.sqrt(cin * k ** 3)), rnd(gen, cout)
    xs = _to_split(ops, x.to(DEV))
    assert ops.conv_valid_split_pg_supported((n, cin, s), cout, k, 1)
    got = ops.conv3d_valid_leaky_split_pg(xs, ops.pack_convv_split_pg_weight(w.to(DEV), s, 1), b.to(DEV), cout, k, 1, 0.2)
    assert isinstance(got, ops.SplitActs) and tuple(got.shape) == (n, cout, s - 2, s - 2, s - 2)
    ref = F.leaky_relu(F.conv3d(_from_split(xs).double().cpu(), w.double(), b.double()), 0.2)
    err = (_from_split(got).double().cpu() - ref).abs()
    print(f'valid split pg {spec}: rms {err.pow(2).mean().sqrt():.3e} max {err.max():.3e}')
    assert err.max() <= 1e-5
    if cout % 4 == 0 and ops.conv_valid_split_supported(x.to(DEV), cout, k, 1):
        v1 = ops.conv3d_valid_leaky_split(xs, ops.pack_convv_split_weight(w.to(DEV), s, 1), b.to(DEV), cout, k, 1, 0.2, out_split=True)
        e1 = (_from_split(v1).double().cpu() - ref).abs()
        assert err.pow(2).mean().sqrt() <= 1.05 * e1.pow(2).mean().sqrt() and err.max() <= 1.25 * e1.max()
        assert (_from_split(v1) - _from_split(got)).abs().max() <= 4e-6


def test_conv3d_valid_leaky_split_pg_refuses_what_it_was_not_built_for(ops):
    """only the instantiation built: 12 -> 17..24 couts in fours, k = 3, stride 1, even edges 64..254; a LeakyReLU slope outside [0, 1] is an error (max form)"""
    for shape, cout, k, stride in [((1, 12, 141), 24, 3, 1), ((1, 12, 60), 24, 3, 1), ((1, 24, 138), 48, 3, 2), ((1, 12, 140), 16, 3, 1), ((1, 12, 140), 24, 5, 1),
                                   ((1, 8, 140), 24, 3, 1)]:
        assert not ops.conv_valid_split_pg_supported(shape, cout, k, stride), (shape, cout, k, stride)
    gen = torch.Generator().manual_seed(3)
    w, b = rnd(gen, 24, 12, 3, 3, 3).to(DEV), rnd(gen, 24).to(DEV)
    with pytest.raises(ValueError):
        ops.pack_convv_split_pg_weight(w, 141, 1)
    xs = _to_split(ops, rnd(gen, 1, 12, 64, 64, 64).to(DEV))
    wp = ops.pack_convv_split_pg_weight(w, 64, 1)
    with pytest.raises(RuntimeError, match='slope'):
        ops.conv3d_valid_leaky_split_pg(xs, wp, b, 24, 3, 1, 1.5)
    with pytest.raises(TypeError):
        ops.conv3d_valid_leaky_split_pg(rnd(gen, 1, 12, 64, 64, 64).to(DEV), wp, b, 24, 3, 1, 0.2)


def test_patch_encoder_grid_layer_takes_the_persistent_form(ops):
    """PCPatch48 on a padded 128^3 chunk (C5): layer 2 of the fully-convolutional evaluation runs through rf_conv3d_valid_leaky_split_pg, and the embeddings
    are those of the route without it to within the encoder's own tolerance (2e-6 on unit-norm-scale outputs)"""
    from model.retrieval import PCPatch48
    torch.manual_seed(5)
    m = PCPatch48(12, 64).to(DEV)
    gen = torch.Generator().manual_seed(9)
    grid = rnd(gen, 1, 1, 144, 144, 144).to(DEV)
    calls = []
    orig = ops.conv3d_valid_leaky_split_pg
    ops.conv3d_valid_leaky_split_pg = lambda *a, **kw: (calls.append(a[0].shape), orig(*a, **kw))[1]
    try:
        with torch.no_grad():
            on = m.forward_grid(grid, 48, 32)
            ops.USE_CONVV_PG = False
            off = m.forward_grid(grid, 48, 32)
    finally:
        ops.USE_CONVV_PG = True
        ops.conv3d_valid_leaky_split_pg = orig
    assert calls == [torch.Size([1, 12, 140, 140, 140])]
    assert tuple(on.shape) == (64, 64, 1, 1, 1)
    assert (on - off).abs().max() <= 2e-6 * max(1.0, float(off.abs().max()))


def test_split_form_tensor_layout(ops):
    """the split form is what include/rfuse.h says: [n][c/4][h | l][voxel][4 halves] with h = f16(x/16), l = f16((x/16 - h) * 2^11) -- read back from the
    VALU first layer and the split layer, against the fp32 outputs of the same calls"""
    gen = torch.Generator().manual_seed(12)
    x, w, b = rnd(gen, 2, 1, 20, 20, 20), rnd(gen, 8, 1, 3, 3, 3, scale=0.3), rnd(gen, 8)
    xd, wt = x.to(DEV), ops.pack_convv_valu_weight(w.to(DEV))
    y = ops.conv3d_valid_leaky_valu(xd, wt, b.to(DEV), 1, 0.2)
    ys = ops.conv3d_valid_leaky_valu(xd, wt, b.to(DEV), 1, 0.2, out_split=True)
    w2, b2 = rnd(gen, 16, 8, 3, 3, 3, scale=0.1), rnd(gen, 16)
    wp = ops.pack_convv_split_weight(w2.to(DEV), 18, 1)
    z = ops.conv3d_valid_leaky_split(y, wp, b2.to(DEV), 16, 3, 1, 0.2)
    zs = ops.conv3d_valid_leaky_split(ys, wp, b2.to(DEV), 16, 3, 1, 0.2, out_split=True)
    for fp32, sp in ((y, ys), (z, zs)):
        n, c, e = fp32.shape[0], fp32.shape[1], fp32.shape[2]
        halves = sp.data.view(torch.float16).reshape(n, c // 4, 2, e, e, e, 4).float().cpu()
        v = (fp32.cpu() / 16).reshape(n, c // 4, 4, e, e, e).permute(0, 1, 3, 4, 5, 2)
        h = v.half().float()
        assert torch.equal(halves[:, :, 0], h)
        assert torch.equal(halves[:, :, 1], ((v - h) * 2048).half().float())
    assert torch.equal(ops.conv3d_valid_leaky_split(ys, wp, b2.to(DEV), 16, 3, 1, 0.2), z)


def test_conv3d_valid_split_saturates_instead_of_overflowing(ops):
    """activations beyond the f16 range (|x|/16 > 65504) are clamped, not turned into inf / NaN"""
    gen = torch.Generator().manual_seed(77)
    x, w, b = rnd(gen, 1, 8, 12, 12, 12), rnd(gen, 16, 8, 3, 3, 3, scale=0.05), rnd(gen, 16)
    x[0, 3, 5, 5, 5] = 3.0e6
    got = ops.conv3d_valid_leaky_split(x.to(DEV), ops.pack_convv_split_weight(w.to(DEV), 12, 1), b.to(DEV), 16, 3, 1, 0.2)
    assert torch.isfinite(got).all()


def test_cpu_tensors_raise(ops):
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.maxpool2(torch.zeros(1, 1, 2, 2, 2))


@pytest.mark.parametrize('case', [(3, 8, 16, 16, 8), (70, 16, 2, 64, 8), (9, 16, 4, 32, 8), (2, 1, 16, 8, 8), (2, 16, 32, 16, 8), (5, 12, 8, 24, 6)])
def test_fused_groupnorm_statistics_match_recomputed(ops, case):
    """conv / max-pool epilogue statistics (rf_conv3d_k3_gn_relu_stats, rf_maxpool3d_2_stats -> rf_gn_from_stats) give the
    same scale/shift as re-reading the tensors (rf_gn_stats), incl. the skip+upsample two-source case."""
    n, cin, edge, cout, groups = case
    gen = torch.Generator().manual_seed(sum(case))
    x = rnd(gen, n, cin, edge, edge, edge).relu_().to(DEV)
    gamma, beta = (1 + 0.2 * rnd(gen, cin)).to(DEV), (0.2 * rnd(gen, cin)).to(DEV)
    w = rnd(gen, cout, cin, 3, 3, 3, scale=1.0 / np.sqrt(27 * cin)).to(DEV)
    aff = ops.gn_affine(x, None, gamma, beta, groups)
    y = ops.conv3d_gn_relu(x, None, aff, ops.pack_conv3_weight(w), cout)
    assert getattr(y, '_rf_stats', None) is not None, 'conv did not emit statistics'
    g2, b2 = (1 + 0.2 * rnd(gen, cout)).to(DEV), (0.2 * rnd(gen, cout)).to(DEV)
    fused = ops.gn_affine(y, None, g2, b2, groups)
    plain = ops.gn_affine(y.clone(), None, g2, b2, groups)            # clone(): no attached statistics -> re-read path
    same_affine(fused, plain, 'scale from fused stats')
    if edge >= 4:
        pooled = ops.maxpool2(y)
        g3, b3 = (1 + 0.2 * rnd(gen, cout)).to(DEV), (0.2 * rnd(gen, cout)).to(DEV)
        f2 = ops.gn_affine(pooled, None, g3, b3, groups)
        p2 = ops.gn_affine(pooled.clone(), None, g3, b3, groups)
        same_affine(f2, p2, 'scale from pooled stats')
        # decoder read: skip = y (full res), upsampled = a conv output at half resolution
        z = ops.conv3d_gn_relu(pooled, None, f2, ops.pack_conv3_weight(rnd(gen, 2 * cout, cout, 3, 3, 3, scale=0.1).to(DEV)), 2 * cout)
        g4, b4 = (1 + 0.2 * rnd(gen, 3 * cout)).to(DEV), (0.2 * rnd(gen, 3 * cout)).to(DEV)
        f3 = ops.gn_affine(y, z, g4, b4, groups)
        p3 = ops.gn_affine(y.clone(), z.clone(), g4, b4, groups)
        same_affine(f3, p3, 'scale, two sources')
    y.mul_(2.0)                                                            # in-place edit invalidates the attached statistics
    stale = ops.gn_affine(y, None, g2, b2, groups)
    same_affine(stale, ops.gn_affine(y.clone(), None, g2, b2, groups), 'stale stats must not be used')


@pytest.mark.parametrize('case', [(65536, 128, 128), (4096, 32, 128), (1000, 56, 50), (300, 7, 513), (16, 128, 128), (70000, 512, 256)])
def test_linear_wgrad_split_k_gemm(ops, case):
    """rf_linear_wgrad: dW = a^T b over K rows (the Linear layers' weight gradient, rfuse/autograd.py) vs float64; ragged M / N / K,
    K smaller than one slice, and run-to-run bit-identical (fixed-order float64 slice sum, no atomics)."""
    k, m, n = case
    gen = torch.Generator().manual_seed(k + m + n)
    a, b = rnd(gen, k, m).to(DEV), rnd(gen, k, n).to(DEV)
    got = ops.linear_wgrad(a, b)
    want = a.double().t() @ b.double()
    scale = float(want.abs().max())
    assert float((got.double() - want).abs().max()) <= 2e-6 * scale * max(1.0, (k / 4096) ** 0.5)
    assert torch.equal(got, ops.linear_wgrad(a, b))


def test_split_forms_fall_back_to_fp32_outside_their_range(ops):
    """The split-operand kernels carry weights x 16 and activations / 16 as f16 pairs and clamp beyond 65504.
    A layer whose parameters could reach that range (|w| > 511, or |gamma| sqrt(group elements) + |beta| > 1.05e6) must run the fp32 kernels and
    give THEIR bits -- not a finite-but-clamped result."""
    from model.unet import SingleConv
    gen = torch.Generator().manual_seed(21)
    x = rnd(gen, 2048, 16, 8, 8, 8).relu_().to(DEV)
    layer = SingleConv(16, 16, num_groups=8).to(DEV)
    with torch.no_grad():
        layer.conv.weight.copy_(rnd(gen, 16, 16, 3, 3, 3, scale=0.05))
        layer.groupnorm.weight.copy_(1.0 + 0.2 * rnd(gen, 16))
        layer.groupnorm.bias.copy_(rnd(gen, 16, scale=0.3))

    def run(arith):
        saved, ops.CONV_ARITH = ops.CONV_ARITH, arith
        try:
            with torch.no_grad():
                return layer(x).clone()
        finally:
            ops.CONV_ARITH = saved

    assert ops.split_range_ok(layer.conv.weight, layer.groupnorm.weight, layer.groupnorm.bias, 2 * 512)
    in_range_split, in_range_fp32 = run('split'), run('fp32')
    assert not torch.equal(in_range_split, in_range_fp32)                  # inside the range the split kernel is the one that runs
    assert (in_range_split - in_range_fp32).abs().max().item() <= 1e-4 * in_range_fp32.abs().max().item()
    for what in ('weight', 'gamma', 'beta'):
        with torch.no_grad():
            saved = {k: v.clone() for k, v in layer.state_dict().items()}
            if what == 'weight':
                layer.conv.weight[3, 5, 1, 1, 1] = 3.0e4                  # x 16 = 4.8e5 > 65504
            elif what == 'gamma':
                layer.groupnorm.weight[2] = 4.0e4                          # |GN output| can reach 4e4 * sqrt(1024) = 1.3e6 > 65504 * 16
            else:
                layer.groupnorm.bias[7] = 2.0e6
        assert not ops.split_range_ok(layer.conv.weight, layer.groupnorm.weight, layer.groupnorm.bias, 2 * 512), what
        got, ref = run('split'), run('fp32')
        assert torch.isfinite(got).all()
        assert torch.equal(got, ref), 'out-of-range %s: the layer did not take the fp32 kernels' % what
        layer.load_state_dict(saved)
    assert torch.equal(run('split'), in_range_split)                       # back in range: the split kernel again


@pytest.mark.parametrize('n,pool', [(2048, 'only'), (2049, 'also'), (2048, None)])
def test_presplit_route_of_a_level0_double_conv(ops, n, pool):
    """Activations in split form end to end: the first conv of a level-0 DoubleConv (1 -> 8 @16^3) writes the second conv's input
    already normalised by ITS GroupNorm and split into f16 pairs (rf_conv3d_cin1_presplit), the second conv (8 -> 16, rf_conv3d_split_pre_k3_relu)
    stages copies.  Same arithmetic as the plain route (conv, gn_affine, cs_split8) -- only the order of the float64 statistics sums differs --
    so the two routes must agree to fp32 round-off, and both with float64 torch."""
    from model.unet import DoubleConv
    gen = torch.Generator().manual_seed(31)
    x = rnd(gen, n, 1, 16, 16, 16).to(DEV)
    blk = DoubleConv(1, 16, encoder=True, num_groups=8).to(DEV)
    with torch.no_grad():
        for name, p in blk.named_parameters():
            if 'groupnorm.weight' in name:
                p.copy_(1.0 + 0.3 * rnd(gen, *p.shape))
            elif 'groupnorm.bias' in name:
                p.copy_(rnd(gen, *p.shape, scale=0.4))
            else:
                p.copy_(rnd(gen, *p.shape, scale=0.2))

    def run(flag):
        saved, ops.USE_PRESPLIT = ops.USE_PRESPLIT, flag
        try:
            with torch.no_grad():
                return blk(x, pool=pool) if pool is not None else blk(x)
        finally:
            ops.USE_PRESPLIT = saved

    with torch.no_grad():
        assert blk._presplit_ok(x)
    fast, plain = run(True), run(False)
    fast = fast if isinstance(fast, tuple) else (fast,)
    plain = plain if isinstance(plain, tuple) else (plain,)
    with torch.no_grad():
        c1, c2 = blk.SingleConv1, blk.SingleConv2
        xd = x.double()
        y = torch.nn.functional.group_norm(xd, 1, c1.groupnorm.weight.double(), c1.groupnorm.bias.double(), 1e-5)
        y = torch.nn.functional.conv3d(y, c1.conv.weight.double(), padding=1).relu()
        y = torch.nn.functional.group_norm(y, 8, c2.groupnorm.weight.double(), c2.groupnorm.bias.double(), 1e-5)
        ref = torch.nn.functional.conv3d(y, c2.conv.weight.double(), padding=1).relu()
        refs = {None: (ref,), 'also': (ref, torch.nn.functional.max_pool3d(ref, 2)), 'only': (None, torch.nn.functional.max_pool3d(ref, 2))}[pool]
    for f, p_, r in zip(fast, plain, refs):
        assert (f is None) == (p_ is None) == (r is None)
        if f is None:
            continue
        scale = r.abs().max().item()
        ef, ep = (f.double() - r).abs().max().item() / scale, (p_.double() - r).abs().max().item() / scale
        assert (f - p_).abs().max().item() <= 2e-6 * scale, 'routes differ by %.2e' % ((f - p_).abs().max().item() / scale)
        assert ef <= max(2e-6, 1.5 * ep), 'pre-split route %.2e from float64, plain route %.2e' % (ef, ep)
        if getattr(f, '_rf_stats', None) is not None:                          # the statistics that ride along feed the next GroupNorm
            sf, sp = f._rf_stats[0].sum(dim=2), p_._rf_stats[0].sum(dim=2)
            assert torch.allclose(sf, sp, rtol=1e-5, atol=1e-5)


def test_prepooled_handover_between_the_first_two_levels(ops):
    """The retrieval backbone's level 0 -> MaxPool3d(2) -> level 1 (reference model/unet.py:230-253) with the pooled tensor handed over pre-split
    (rf_conv3d_split_pre_k3_relu_pool_presplit -> rf_conv3d_split_pre_presplit -> rf_conv3d_split_pre_k3_relu): against the route with an fp32 pooled tensor
    (rf_gn_from_stats + a consumer that normalises and splits) and against float64 torch"""
    from model.unet import UNet3D
    torch.manual_seed(77)
    net = UNet3D(1, 16, f_maps=[16, 32, 64, 128], num_groups=8, num_levels=4, is_segmentation=False, remove_n_final_layers=1).to(DEV).eval()
    with torch.no_grad():
        for m in net.modules():
            if hasattr(m, 'groupnorm'):
                m.groupnorm.weight.add_(0.2 * torch.randn_like(m.groupnorm.weight)); m.groupnorm.bias.add_(0.2 * torch.randn_like(m.groupnorm.bias))
    gen = torch.Generator().manual_seed(6)
    x = rnd(gen, 2100, 1, 16, 16, 16)
    e0, e1 = net.encoders[0], net.encoders[1]
    with torch.no_grad():
        assert e1.basic_module.accepts_prepooled(2100, 16, 8)
        outs = {}
        for flag in (True, False):
            ops.USE_PREPOOL = flag
            _, pooled = e0(x.to(DEV), pool='only', next_block=e1.basic_module)
            assert isinstance(pooled, ops.PreSplit) == flag
            outs[flag] = e1(None, prepooled=pooled, pool='also')
        ops.USE_PREPOOL = True
        x64 = x[:64].double()
        for blk, pool_first in ((e0.basic_module, False), (e1.basic_module, True)):
            if pool_first:
                x64 = F.max_pool3d(x64, 2)
            for sc in (blk.SingleConv1, blk.SingleConv2):
                gn = sc.groupnorm
                g = 1 if gn.num_channels < gn.num_groups else gn.num_groups
                x64 = F.relu(F.conv3d(F.group_norm(x64, g, gn.weight.double().cpu(), gn.bias.double().cpu(), gn.eps), sc.conv.weight.double().cpu(), padding=1))
    for k in (0, 1):
        a_, b_ = outs[True][k], outs[False][k]
        scale = float(b_.abs().max())
        assert (a_ - b_).abs().max().item() <= 2e-6 * max(1.0, scale), 'routes differ by %.2e of %.2f' % ((a_ - b_).abs().max().item(), scale)
    close(outs[True][0][:64], x64.float(), 1e-5, 'level 1 output, pre-split hand-over across the max-pool')
    close(outs[True][1][:64], F.max_pool3d(x64, 2).float(), 1e-5, 'its fused pool')


@pytest.mark.parametrize('shape', [(32, 64, 56, 16, 1030), (32, 48, 48, 16, 1025), (32, 64, 56, 16, 2100), (32, 48, 56, 32, 2070)])     # >= 2048 samples: the persistent producer + the persistent z-column consumer (parity-major hand-over); 1030: the persistent producer, linear order; c1 = 48: six low-res groups
def test_presplit_route_of_a_decoder_conv_pair(ops, shape):
    """StepDownDoubleConv of the retrieval backbone's last decoder (96 -> 56 -> 16 @8^3, reference model/unet.py:149-159): the first conv hands the second
    its input pre-split (rf_conv3d_up_split_presplit -> rf_conv3d_split_pre_k3_relu, the multi-chunk consumer) -- against float64 torch and against the
    plain route (fp32 intermediate + rf_gn_from_stats)"""
    from model.unet import StepDownDoubleConv
    c0, c1, cmid, cout, n = shape
    torch.manual_seed(sum(shape))
    blk = StepDownDoubleConv(c0 + c1, cout, encoder=False, num_groups=8).to(DEV).eval()
    assert blk.SingleConv1.conv.out_channels == cmid
    with torch.no_grad():
        for g in (blk.SingleConv1.groupnorm, blk.SingleConv2.groupnorm):
            g.weight.add_(0.2 * torch.randn_like(g.weight)); g.bias.add_(0.2 * torch.randn_like(g.bias))
    gen = torch.Generator().manual_seed(3)
    skip, low = rnd(gen, n, c0, 8, 8, 8).relu_(), rnd(gen, n, c1, 4, 4, 4).relu_()
    with torch.no_grad():
        from model import unet as unet_mod
        assert unet_mod._decoder_pair_presplit_ok(blk.SingleConv1, blk.SingleConv2, skip.to(DEV), low.to(DEV))
        got = blk(skip.to(DEV), low.to(DEV))
        ops.USE_PRESPLIT = False
        plain = blk(skip.to(DEV), low.to(DEV))
        ops.USE_PRESPLIT = True
        x64 = torch.cat((skip, F.interpolate(low, scale_factor=2, mode='nearest')), 1).double()
        for sc in (blk.SingleConv1, blk.SingleConv2):
            gn = sc.groupnorm
            x64 = F.relu(F.conv3d(F.group_norm(x64, gn.num_groups, gn.weight.double().cpu(), gn.bias.double().cpu(), gn.eps), sc.conv.weight.double().cpu(), padding=1))
    scale = float(x64.abs().max())
    e_pre, e_plain = (got.cpu().double() - x64).abs().max().item(), (plain.cpu().double() - x64).abs().max().item()
    print(f'decoder pair {shape}: |presplit - f64| {e_pre:.3e}, |plain - f64| {e_plain:.3e}, |presplit - plain| {(got - plain).abs().max().item():.3e} of {scale:.2f}')
    assert e_pre <= 1e-5 * max(1.0, scale) and e_pre <= 1.5 * e_plain + 1e-7
    assert (got - plain).abs().max().item() <= 2e-6 * max(1.0, scale)


def test_parity_major_handover_of_the_decoder_pair_equals_the_linear_one(ops):
    """The persistent decoder-form producer (rf_conv3d_up_split_presplit_pm, k_conv3_up_split_pp) writes the voxel slots of its pre-split output in parity-major
    order -- slot ((z & 1) 4 + (y & 1) 2 + (x & 1)) 64 + (z >> 1) 16 + (y >> 1) 4 + (x >> 1) -- and rf_conv3d_split_pre_pm_k3_relu reads that order: the bytes are the
    linear entry point's bytes permuted, the consumer's output is bit-equal, and statistics requested from the producer are the plain kernel's (reference
    model/refinement.py:64-73, model/unet.py:149-159: dec1 of the retrieval backbone)"""
    n, c0, c1, cmid, cout, groups = 2100, 32, 64, 56, 16, 8
    gen = torch.Generator().manual_seed(77)
    skip, low = rnd(gen, n, c0, 8, 8, 8).relu_().to(DEV), rnd(gen, n, c1, 4, 4, 4).relu_().to(DEV)
    w1, w2 = (rnd(gen, cmid, c0 + c1, 3, 3, 3) * 0.05).to(DEV), (rnd(gen, cout, cmid, 3, 3, 3) * 0.05).to(DEV)
    g1w, g1b = (1 + 0.2 * rnd(gen, c0 + c1)).to(DEV), (0.2 * rnd(gen, c0 + c1)).to(DEV)
    g2w, g2b = (1 + 0.2 * rnd(gen, cmid)).to(DEV), (0.2 * rnd(gen, cmid)).to(DEV)
    aff = ops.gn_affine(skip, low, g1w, g1b, groups, 1e-5)
    wp1, wp2 = ops.pack_conv3_up_split_weight(w1, c0), ops.pack_conv3_split_weight(w2)
    assert ops.conv_up_split_presplit_pm_supported(skip, low, cmid, groups, cout)
    lin = ops.conv3d_up_split_presplit(skip, low, aff, wp1, cmid, g2w, g2b, groups, 1e-5)
    pm = ops.conv3d_up_split_presplit(skip, low, aff, wp1, cmid, g2w, g2b, groups, 1e-5, parity_major=True)
    z, y, x = torch.meshgrid(torch.arange(8), torch.arange(8), torch.arange(8), indexing='ij')
    perm = (((z & 1) * 4 + (y & 1) * 2 + (x & 1)) * 64 + (z >> 1) * 16 + (y >> 1) * 4 + (x >> 1)).reshape(-1).to(DEV)
    lin5, pm5 = lin.view(n, cmid // 8, 2, 512, 16), pm.view(n, cmid // 8, 2, 512, 16)
    assert torch.equal(pm5[:, :, :, perm], lin5), 'parity-major bytes are not the linear bytes permuted'
    out_lin = ops.conv3d_split_pre_relu(lin, cmid, n, 8, wp2, cout)
    out_pm = ops.conv3d_split_pre_relu(pm, cmid, n, 8, wp2, cout, parity_major=True)
    assert torch.equal(out_lin, out_pm)
    # the producer's optional statistics output (per (sample, cout): sum and sum of squares of the ReLU'd conv output) against the plain kernel's
    from rfuse import _lib
    lib = _lib.load()
    st_pm = torch.empty((n, cmid, 2), dtype=torch.float64, device=DEV)
    scratch = torch.empty_like(pm)
    p = lambda t_: t_.data_ptr() if t_ is not None else None
    _lib.check(lib.rf_conv3d_up_split_presplit_pm(p(skip), c0, p(low), c1, n, 8, p(aff), p(wp1), cmid, p(g2w), p(g2b), groups, 1e-5, p(scratch), p(st_pm),
                                                  torch.cuda.current_stream().cuda_stream), 'rf_conv3d_up_split_presplit_pm')
    assert torch.equal(scratch, pm)
    plain = ops.conv3d_up_split_gn_relu(skip, low, aff, wp1, cmid)
    ref_sum, ref_sq = plain.double().sum(dim=(2, 3, 4)), (plain.double() ** 2).sum(dim=(2, 3, 4))
    assert (st_pm[..., 0] - ref_sum).abs().max().item() <= 1e-6 * ref_sum.abs().max().item()
    assert (st_pm[..., 1] - ref_sq).abs().max().item() <= 1e-6 * ref_sq.abs().max().item()
    # against float64 on a few samples
    x64 = torch.cat((skip[:6].cpu(), F.interpolate(low[:6].cpu(), scale_factor=2, mode='nearest')), 1).double()
    for gw, gb, w in ((g1w, g1b, w1), (g2w, g2b, w2)):
        x64 = F.relu(F.conv3d(F.group_norm(x64, groups, gw.double().cpu(), gb.double().cpu(), 1e-5), w.double().cpu(), padding=1))
    close(out_pm[:6], x64.float(), 1e-5, 'decoder pair through the parity-major hand-over')


@pytest.mark.parametrize('n', [1030, 2070])
def test_presplit_route_of_an_encoder_pair_on_whole_samples(ops, n):
    """DoubleConv of an encoder level on whole 8^3 samples (the retrieval backbone's 16 -> 16 -> 32 @8^3, fused pool): the split box kernel hands the second
    conv its input pre-split (rf_conv3d_split_presplit) -- bit-equal to the plain route, close to float64"""
    from model.unet import DoubleConv
    torch.manual_seed(21)
    blk = DoubleConv(16, 32, encoder=True, num_groups=8).to(DEV).eval()
    with torch.no_grad():
        for g in (blk.SingleConv1.groupnorm, blk.SingleConv2.groupnorm):
            g.weight.add_(0.2 * torch.randn_like(g.weight)); g.bias.add_(0.2 * torch.randn_like(g.bias))
    gen = torch.Generator().manual_seed(4)
    x = rnd(gen, n, 16, 8, 8, 8).relu_()                            # 2070: the producer on the persistent z-column form (k_conv3_split_zcm, pre-split epilogue)
    with torch.no_grad():
        assert blk._box_pair_presplit_ok(x.to(DEV))
        outs = {}
        for flag in (True, False):
            ops.USE_PRESPLIT = flag
            outs[flag] = [blk(x.to(DEV)), blk(x.to(DEV), pool='also'), blk(x.to(DEV), pool='only')]
        ops.USE_PRESPLIT = True
        x64 = x.double()
        for sc in (blk.SingleConv1, blk.SingleConv2):
            gn = sc.groupnorm
            x64 = F.relu(F.conv3d(F.group_norm(x64, gn.num_groups, gn.weight.double().cpu(), gn.bias.double().cpu(), gn.eps), sc.conv.weight.double().cpu(), padding=1))
    close(outs[True][0], x64.float(), 1e-5, 'encoder pair, pre-split route')
    assert torch.equal(outs[True][0], outs[False][0])
    assert torch.equal(outs[True][1][0], outs[False][1][0]) and torch.equal(outs[True][1][1], outs[False][1][1])
    assert outs[True][2][0] is None and torch.equal(outs[True][2][1], outs[False][2][1])
    assert torch.equal(outs[True][1][1], F.max_pool3d(outs[True][0], 2))


@pytest.mark.parametrize('nf,batch', [(16, 3), (12, 3), (16, 5)])
def test_final_decoder_head_in_the_conv_epilogue(ops, nf, batch):
    """Superresolution08FinalDecoder (reference model/refinement.py:48-61): the 1x1x1 conv + tanh (+ network_pred_to_df) run in the epilogue of the up
    stage's second conv (rf_conv3d_split_k3_gn_relu_pointwise_tanh) -- the same bits as the conv followed by rf_conv1x1_tanh, and the float64 value"""
    import model as rf_model
    torch.manual_seed(nf)
    dec = rf_model.Superresolution08FinalDecoder(nf, 'gcr').to(DEV).eval()
    gen = torch.Generator().manual_seed(8)
    x = rnd(gen, batch, nf, 32, 32, 32).relu_()                     # batch 5: 2560 boxes, the persistent z-column form takes both routes
    with torch.no_grad():
        dc = dec.network[0].basic_module
        y1 = dc.SingleConv1(None, x.to(DEV))
        assert ops.conv_split_pointwise_supported(y1, nf)
        fused, fused_df = dec(x.to(DEV)), dec.forward_df(x.to(DEV), 0.375)
        y2 = dc.SingleConv2(y1)
        plain = ops.conv1x1_tanh(y2, dec.network[1].weight, dec.network[1].bias)
        plain_df = ops.conv1x1_tanh(y2, dec.network[1].weight, dec.network[1].bias, post_add=1.0, post_mul=0.375 / 2)
        x64 = F.interpolate(x.double(), scale_factor=2, mode='nearest')
        for sc in (dc.SingleConv1, dc.SingleConv2):
            gn = sc.groupnorm
            x64 = F.relu(F.conv3d(F.group_norm(x64, gn.num_groups, gn.weight.double().cpu(), gn.bias.double().cpu(), gn.eps), sc.conv.weight.double().cpu(), padding=1))
        ref = torch.tanh(F.conv3d(x64, dec.network[1].weight.double().cpu(), dec.network[1].bias.double().cpu()))
    assert torch.equal(fused, plain) and torch.equal(fused_df, plain_df)
    close(fused, ref.float(), 1e-5, 'final decoder with the fused head')
    # the channel-interleaved hand-over between the two convs (taken from 2048 boxes on) changes the layout of the intermediate, not a bit of the result
    took_ch8 = ops.conv_up_split_ch8_supported(x.to(DEV), nf, nf)
    assert took_ch8 == (batch >= 4 and nf % 8 == 0)
    saved, ops.USE_CH8 = ops.USE_CH8, False
    try:
        with torch.no_grad():
            assert torch.equal(dec(x.to(DEV)), fused) and torch.equal(dec.forward_df(x.to(DEV), 0.375), fused_df)
    finally:
        ops.USE_CH8 = saved


def test_in_kernel_gumbel_sampler(ops):
    """rf_attn_weights_sampled draws the Gumbel noise of gumbel_softmax(hard=True) (reference model/attention.py:100-103) inside the kernel:
    (a) the noise it reports, fed to the explicit-noise entry point, reproduces its weights bit for bit; (b) the noise is Gumbel(0, 1) --
    moments and quantiles against torch's own -log(Exponential(1)) sampler; (c) consecutive calls draw different noise, equal seeds equal noise."""
    gen = torch.Generator().manual_seed(41)
    rows, K = 1 << 18, 8
    xf, pf = rnd(gen, rows, 32).to(DEV), rnd(gen, rows * K, 32).to(DEV)
    st = ops.gumbel_rng_state(DEV, seed=1234)
    w, sw, nz = ops.attn_weights_sampled(xf, pf, K, 25.0, st, want_noise=True)
    w2, sw2 = ops.attn_weights(xf, pf, nz, K, ops.ATTN_GUMBEL_HARD, 25.0)
    assert torch.equal(w, w2) and torch.equal(sw, sw2)
    assert st.cpu().tolist() == [1234, 1, 0]
    _, _, nz_b = ops.attn_weights_sampled(xf, pf, K, 25.0, st, want_noise=True)
    assert not torch.equal(nz, nz_b) and st.cpu().tolist()[1] == 2
    _, _, nz_c = ops.attn_weights_sampled(xf, pf, K, 25.0, ops.gumbel_rng_state(DEV, seed=1234), want_noise=True)
    assert torch.equal(nz, nz_c)
    ref = -torch.empty(rows * K, device=DEV).exponential_().log()
    a, b = nz.flatten().double(), ref.double()
    assert torch.isfinite(a).all()
    assert abs(a.mean().item() - 0.5772156649) < 5e-3 and abs(a.var().item() - 1.6449340668) < 2e-2
    qs = torch.tensor([0.001, 0.01, 0.1, 0.25, 0.5, 0.75, 0.9, 0.99, 0.999], device=DEV, dtype=torch.float64)
    qa, qb = torch.quantile(a[:1 << 20], qs), torch.quantile(b[:1 << 20], qs)
    exact = -torch.log(-torch.log(qs))
    assert (qa - exact).abs().max().item() < 0.03 and (qb - exact).abs().max().item() < 0.03
    # independence across the K draws of a row and across rows: correlations of a white sample
    m = nz.double() - nz.double().mean()
    c01 = (m[:, 0] * m[:, 1]).mean().item() / m.var().item()
    crow = (m[:-1, 0] * m[1:, 0]).mean().item() / m.var().item()
    assert abs(c01) < 0.01 and abs(crow) < 0.01
