"""Dev tool (GPU box): where does the HIP path's distance from the float64 truth come from?

For every GroupNorm+conv+ReLU layer of the U-Net backbone, the retrieval backbone and the decoder of a golden fixture, feed the
SAME fp32 input (the oracle's activation) to torch-CPU fp32 (= what the reference runs) and to the HIP kernel, and compare both
with the float64 evaluation of that one layer: per-layer error GENERATION, isolated from propagation.

    python tools/error_budget.py net_C4
"""
import contextlib, io, sys
from pathlib import Path
import numpy as np, torch, torch.nn.functional as F
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO), str(REPO / 'retrieval-fuse_amd'), str(REPO / 'tests')]
import helpers
from oracle import refpath
from rfuse import configs as rf_configs
import model

name = sys.argv[1] if len(sys.argv) > 1 else 'net_C1'
fix = helpers.load_fixture(name)
cfg0 = rf_configs.get_config(str(fix['cfg_name']))
with contextlib.redirect_stdout(io.StringIO()):
    mods = {'unet_backbone': model.get_unet_backbone(cfg0), 'decoder': model.get_decoder(cfg0),
            'retrieval_backbone': model.get_retrieval_backbone(cfg0), 'patched_attention_block': model.get_attention_block(cfg0)}
shapes = {k: {n: tuple(v.shape) for n, v in m.state_dict().items()} for k, m in mods.items()}
cfg, x_in, retr, sds = helpers.fixture_problem(fix, shapes)
dev = torch.device('cuda:0')
for k, m in mods.items():
    m.load_state_dict(sds[k]); m.to(dev).eval()
torch.set_num_threads(32)

records = []
orig = refpath.single_conv_gcr
cur = [None]
def rec(x, sd, prefix, g):
    y = orig(x, sd, prefix, g)
    records.append((cur[0], prefix, x, y))
    return y
refpath.single_conv_gcr = rec
K, B = cfg['K'], x_in.shape[0]
with torch.no_grad():
    cur[0] = 'unet_backbone'
    xb = refpath.unet_backbone(torch.from_numpy(x_in), sds['unet_backbone'], cfg)
    cur[0] = 'retrieval_backbone'
    rt = torch.from_numpy(retr)[:, :K].reshape(B * K, 1, 64, 64, 64)
    feat = refpath.retrieval_backbone(refpath.unfold3d(rt, 16), sds['retrieval_backbone'], cfg)
    x_retr = refpath.fold3d(feat, 4, 8, cfg['nf'])
    noise = torch.from_numpy(fix['gumbel_noise']) if 'gumbel_noise' in fix else None
    xa = refpath.patched_attention_block(xb, x_retr, sds['patched_attention_block'], cfg, noise)
    cur[0] = 'decoder'
    refpath.final_decoder(xa, sds['decoder'], cfg)
G = cfg['nf'] // 2
print('%-64s %-14s  torch-vs-f64 (max, rms)    hip-vs-f64 (max, rms)   ratio(rms)  |y|max' % ('layer', 'shape'))
with torch.no_grad():
    for mname, prefix, xin, yref in records:
        sc = dict(mods[mname].named_modules())[prefix]
        got = sc(xin.to(dev).contiguous()).cpu().double()
        sd = sds[mname]
        g = 1 if xin.shape[1] < G else G
        xn64 = F.group_norm(xin.double(), g, sd[prefix + '.groupnorm.weight'].double(), sd[prefix + '.groupnorm.bias'].double(), 1e-5)
        y64 = F.relu(F.conv3d(xn64, sd[prefix + '.conv.weight'].double(), None, padding=1))
        et, eg = (yref.double() - y64).abs(), (got - y64).abs()
        rt_, rg_ = et.pow(2).mean().sqrt().item(), eg.pow(2).mean().sqrt().item()
        # torch with the normalisation done in float64 and only the conv in fp32: separates GroupNorm-apply error from conv error
        yc = F.relu(F.conv3d(xn64.float(), sd[prefix + '.conv.weight'], None, padding=1)).double()
        rc_ = (yc - y64).pow(2).mean().sqrt().item()
        print(f'{mname[:9]}.{prefix:54s} {xin.shape[1]:3d}->{yref.shape[1]:3d}@{xin.shape[2]:<3d}n{xin.shape[0]:<4d} '
              f'{et.max().item():.2e} {rt_:.2e}      {eg.max().item():.2e} {rg_:.2e}     {rg_ / max(rt_, 1e-30):5.2f}   {yref.abs().max():.2f}   conv-only-fp32 rms {rc_:.2e}')
