"""Dev tool (GPU box): which stage of the HIP path contributes how much to the distance of ``df`` from the float64 truth?
Replaces stage outputs of the HIP path by the float64 truth (rounded to fp32) one at a time.

    python tools/error_sources.py net_C4
"""
import contextlib, io, sys
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO), str(REPO / 'retrieval-fuse_amd'), str(REPO / 'tests')]
import helpers
from oracle import refpath
from rfuse import configs as rf_configs
import model
from model.attention import Unfold3D

name = sys.argv[1] if len(sys.argv) > 1 else 'net_C4'
fix, truth = helpers.load_truth(name)
cfg0 = rf_configs.get_config(str(fix['cfg_name']))
with contextlib.redirect_stdout(io.StringIO()):
    mods = {'unet_backbone': model.get_unet_backbone(cfg0), 'decoder': model.get_decoder(cfg0),
            'retrieval_backbone': model.get_retrieval_backbone(cfg0), 'patched_attention_block': model.get_attention_block(cfg0)}
shapes = {k: {n: tuple(v.shape) for n, v in m.state_dict().items()} for k, m in mods.items()}
cfg, x_in, retr, sds = helpers.fixture_problem(fix, shapes)
dev = torch.device('cuda:0')
for k, m in mods.items():
    m.load_state_dict(sds[k]); m.to(dev).eval()
torch.set_num_threads(64)
K, B = cfg['K'], x_in.shape[0]
trunc_t = float(fix['target_trunc'])
sds64 = {m: {k: v.double() for k, v in sd.items()} for m, sd in sds.items()}
noise = torch.from_numpy(fix['gumbel_noise']) if 'gumbel_noise' in fix else None
with torch.no_grad():
    xb64 = refpath.unet_backbone(torch.from_numpy(x_in).double(), sds64['unet_backbone'], cfg)
    rt = torch.from_numpy(retr)[:, :K].reshape(B * K, 1, 64, 64, 64)
    feat64 = refpath.retrieval_backbone(refpath.unfold3d(rt.double(), 16), sds64['retrieval_backbone'], cfg)      # patch-major
    xb32 = refpath.unet_backbone(torch.from_numpy(x_in), sds['unet_backbone'], cfg)
    feat32 = refpath.retrieval_backbone(refpath.unfold3d(rt, 16), sds['retrieval_backbone'], cfg)
    xb_g = mods['unet_backbone'](torch.from_numpy(x_in).to(dev))
    feat_g = mods['retrieval_backbone'](Unfold3D(16, 1)(rt.to(dev)))
    ng = noise.to(dev) if noise is not None else None

    def finish(xb, feat):
        x = mods['patched_attention_block'].forward_patch_major(xb.contiguous(), feat.contiguous(), 8, ng)
        return mods['decoder'].forward_df(x, trunc_t).cpu().numpy()

    def finish_cpu(xb, feat):
        x_retr = refpath.fold3d(feat, 4, 8, cfg['nf'])
        x = refpath.patched_attention_block(xb, x_retr, sds['patched_attention_block'], cfg, noise)
        return refpath.network_pred_to_df(refpath.final_decoder(x, sds['decoder'], cfg), trunc_t).numpy()

    rows = [
        ('reference (torch-CPU fp32, all stages)', fix['df']),
        ('HIP all stages', finish(xb_g, feat_g)),
        ('HIP attention+decoder on TRUTH x_back, TRUTH feats', finish(xb64.float().to(dev), feat64.float().to(dev))),
        ('HIP, x_back := truth', finish(xb64.float().to(dev), feat_g)),
        ('HIP, feats := truth', finish(xb_g, feat64.float().to(dev))),
        ('HIP attention+decoder on torch-fp32 x_back, feats', finish(xb32.to(dev), feat32.to(dev))),
        ('torch attention+decoder on TRUTH x_back, TRUTH feats', finish_cpu(xb64.float(), feat64.float())),
        ('torch attention+decoder on HIP x_back, HIP feats', finish_cpu(xb_g.cpu(), feat_g.cpu())),
    ]
print('stage errors vs f64:  x_back torch %.2e hip %.2e (rms %.2e / %.2e) | feats torch %.2e hip %.2e (rms %.2e / %.2e)' % (
    (xb32.double() - xb64).abs().max(), (xb_g.cpu().double() - xb64).abs().max(), (xb32.double() - xb64).pow(2).mean().sqrt(), (xb_g.cpu().double() - xb64).pow(2).mean().sqrt(),
    (feat32.double() - feat64).abs().max(), (feat_g.cpu().double() - feat64).abs().max(), (feat32.double() - feat64).pow(2).mean().sqrt(), (feat_g.cpu().double() - feat64).pow(2).mean().sqrt()))
for label, df in rows:
    p = helpers.error_profile(df, truth['df_f64'])
    print('%-58s max %.2e rms %.2e p99.9 %.2e frac>1e-4 %.5f' % (label, p['max'], p['rms'], p['p99.9'], p['frac>1e-4']))
