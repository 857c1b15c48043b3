"""Dev tool (GPU box): per-layer error of the HIP U-Net path against the oracle, feeding each layer the ORACLE's input
(isolates per-layer error) and also reporting the chained error."""
import contextlib, io, sys
from pathlib import Path
import numpy as np, torch, torch.nn.functional as F
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO), str(REPO / 'retrieval-fuse_amd'), str(REPO / 'tests')]
import helpers
from oracle import refpath
from rfuse import configs as rf_configs, ops
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

# hook the oracle's single_conv to record (input parts, output) per layer
records = []
orig = refpath.single_conv_gcr
def rec(x, sd, prefix, g):
    y = orig(x, sd, prefix, g)
    records.append((prefix, x, y))
    return y
refpath.single_conv_gcr = rec
with torch.no_grad():
    xb_ref = refpath.unet_backbone(torch.from_numpy(x_in), sds['unet_backbone'], cfg)
G = cfg['nf'] // 2
mod = mods['unet_backbone']
named = dict(mod.named_modules())
print('layer, cin->cout @edge, isolated err (oracle input), |ref|max')
with torch.no_grad():
    for prefix, xin, yref in records:
        sc = named[prefix]
        got = sc(xin.to(dev).contiguous())          # concat/upsample already materialised by the oracle: single source
        err = (got.cpu() - yref).abs().max().item()
        # fp64 truth for this layer from the oracle's input
        g = 1 if xin.shape[1] < G else G
        sd = sds['unet_backbone']
        y64 = F.relu(F.conv3d(F.group_norm(xin.double(), g, sd[prefix + '.groupnorm.weight'].double(), sd[prefix + '.groupnorm.bias'].double(), 1e-5),
                              sd[prefix + '.conv.weight'].double(), None, padding=1))
        e_ref64 = (yref.double() - y64).abs().max().item()
        e_got64 = (got.cpu().double() - y64).abs().max().item()
        print(f'{prefix:55s} {xin.shape[1]:4d}->{yref.shape[1]:4d} @{xin.shape[2]:3d}  hip-vs-torch {err:.2e}  torch-vs-f64 {e_ref64:.2e}  hip-vs-f64 {e_got64:.2e}  |ref| {yref.abs().max():.2f}')
    xb = mod(torch.from_numpy(x_in).to(dev))
print('chained x_back err', (xb.cpu() - xb_ref).abs().max().item())
