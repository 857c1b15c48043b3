"""Dev tool (GPU box): N4 measurement -- one training step of the reference's phase-3 graph (trainer/train_refinement.py:108-116
forward_full -> L1 on df -> backward through all four networks, Adam step) through the drop-in modules in grad mode, timed with HIP
events, next to the same step of the oracle on the host CPU (torch fp32 autograd).  Prints one JSON line.

    python tools/train_bench.py [config] [batch] [steps]
"""
import contextlib
import io
import json
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO), str(REPO / 'retrieval-fuse_amd'), str(REPO / 'tests')]
import torch

import model
from model.attention import Fold3D, Unfold3D
from rfuse import _lib, configs as rf_configs

name = sys.argv[1] if len(sys.argv) > 1 else 'C3'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
cfg = rf_configs.get_config(name)
_, trunc_t = rf_configs.truncations(cfg)
K, S = cfg['K'], cfg['dataset_train']['input_chunk_size'] if 'dataset_train' in cfg and 'input_chunk_size' in cfg['dataset_train'] else 8
dev = torch.device('cuda:0')
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    mods = {'unet_backbone': model.get_unet_backbone(cfg), 'decoder': model.get_decoder(cfg),
            'retrieval_backbone': model.get_retrieval_backbone(cfg), 'patched_attention_block': model.get_attention_block(cfg)}
for m in mods.values():
    m.to(dev).train()
params = [p for m in mods.values() for p in m.parameters()]
opt = torch.optim.Adam(params, lr=1e-4)
gen = torch.Generator().manual_seed(3)
x_in = torch.randn(B, 1, S, S, S, generator=gen).to(dev)
retr = torch.randn(B, K, 64, 64, 64, generator=gen).to(dev)
target = (torch.rand(B, 1, 64, 64, 64, generator=gen) * trunc_t).to(dev)
unfold, fold = Unfold3D(16, 1), Fold3D(4, 8, cfg['nf'])


def step():
    opt.zero_grad(set_to_none=True)
    x_back = mods['unet_backbone'](x_in)
    feats = mods['retrieval_backbone'](unfold(retr.reshape(B * K, 1, 64, 64, 64)))
    x_attn = mods['patched_attention_block'](x_back, fold(feats))
    df = (mods['decoder'](x_attn) + 1) * trunc_t / 2
    loss = (df - target).abs().mean()
    loss.backward()
    opt.step()
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(steps):
    loss = step()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / steps

# per-entry-point table of one step (HIP events around every C-ABI launch)
lib = _lib.load()
records = []
lib.start_profile(records)
step()
torch.cuda.synchronize()
lib.stop_profile()
agg = {}
for entry, _, a0, a1, _nulls in records:
    agg[entry] = agg.get(entry, 0.0) + a0.elapsed_time(a1)
top = sorted(agg.items(), key=lambda kv: -kv[1])[:8]

# the oracle's step on the host CPU (fp32 autograd), bounded sample
from oracle import refpath
sds = {k: {n: v.detach().cpu().clone().requires_grad_(True) for n, v in m.state_dict().items() if v.dtype.is_floating_point} for k, m in mods.items()}
torch.set_num_threads(min(32, torch.get_num_threads()))
bc = min(B, 2)
t0 = time.time()
noise_c = (-torch.empty(bc * 4096, K).exponential_().log()) if cfg['attn_retrieval_mode'] else None
dfo = refpath.forward_full(sds, cfg, x_in[:bc].cpu(), retr[:bc].cpu(), trunc_t, noise_c)
(dfo - target[:bc].cpu()).abs().mean().backward()
cpu_s = time.time() - t0
print(json.dumps({'metric': 'training steps/s (forward_full + backward + Adam, phase 3: all four networks trainable)', 'config': name, 'chunks_per_step': B,
                  'ms_per_step': ms, 'chunks_per_s': B / ms * 1e3, 'loss': float(loss),
                  'hip_entry_points_ms': {k: round(v, 3) for k, v in top}, 'hip_ms_total': round(sum(agg.values()), 3),
                  'cpu_oracle': {'chunks_per_s': bc / cpu_s, 'threads': torch.get_num_threads(), 'sample': '%d chunks, fp32 torch autograd of oracle/refpath.py' % bc}}))
