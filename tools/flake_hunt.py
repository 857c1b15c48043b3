"""Dev tool (GPU box): hunt the rare failure of tests/test_autograd_gpu.py::test_attention_feature_encoder_gradients in fresh processes."""
import contextlib, io, sys
from pathlib import Path
import numpy as np, torch
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO), str(REPO / 'retrieval-fuse_amd'), str(REPO / 'tests')]
from oracle import refpath
import testkit
poison = len(sys.argv) > 1 and sys.argv[1] == 'poison'
gpu = torch.device('cuda:0')
if poison:
    blocks = [torch.full((64 * 1024 * 1024,), float('nan'), dtype=torch.float32, device=gpu) for _ in range(6)]
    blocks += [torch.full((256 * 1024 - 64,), float('nan'), dtype=torch.float32, device=gpu) for _ in range(256)]
    del blocks
    tk = testkit.load(); tk.rft_poison_lds(None); tk.rft_poison_vgprs(None); torch.cuda.synchronize()
from model.attention import AttentionFeatureEncoder
gen = torch.Generator().manual_seed(3)
with contextlib.redirect_stdout(io.StringIO()):
    enc = AttentionFeatureEncoder(16, 32, 2).to(gpu)
x = torch.randn(700, 128, generator=gen); r = torch.randn(700, 32, generator=gen)
sd = {'t.' + k: v.detach().cpu().double().requires_grad_(True) for k, v in enc.state_dict().items()}
xo = x.double().requires_grad_(True)
yo = refpath.attention_feature_encoder(xo, sd, 't'); (yo * r.double()).sum().backward()
bad_any = False
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    for p in enc.parameters(): p.grad = None
    xg = x.to(gpu).requires_grad_(True)
    y = enc(xg); (y * r.to(gpu)).sum().backward()
    d = (xg.grad.cpu().double() - xo.grad).abs()
    rel = d.max().item() / xo.grad.abs().max().item()
    msgs = []
    if rel > 1e-4:
        idx = (d > 1e-4 * xo.grad.abs().max()).nonzero()
        msgs.append('dx rel %.3e: %d bad elements, rows %s cols %s' % (rel, len(idx), sorted(set(idx[:, 0].tolist()))[:20], sorted(set(idx[:, 1].tolist()))[:20]))
    for name, p in enc.named_parameters():
        g = sd['t.' + name].grad
        rp = (p.grad.cpu().double() - g).abs().max().item() / g.abs().max().item()
        if rp > 1e-4: msgs.append('%s rel %.3e' % (name, rp))
    ry = (y.detach().cpu().double() - yo.detach()).abs().max().item() / yo.abs().max().item()
    if ry > 1e-5: msgs.append('y rel %.3e' % ry)
    if msgs:
        bad_any = True
        print('iteration', it, '|', ' ; '.join(msgs), flush=True)
print('FAIL' if bad_any else 'ok', flush=True)
