"""Dev probe (GPU box): does the fp32 MFMA accumulate with round-to-nearest or with truncation?  All-positive operands make a
truncating accumulator lose a systematic ~n*eps/2, a rounding one a zero-mean ~sqrt(n)*eps."""
import sys
from pathlib import Path
import torch
REPO = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(REPO), str(REPO / 'retrieval-fuse_amd')]
from rfuse import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
for nin in (64, 432, 1024, 4096):
    x = torch.rand(256, nin, generator=g) + 1.0
    w = torch.rand(64, nin, generator=g) + 1.0
    y64 = x.double() @ w.double().T
    y = ops.linear(x.to(dev), ops.pack_linear_weight(w.to(dev)), None, 64).cpu().double()
    rel = (y - y64) / y64
    # fp32 sequential round-to-nearest emulation on the CPU (k order 0..n-1) and torch's own fp32 matmul
    seq = torch.zeros(256, 64)
    for k in range(nin):
        seq = seq + x[:, k:k + 1] * w[:, k][None, :]
    rel_seq = (seq.double() - y64) / y64
    rel_mm = ((x @ w.T).double() - y64) / y64
    print('K=%5d  mfma: mean %+.3e rms %.3e   | cpu sequential RN (unfused mul, add): mean %+.3e rms %.3e | torch matmul: mean %+.3e rms %.3e'
          % (nin, rel.mean(), rel.pow(2).mean().sqrt(), rel_seq.mean(), rel_seq.pow(2).mean().sqrt(), rel_mm.mean(), rel_mm.pow(2).mean().sqrt()))
