"""Training losses the reference's trainers import from ``model.loss`` (trainer/train_refinement.py:13, trainer/train_retrieval.py:9).

Not on the refinement-inference hot path (losses are out of scope) -- they are here so that the drop-in ``model`` package
resolves every name the reference's callers import.  Plain torch on whatever device the inputs live on: the reference's
``mask.cuda(zis.device)`` (model/loss.py:57,62) pins the mask to a CUDA device and fails on CPU tensors; here the mask is built on the
inputs' device.  Values equal the reference's (tests/test_loss_cpu.py against tests/golden/loss.npz, produced by the reference's module).
"""
import torch
import torch.nn.functional as F


def _negatives_mask(batch_size, device):
    """[2B, 2B] bool: True where column j is neither row i itself nor its positive partner (i + B) mod 2B (reference model/loss.py:25-32)."""
    i = torch.arange(2 * batch_size, device=device)
    delta = (i[None, :] - i[:, None]) % (2 * batch_size)
    return (delta != 0) & (delta != batch_size)


class NTXentLoss(torch.nn.Module):
    """Normalised-temperature cross entropy over a batch of positive pairs (zis[i], zjs[i]) -- reference model/loss.py:5-69.

    ``forward(zis, zjs, iou_matrix=None)``: similarities of the 2B stacked representations ``[zjs; zis]``; per row the positive logit is
    the partner's similarity, the negatives are the other 2B - 2 in column order; cross entropy against the positive, summed and divided
    by 2B.  With ``iou_matrix`` ([2B, 2B]) the negatives' temperature is raised towards 1 where the two patches overlap:
    ``t + (1 - t) * sigmoid(iou * sig_scale + sig_shift)``.
    """

    def __init__(self, temperature, use_cosine_similarity, sig_scale=80, sig_shift=-65):
        super().__init__()
        self.temperature = temperature
        self.use_cosine_similarity = bool(use_cosine_similarity)
        self.sig_scale = sig_scale
        self.sig_shift = sig_shift

    def similarity(self, reps):
        if self.use_cosine_similarity:
            return F.cosine_similarity(reps.unsqueeze(1), reps.unsqueeze(0), dim=-1)
        return torch.tensordot(reps.unsqueeze(1), reps.T.unsqueeze(0), dims=2)

    def forward(self, zis, zjs, iou_matrix=None):
        b = zis.shape[0]
        reps = torch.cat([zjs, zis], dim=0)
        sim = self.similarity(reps)
        rows = torch.arange(2 * b, device=sim.device)
        positives = sim[rows, (rows + b) % (2 * b)].unsqueeze(1)
        mask = _negatives_mask(b, sim.device)
        negatives = sim[mask].view(2 * b, 2 * b - 2)
        if iou_matrix is None:
            logits = torch.cat([positives, negatives], dim=1) / self.temperature
        else:
            overlap = torch.sigmoid(iou_matrix[mask].view(2 * b, 2 * b - 2) * self.sig_scale + self.sig_shift)
            logits = torch.cat([positives / self.temperature, negatives / (self.temperature + (1 - self.temperature) * overlap)], dim=1)
        target = torch.zeros(2 * b, dtype=torch.long, device=sim.device)
        return F.cross_entropy(logits, target, reduction='sum') / (2 * b)


def patch_style_loss(zis, zjs):
    """MSE between the Gram matrices of the two feature sets, the second one detached (reference model/loss.py:72-75)."""
    return F.mse_loss(zis @ zis.t(), (zjs @ zjs.t()).detach())


def get_cosine_similarity(pred_norms, target_norms):
    """Mean cosine similarity between predicted and target normals [B, 3, D, H, W] over the voxels where both are non-zero
    (reference model/loss.py:78-85)."""
    p = pred_norms.permute(0, 2, 3, 4, 1).reshape(-1, 3)
    t = target_norms.permute(0, 2, 3, 4, 1).reshape(-1, 3)
    valid = (p.norm(dim=1) != 0) & (t.norm(dim=1) != 0)
    return F.cosine_similarity(F.normalize(p[valid], p=2, dim=1), F.normalize(t[valid], p=2, dim=1)).mean()
