"""SDF regularisers (python/regularizations.py:5-25)."""
import torch


def eval_discrete_laplacian_reg(data, _=None):
    """sum_voxels (c - mean of the 6 clamped neighbours)^2."""
    d = data[..., 0] if data.dim() == 4 else data

    def shifted(axis, step):
        n = d.shape[axis]
        idx = torch.clamp(torch.arange(n, device=d.device) + step, 0, n - 1)
        return d.index_select(axis, idx)

    nb = sum(shifted(a, s) for a in range(3) for s in (-1, 1)) / 6.0
    return ((d - nb) ** 2).sum()
