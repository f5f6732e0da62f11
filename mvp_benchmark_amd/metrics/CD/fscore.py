"""F-score of a completion at a squared-distance threshold.

Counterpart of the reference's utils/metrics/CD/fscore.py:3-16: same name,
arguments and (fscore, precision_1, precision_2) return triple; clouds with no
point under the threshold on either side score 0 (the reference's NaN -> 0).
"""
import torch


def fscore(dist1, dist2, threshold=0.0001):
    """dist1 (B, N): squared NN distances gt -> prediction; dist2 (B, M): the
    other direction (both from `cd`).  `threshold` applies to the SQUARED
    distance.  Returns three (B,) tensors."""
    precision_1 = torch.mean((dist1 < threshold).float(), dim=1)
    precision_2 = torch.mean((dist2 < threshold).float(), dim=1)
    total = precision_1 + precision_2
    harmonic = 2 * precision_1 * precision_2 / total
    # 0/0 (nothing matched) and empty clouds give NaN: define those as 0
    return torch.where(torch.isnan(harmonic), torch.zeros_like(harmonic), harmonic), precision_1, precision_2
