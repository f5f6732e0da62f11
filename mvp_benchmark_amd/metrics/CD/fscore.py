"""F-score at a threshold on squared Chamfer distances -- counterpart of the
reference's utils/metrics/CD/fscore.py:3-16 (same name, arguments, return
triple and NaN->0 rule)."""
import torch


def fscore(dist1, dist2, threshold=0.0001):
    """
    :param dist1: (B, N) squared distances gt -> prediction
    :param dist2: (B, M) squared distances prediction -> gt
    :param threshold: threshold on the SQUARED distance
    :return: fscore, precision_1, precision_2  -- each (B,)
    """
    precision_1 = (dist1 < threshold).float().mean(dim=1)
    precision_2 = (dist2 < threshold).float().mean(dim=1)
    f = 2 * precision_1 * precision_2 / (precision_1 + precision_2)
    f[torch.isnan(f)] = 0
    return f, precision_1, precision_2
