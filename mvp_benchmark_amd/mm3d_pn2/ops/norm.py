"""NaiveSyncBatchNorm1d / 2d -- NAME-ONLY re-exports (out of scope).

The reference's utils/mm3d_pn2/ops/norm.py:27-133 holds its one explicit collective (an
all_gather of batch statistics, :17,23) for detection heads; no completion or registration
model on the hot path has a BatchNorm layer that crosses ranks.  The names are kept so that
``from mm3d_pn2 import NaiveSyncBatchNorm1d`` keeps importing; they ARE PyTorch's own modules:
under DDP, ``torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)`` is the supported way to
synchronise statistics over RCCL, and this package does not re-implement it."""
from torch import nn as nn

NaiveSyncBatchNorm1d = nn.BatchNorm1d
NaiveSyncBatchNorm2d = nn.BatchNorm2d
