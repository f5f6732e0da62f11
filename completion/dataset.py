"""MVP completion dataset -- counterpart of the reference's
completion/dataset.py:8-46.

On-disk format (completion/README.md:21-32): HDF5 with `incomplete_pcds`
(62400, 2048, 3), `complete_pcds` (2400, 2048, 3) and `labels`; partial cloud i
pairs with complete cloud i // 26.  `MVP_CP` reads those files with h5py when it
is installed and with the in-tree reader (h5lite.py) otherwise.
`SyntheticMVP` yields the same sample tuples from a seeded generator (no dataset
or network exists in the build environment): every "shape" is a uniform cloud
in [0,1)^3 and its 26 partial views are half-space cuts re-sampled to 2048
points.  It is used ONLY when the cfg says `synthetic: True`; a missing .h5
file is an error, as in the reference.
"""
import logging
import os

import numpy as np
import torch
import torch.utils.data as data

VIEWS_PER_SHAPE = 26
_FILES = {"train": './data/MVP_Train_CP.h5', "val": './data/MVP_Test_CP.h5',
          "test": './data/MVP_ExtraTest_Shuffled_CP.h5'}


def h5_module():
    """h5py when installed, else the in-tree reader / writer with the same calls."""
    try:
        import h5py
        return h5py
    except ImportError:
        import h5lite
        return h5lite


class MVP_CP(data.Dataset):
    def __init__(self, prefix="train", file_path=None):
        if prefix not in _FILES:
            raise ValueError("ValueError prefix should be [train/val/test] ")
        self.prefix = prefix
        self.file_path = file_path or _FILES[prefix]
        with h5_module().File(self.file_path, 'r') as f:
            self.input_data = np.array(f['incomplete_pcds'][()])
            if prefix != "test":
                self.gt_data = np.array(f['complete_pcds'][()])
                self.labels = np.array(f['labels'][()])
        self.len = self.input_data.shape[0]

    def __len__(self):
        return self.len

    def __getitem__(self, index):
        partial = torch.from_numpy(self.input_data[index])
        if self.prefix == "test":
            return partial
        complete = torch.from_numpy(self.gt_data[index // VIEWS_PER_SHAPE])
        return self.labels[index], partial, complete


class SyntheticMVP(data.Dataset):
    """Same (label, partial (2048,3), complete (num_points,3)) tuples as MVP_CP,
    generated on the fly from a seed."""

    def __init__(self, prefix="train", num_shapes=8, num_points=2048, num_partial=2048,
                 views=VIEWS_PER_SHAPE, seed=0):
        self.prefix = prefix
        self.num_shapes, self.num_points, self.num_partial = num_shapes, num_points, num_partial
        self.views = views
        self.seed = seed + {"train": 0, "val": 1000, "test": 2000}[prefix]
        self.len = num_shapes * views

    def __len__(self):
        return self.len

    def _complete(self, shape):
        g = torch.Generator().manual_seed(self.seed * 100003 + shape)
        return torch.rand(self.num_points, 3, generator=g)

    def __getitem__(self, index):
        shape = index // self.views
        complete = self._complete(shape)
        g = torch.Generator().manual_seed(self.seed * 7919 + index)
        # a "view": keep the half-space facing a random direction, resample to 2048
        direction = torch.randn(3, generator=g)
        keep = ((complete - 0.5) @ direction) > 0
        visible = complete[keep] if int(keep.sum()) > 16 else complete
        pick = torch.randint(0, visible.shape[0], (self.num_partial,), generator=g)
        partial = visible[pick]
        if self.prefix == "test":
            return partial
        return shape % 16, partial, complete


def build_dataset(args, prefix):
    """MVP_CP on ./data/*.h5 (or the cfg's `data_dir`); SyntheticMVP only when the
    cfg sets `synthetic: True`.  A missing file raises -- training on noise because
    of a wrong working directory must not happen silently."""
    if not args.get("synthetic"):
        path = _FILES[prefix]
        if args.get("data_dir"):
            path = os.path.join(args.get("data_dir"), os.path.basename(path))
        if not os.path.exists(path):
            raise FileNotFoundError(
                "%s not found (run from completion/ with the MVP files under ./data, set `data_dir`, "
                "or set `synthetic: True` in the cfg to use the generated stand-in)" % path)
        return MVP_CP(prefix, path)
    logging.warning("build_dataset(%s): cfg has synthetic=True -- using generated stand-in clouds, "
                    "NOT the MVP dataset", prefix)
    shapes = args.get("synthetic_%s_shapes" % ("train" if prefix == "train" else "val")) or 4
    return SyntheticMVP(prefix, num_shapes=int(shapes), num_points=int(args.get("num_points") or 2048),
                        seed=int(args.get("manual_seed") or 0))
