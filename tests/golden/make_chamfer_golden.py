"""Generate golden Chamfer / F-score vectors FROM THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference):
    python tests/golden/make_chamfer_golden.py
It imports the reference's utils/metrics/CD/chamfer_python.py (distChamfer,
the only CPU path of the op layer) and utils/metrics/CD/fscore.py unmodified
via importlib, evaluates them on seeded inputs and stores inputs + outputs as
tests/golden/chamfer_golden.npz.  The .npz is data (inputs and expected
outputs); no reference source travels with it.

cd_p / cd_t follow completion/model_utils.py:67-77 (calc_cd calls
cd()(gt, output); cd_p = (mean sqrt d1 + mean sqrt d2)/2, cd_t = mean d1 +
mean d2); model_utils itself cannot be imported here (it imports the CUDA
extensions at module import, :19-21), so those two lines are applied to the
reference's distChamfer outputs.
"""
import importlib.util
import os

import numpy as np
import torch

REF = "/root/reference/utils/metrics/CD"
HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def cases():
    g = torch.Generator().manual_seed(0)
    out = {}
    # the reference's own parity contract shape (unit_test.py:15-16)
    out["unit"] = (torch.rand(4, 100, 3, generator=g), torch.rand(4, 200, 3, generator=g))
    # BASELINE config 1: CD_L2 on random CPU tensors B=4, 2048 vs 2048
    out["cfg1"] = (torch.rand(4, 2048, 3, generator=g), torch.rand(4, 2048, 3, generator=g))
    # ragged sizes: not multiples of any tile, N < M, M = 1, N = 1
    out["ragged"] = (torch.rand(2, 513, 3, generator=g), torch.rand(2, 37, 3, generator=g))
    out["m1"] = (torch.rand(3, 70, 3, generator=g), torch.rand(3, 1, 3, generator=g))
    out["n1m1"] = (torch.rand(1, 1, 3, generator=g), torch.rand(1, 1, 3, generator=g))
    out["tile_edge"] = (torch.rand(1, 1025, 3, generator=g), torch.rand(1, 1040, 3, generator=g))
    # exact ties: lattice with spacing 1/8 (all arithmetic exact in fp32 and
    # fp64, so both forms tie exactly and the lowest index must win)
    lat = torch.stack(torch.meshgrid(*[torch.arange(5) / 8.0] * 3, indexing="ij"), -1).reshape(-1, 3)
    perm = torch.randperm(lat.shape[0], generator=g)
    a = lat[perm][None].repeat(2, 1, 1).contiguous()
    b = (lat + 1.0 / 16)[torch.randperm(lat.shape[0], generator=g)][None].repeat(2, 1, 1).contiguous()
    out["lattice_ties"] = (a.float(), b.float())
    # duplicated points (distance 0 ties)
    d = torch.rand(1, 64, 3, generator=g)
    out["duplicates"] = (torch.cat([d, d], 1), torch.cat([d, d, d], 1))
    return out


def main():
    cp = _load("chamfer_python")
    fs = _load("fscore")
    blob = {}
    for name, (a, b) in cases().items():
        d1, d2, i1, i2 = cp.distChamfer(a, b)
        f, p1, p2 = fs.fscore(d1, d2)
        f_loose, _, _ = fs.fscore(d1, d2, 0.01)
        cd_p = (torch.sqrt(d1).mean(1) + torch.sqrt(d2).mean(1)) / 2
        cd_t = d1.mean(1) + d2.mean(1)
        for k, v in dict(a=a, b=b, dist1=d1, dist2=d2, idx1=i1, idx2=i2, f=f, p1=p1,
                         p2=p2, f_loose=f_loose, cd_p=cd_p, cd_t=cd_t).items():
            blob["%s/%s" % (name, k)] = v.numpy()
    path = os.path.join(HERE, "chamfer_golden.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path), "bytes;", "torch", torch.__version__)


if __name__ == "__main__":
    main()
