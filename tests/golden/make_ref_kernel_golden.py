"""Writes tests/golden/ref_kernel_golden.npz: OUTPUTS OF THE REFERENCE'S OWN KERNELS, run on an MI355X.

Run on a GPU box, with oracle/_ref built (oracle/build_ref_gpu.sh, done by build() in the container that has
/root/reference):

    gpurun -- 'python tests/golden/make_ref_kernel_golden.py gpurun_out/ref_kernel_golden.npz'

then copy the file to tests/golden/.  Inputs are NOT stored: every case regenerates them from the seeds below
(`conftest.rand_clouds` = numpy's default_rng, bit-stable).  Stored per case and per build ("" = hipcc's default contraction,
"_nofma" = -ffp-contract=off): the kernels' outputs, indices as int16/int32, values as float32 -- data only.
tests/test_oracle.py::test_oracle_matches_reference_kernel_outputs holds the CPU oracle against them (bit for bit in the
no-contraction mode; index for index and to the last place in the canonical mode), so the pin survives on machines without
a GPU and without oracle/_ref.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))

import ref_kernels as ref  # noqa: E402
from ref_kernel_cases import CASES, inputs  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy()


def run(kind, arg, x, v):
    """-> dict of output arrays of the reference kernel for one case."""
    if kind == "fps":
        return {"idx": host(ref.fps(dev(x["xyz"]), arg["m"], v))}
    if kind == "fps_dist":
        return {"idx": host(ref.fps_with_dist(dev(x["dist"]), arg["m"], v))}
    if kind == "ball_query":
        return {"idx": host(ref.ball_query(arg["lo"], arg["hi"], arg["s"], dev(x["xyz"]), dev(x["ctr"]), v))}
    if kind == "knn":
        i, d = ref.knn(arg["k"], dev(x["xyz"]), dev(x["ctr"]), v)
        return {"idx": host(i), "dist2": host(d)}
    if kind == "three_nn":
        d, i = ref.three_nn(dev(x["ctr"]), dev(x["xyz"]), v)
        return {"idx": host(i), "dist2": host(d)}
    if kind == "three_interpolate":
        return {"out": host(ref.three_interpolate(dev(x["feat"]), dev(x["idx"]), dev(x["w"]), v))}
    if kind == "gather":
        return {"out": host(ref.gather_points(dev(x["feat"]), dev(x["idx"]), v))}
    if kind == "group":
        return {"out": host(ref.grouping_operation(dev(x["feat"]), dev(x["idx"]), v))}
    if kind == "chamfer":
        d1, d2, i1, i2 = ref.chamfer_forward(dev(x["a"]), dev(x["c"]), v)
        return {"dist1": host(d1), "dist2": host(d2), "idx1": host(i1), "idx2": host(i2)}
    if kind == "emd":
        runs = [ref.emd_forward(dev(x["a"]), dev(x["c"]), arg["eps"], arg["iters"], v) for _ in range(3)]
        assert all(torch.equal(runs[0][1], r[1]) for r in runs[1:]), "a fixture case must be deterministic on the reference"
        d, a, _ = runs[0]
        res = {"dist": host(d), "assignment": host(a)}
        if arg.get("grad"):
            res["gradxyz1"] = host(ref.emd_backward(dev(x["a"]), dev(x["c"]), dev(x["g"]), a, v)[0])
        return res
    raise KeyError(kind)


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "ref_kernel_golden.npz")
    blob = {}
    for name, (kind, arg) in CASES.items():
        x = inputs(name)
        for v in ("", "_nofma"):
            for key, arr in run(kind, arg, x, v).items():
                if arr.dtype == np.int32 and arr.size and arr.max() < 32768 and arr.min() >= -32768:
                    arr = arr.astype(np.int16)
                blob[f"{name}/{v or 'default'}/{key}"] = arr
    blob["_meta/device"] = np.frombuffer(torch.cuda.get_device_name(0).encode(), dtype=np.uint8)
    np.savez_compressed(out, **blob)
    print(out, os.path.getsize(out), "bytes,", len(blob), "arrays")


if __name__ == "__main__":
    main()
