"""Cases of tests/golden/ref_kernel_golden.npz (outputs of the reference's own kernels on an MI355X): name -> (kind, args),
and the seeded inputs of each.  Shared by tests/golden/make_ref_kernel_golden.py (GPU box) and tests/test_oracle.py (CPU)."""
import numpy as np

from conftest import rand_clouds


def _lattice(k):
    g = np.stack(np.meshgrid(*[np.arange(k, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(1, -1, 3) / k
    return np.concatenate([g, g[:, ::-1]], 0).copy()


CASES = {
    "fps_1000_300": ("fps", dict(m=300)),
    "fps_2048_512": ("fps", dict(m=512)),
    "fps_8193_256": ("fps", dict(m=256)),           # the 1024-thread block with a ragged tail
    "fps_lattice": ("fps", dict(m=200)),            # exact ties: the bit-reversed slot order of the LDS tree decides
    "fps_dist_700_128": ("fps_dist", dict(m=128)),
    "ball_query_r02_s32": ("ball_query", dict(lo=0.0, hi=0.2, s=32)),
    "ball_query_ring_s16": ("ball_query", dict(lo=0.05, hi=0.3, s=16)),
    "ball_query_lattice": ("ball_query", dict(lo=0.0, hi=0.3, s=16)),
    "knn_k8": ("knn", dict(k=8)),
    "knn_k20": ("knn", dict(k=20)),
    "knn_lattice_k9": ("knn", dict(k=9)),
    "three_nn": ("three_nn", dict()),
    "three_nn_lattice": ("three_nn", dict()),
    "three_interpolate": ("three_interpolate", dict()),
    "gather": ("gather", dict()),
    "group": ("group", dict()),
    "chamfer_777_1300": ("chamfer", dict()),
    "chamfer_100_200": ("chamfer", dict()),
    "chamfer_dups": ("chamfer", dict()),
    "emd_1024_50": ("emd", dict(eps=0.005, iters=50, grad=True)),        # the training setting
    "emd_1024_forced_last": ("emd", dict(eps=0.005, iters=2)),   # emd_cuda.cu:200 with most persons unassigned
    "emd_2048_3000": ("emd", dict(eps=0.004, iters=3000)),    # the eval setting
    "emd_3072_300": ("emd", dict(eps=0.01, iters=300)),       # three blocks per cloud
}


def inputs(name):
    kind = CASES[name][0]
    if name == "fps_lattice":
        return {"xyz": _lattice(8)}
    if kind == "fps":
        n = int(name.split("_")[1])
        return {"xyz": rand_clouds(100 + n, 3, n, 3)}
    if kind == "fps_dist":
        x = rand_clouds(5, 3, 700, 3)
        return {"dist": ((x[:, :, None, :] - x[:, None, :, :]) ** 2).sum(-1).astype(np.float32)}
    if "lattice" in name:
        xyz = _lattice(8)
        return {"xyz": xyz, "ctr": xyz[:, ::5].copy()}
    if kind in ("ball_query", "knn", "three_nn"):
        return {"xyz": rand_clouds(1, 3, 1024, 3), "ctr": rand_clouds(2, 3, 200, 3)}
    rng = np.random.default_rng(4)
    if kind == "three_interpolate":
        w = rand_clouds(8, 2, 300, 3)
        return {"feat": rand_clouds(3, 2, 12, 512), "idx": rng.integers(0, 512, (2, 300, 3)).astype(np.int32),
                "w": (w / w.sum(-1, keepdims=True)).astype(np.float32)}
    if kind == "gather":
        return {"feat": rand_clouds(3, 2, 12, 512), "idx": rng.integers(0, 512, (2, 100)).astype(np.int32)}
    if kind == "group":
        return {"feat": rand_clouds(3, 2, 12, 512), "idx": rng.integers(0, 512, (2, 32, 8)).astype(np.int32)}
    if name == "chamfer_dups":
        g = _lattice(6)[:1]
        return {"a": np.concatenate([g, g], 1).copy(), "c": (g + np.float32(0.125)).copy()}
    if kind == "chamfer":
        n, m = (int(t) for t in name.split("_")[1:])
        return {"a": rand_clouds(n, 2, n, 3), "c": rand_clouds(m + 1, 2, m, 3)}
    if kind == "emd":
        n, it = int(name.split("_")[1]), CASES[name][1]["iters"]
        b = 1 if n > 2048 else 2
        return {"a": rand_clouds(n + it, b, n, 3), "c": rand_clouds(n + it + 1, b, n, 3), "g": rand_clouds(n + it + 2, b, n)}
    raise KeyError(name)
