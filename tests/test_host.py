"""CPU tests of the host side: the C-ABI library loads and exports every
symbol include/mvpops.h declares (no compute without a GPU), the Python
operator surface mirrors the reference's names, the product path refuses to
run on CPU tensors (no fallback), and the pure-PyTorch counterparts
(fscore, distChamfer, calc_cd formulas) match the reference-generated golden
vectors bit for bit."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch
from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from mvp_benchmark_amd import _lib
    header = open(os.path.join(ROOT, "include", "mvpops.h")).read()
    declared = set(re.findall(r"\b(mvp_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 18
    assert declared == set(_lib.exported_symbols())
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mvp_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define MVP_ABI_VERSION (\d+)", header).group(1))
    # per cloud: state 132 B/pt + bound broadcast buffers + cell offsets + barrier granules (3 x 256 B), the bid areas of
    # the gathered-bid rounds (2 parities x (256 bids x 16 B + 8 member words)), hand-over record (96 B), statistics (16 B)
    assert lib.mvp_emd_scratch_bytes(64, 16384) == 64 * (16384 * 132 + 8 * 2048 * 8 + 1732 * 4 + 768 + 2 * (2 * 256 + 8) * 8 + 96 + 16)
    assert lib.mvp_emd_scratch_bytes(2, 32768) == 2 * (32768 * 132 + 8 * 2048 * 8 + 1732 * 4 + 768 + 2 * (2 * 256 + 8) * 8 + 96 + 16)
    # the knobs are process-wide: restore what is touched
    try:
        assert lib.mvp_emd_configure(-1, -1, 0, -1) == 0
        assert lib.mvp_emd_configure(3, -1, -1, -1) == -2           # MVP_EBADARG: cluster width
        assert lib.mvp_emd_configure(-1, -1, -1, 0) == -2           # MVP_EBADARG: resident cap 1..64
        assert lib.mvp_emd_configure(-1, -1, -1, 65) == -2
        assert lib.mvp_emd_configure(-1, -1, -1, 8) == 0
    finally:
        assert lib.mvp_emd_configure(0, -1, _lib.EMD_DEFAULT_SPLIT, 16) == 0


def test_argument_guards_need_no_gpu():
    """Shape guards return error codes before anything is launched."""
    from mvp_benchmark_amd import _lib
    lib = _lib.load()
    null = ctypes.c_void_p(0)
    # emd: n % 1024 != 0, b > 512 (emd_cuda.cu:236-249) -> MVP_EBADSHAPE
    assert lib.mvp_emd_forward(1, 1000, null, null, null, null, 0.005, 50, null, 0, null) == -1
    assert lib.mvp_emd_forward(513, 1024, null, null, null, null, 0.005, 50, null, 0, null) == -1
    assert lib.mvp_emd_forward(1, 1024, null, null, null, null, 0.005, 0, null, 0, null) == -1
    assert lib.mvp_emd_forward(1, 1024, null, null, null, null, 0.0, 50, null, 0, null) == -2     # eps must be > 0
    # null buffers -> MVP_EBADARG
    assert lib.mvp_emd_forward(1, 1024, null, null, null, null, 0.005, 50, null, 0, null) == -2
    assert lib.mvp_chamfer_forward(1, 8, 8, null, null, null, null, null, null, null) == -2
    assert lib.mvp_knn(1, 8, 8, 101, null, null, null, null, null) == -1
    assert lib.mvp_knn(1, 8, 8, 0, null, null, null, null, null) == -1
    # empty batch is a successful no-op
    assert lib.mvp_chamfer_forward(0, 8, 8, null, null, null, null, null, null, null) == 0
    assert lib.mvp_gather_points(0, 3, 8, 8, null, null, null, null) == 0


def test_guards_and_scratch_sizes_of_the_widened_entry_points():
    """Host logic of the entry points beyond the reference's operator set: shape
    guards, scratch formulas, no launch."""
    from mvp_benchmark_amd import _lib
    lib = _lib.load()
    null = ctypes.c_void_p(0)
    # transposed scatter gradients: chunk 1536 up to 2048 destinations, 3072 above; not covered beyond 8192
    # per (cloud, chunk): u16 offsets padded to 8 bytes + u16 entries (weighted: {column, weight} pairs of 8 bytes)
    up8 = lambda v: (v + 7) // 8 * 8
    assert _lib.scatter_scratch_bytes(2, 1024, 3000, 1) == 2 * 2 * up8(up8((1024 + 1) * 2) + 1536 * 2)
    assert _lib.scatter_scratch_bytes(1, 3072, 49152, 1) == 16 * up8(up8((3072 + 1) * 2) + 3072 * 2)
    assert _lib.scatter_scratch_bytes(1, 1024, 3072, 3) == 2 * up8(up8((1024 + 1) * 2) + 1536 * 3 * 8)
    assert _lib.scatter_scratch_bytes(1, 8193, 100, 1) == 0 and _lib.scatter_scratch_bytes(1, 100, 100, 2) == 0
    # ... and the _ws entry points fall back to the plain ones (which accept the empty batch)
    assert lib.mvp_gather_points_grad_ws(0, 3, 8, 8, null, null, null, null, 0, 0, null) == 0
    assert lib.mvp_three_interpolate_grad_ws(1, 3, 8, 0, null, null, null, null, null, 0, 0, null) == -1    # m == 0
    # Gram top-k
    assert lib.mvp_topk_gram(1, 8, 9, null, null, null, null) == -1        # k > n
    assert lib.mvp_topk_gram(1, 8, 0, null, null, null, null) == -1
    assert lib.mvp_topk_gram(1, 8, 4, null, null, null, null) == -2
    assert lib.mvp_topk_gram(0, 8, 4, null, null, null, null) == 0
    # shared-weight aggregation: share in {1,2,4,8,16}
    assert lib.mvp_share_weighted_sum(1, 3, 2, 4, 8, null, null, null, null) == -1
    assert lib.mvp_share_weighted_sum(1, 8, 2, 4, 8, null, null, null, null) == -2
    assert lib.mvp_share_weighted_sum(0, 8, 2, 4, 8, null, null, null, null) == 0
    assert lib.mvp_share_weighted_sum_grad(1, 8, 2, 4, 8, null, null, null, null, null, null) == -2
    # pointwise weight gradient: cout <= 64, positions a multiple of 4; one partial per run of 1024 positions
    assert _lib.pointwise_wgrad_scratch_bytes(2, 24, 24, 4096) == 2 * 4 * (24 * 24 + 24) * 4
    assert _lib.pointwise_wgrad_scratch_bytes(1, 24, 65, 4096) == 0
    assert _lib.pointwise_wgrad_scratch_bytes(1, 24, 24, 4094) == 0
    assert lib.mvp_pointwise_wgrad(1, 24, 24, 4094, null, null, null, null, null, 0, null) == -1
    assert lib.mvp_pointwise_wgrad(1, 24, 24, 4096, null, null, null, null, null, 0, null) == -2
    # chamfer / fps scratch
    assert _lib.fps_scratch_bytes(3, 16384) == 3 * 16384 * 16
    assert lib.mvp_furthest_point_sampling_sorted(1, 0, 4, null, null, null, null, 0, null) == -1


def test_operator_surface_matches_reference_names():
    import mvp_benchmark_amd.metrics as metrics
    import mvp_benchmark_amd.mm3d_pn2 as pn2
    assert metrics.__all__ == ['cd', 'fscore', 'emd']
    for name in ['ball_query', 'knn', 'furthest_point_sample',
                 'furthest_point_sample_with_dist', 'three_interpolate',
                 'three_nn', 'gather_points', 'grouping_operation',
                 'group_points', 'GroupAll', 'QueryAndGroup', 'Points_Sampler',
                 'NaiveSyncBatchNorm1d', 'NaiveSyncBatchNorm2d']:
        assert hasattr(pn2, name), name
    # `group_points` resolves to the submodule, as in the reference
    assert pn2.group_points.__name__.endswith("group_points.group_points")
    assert isinstance(metrics.cd(), torch.nn.Module)
    assert isinstance(metrics.emd(), torch.nn.Module)


def test_drop_in_utils_shims_import():
    """`sys.path.append("../utils"); from metrics import cd, fscore, emd`
    (completion/model_utils.py:19-21) keeps working against utils/."""
    import subprocess
    import sys
    code = ("import sys; sys.path.append(%r); "
            "from metrics import cd, fscore, emd; "
            "from mm3d_pn2 import furthest_point_sample, gather_points, "
            "grouping_operation, ball_query, three_nn; print('ok')"
            % os.path.join(ROOT, "utils"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                         cwd=os.path.join(ROOT, "completion"))
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


def test_no_cpu_fallback():
    from mvp_benchmark_amd._lib import MvpOpsError
    from mvp_benchmark_amd.metrics import cd, emd
    from mvp_benchmark_amd.mm3d_pn2 import furthest_point_sample, gather_points
    a = torch.rand(1, 16, 3)
    with pytest.raises((MvpOpsError, AssertionError, RuntimeError)):
        cd()(a, a)
    with pytest.raises((MvpOpsError, AssertionError, RuntimeError)):
        emd()(torch.rand(1, 1024, 3), torch.rand(1, 1024, 3), 0.005, 5)
    with pytest.raises((MvpOpsError, AssertionError, RuntimeError)):
        furthest_point_sample(a, 4)
    with pytest.raises((MvpOpsError, AssertionError, RuntimeError)):
        gather_points(torch.rand(1, 3, 16), torch.zeros(1, 4, dtype=torch.int32))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under mvp_benchmark_amd/,
    utils/ or completion/ may reference it."""
    bad = []
    for top in ("mvp_benchmark_amd", "utils", "completion"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".cpp", ".h")):
                    txt = open(os.path.join(dirpath, f)).read()
                    if re.search(r"^\s*(import|from)\s+oracle\b", txt, re.M) or "libmvp_oracle" in txt:
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_python_counterparts_match_reference_golden(chamfer_golden):
    """fscore (fscore.py:3-16), distChamfer (chamfer_python.py:18-39) and the
    calc_cd reductions (model_utils.py:67-77): bit-for-bit on this torch."""
    from mvp_benchmark_amd.metrics import fscore
    from mvp_benchmark_amd.metrics.CD.chamfer_python import distChamfer
    for name, c in chamfer_golden.items():
        a, b = torch.tensor(c["a"]), torch.tensor(c["b"])
        for chunk in (None, 1):
            d1, d2, i1, i2 = distChamfer(a, b, chunk=chunk)
            np.testing.assert_array_equal(d1.numpy(), c["dist1"], err_msg=name)
            np.testing.assert_array_equal(d2.numpy(), c["dist2"], err_msg=name)
            np.testing.assert_array_equal(i1.numpy(), c["idx1"], err_msg=name)
            np.testing.assert_array_equal(i2.numpy(), c["idx2"], err_msg=name)
        f, p1, p2 = fscore(d1, d2)
        np.testing.assert_array_equal(f.numpy(), c["f"])
        np.testing.assert_array_equal(p1.numpy(), c["p1"])
        np.testing.assert_array_equal(p2.numpy(), c["p2"])
        np.testing.assert_array_equal(fscore(d1, d2, 0.01)[0].numpy(), c["f_loose"])
        cd_p = (torch.sqrt(d1).mean(1) + torch.sqrt(d2).mean(1)) / 2
        cd_t = d1.mean(1) + d2.mean(1)
        np.testing.assert_array_equal(cd_p.numpy(), c["cd_p"])
        np.testing.assert_array_equal(cd_t.numpy(), c["cd_t"])


def test_emd_lazy_status_check_raises_one_call_late():
    """emd_module files every call's status words (async copy to pinned memory) and examines them
    at the next forward / at check(): a negative word (abandoned cluster wait, internal check) is
    an MvpOpsError, never a silent NaN.  Exercised here with hand-made pending entries."""
    import torch
    from mvp_benchmark_amd import _lib
    from mvp_benchmark_amd.metrics.EMD import emd_module

    class Done:
        def query(self):
            return True

        def synchronize(self):
            pass

    class NotYet(Done):
        def query(self):
            return False

    emd_module._PENDING.clear()
    emd_module._PENDING.append((Done(), torch.tensor([3000, 3000]), "ok call"))
    emd_module._PENDING.append((NotYet(), torch.tensor([3000, -2]), "failed call, still in flight"))
    emd_module.check(block=False)                 # the finished one is fine, the other stays filed
    assert len(emd_module._PENDING) == 1
    with pytest.raises(_lib.MvpOpsError, match="failed call"):
        emd_module.check(block=True)
    assert not emd_module._PENDING
    emd_module.check()                            # nothing pending: no-op


def test_emd_records_reads_the_scratch_tail():
    """_lib.emd_records: per-cloud hand-over records (20 ints) + statistics (2 int64) at the END of the scratch
    buffer, whatever its size -- the layout csrc/emd_common.h writes (no GPU needed: a synthetic buffer)."""
    import numpy as np
    import torch
    from mvp_benchmark_amd import _lib
    b, nbytes = 3, 4096
    buf = np.zeros(nbytes, np.uint8)
    stats = np.array([[3000, 111], [2999, 222], [-2, 0]], np.int64)
    rec = np.zeros((b, _lib.EMD_RECORD_INTS), np.int32)
    rec[:, 0] = [0, 0, 300]; rec[:, 1] = [150, 90, 201]; rec[:, 18] = [70, 0, 101]; rec[:, 19] = [2 * 16 + 8, 0, 1 * 16 + 4]
    buf[nbytes - b * 16:] = stats.view(np.uint8).ravel()
    buf[nbytes - b * 16 - rec.nbytes: nbytes - b * 16] = rec.view(np.uint8).ravel()
    got = _lib.emd_records(torch.from_numpy(buf), nbytes, b)
    assert got["rounds"].tolist() == [3000, 2999, -2] and got["bids"].tolist() == [111, 222, 0]
    assert got["next_round"].tolist() == [0, 0, 300] and got["unassigned"].tolist() == [150, 90, 201]
    assert got["first_handover"].tolist() == [70, 0, 101]
    assert got["final_width"].tolist() == [8, 0, 4] and got["final_launch"].tolist() == [2, 0, 1]


def test_pointwise_routing_rules():
    """mvp_benchmark_amd/pointwise.py: which per-cloud GEMMs go to the MFMA kernels (host logic, no GPU):
    short reductions always; long ones only with a chip's worth of 128 x 128 output tiles and full column tiles."""
    from mvp_benchmark_amd import pointwise as pw
    assert pw.MFMA_TRAIN and pw.MFMA_DGRAD and pw.MFMA_MIN_CH == 1
    fits = pw._gemm_fits
    assert fits(64, 1024, 512, 2048) and fits(64, 512, 1536, 384)          # VRCNet's widest layers
    assert fits(32, 48, 240, 1024)                                         # few workgroups but a short reduction
    assert not fits(32, 1024, 2824, 64) and not fits(32, 1024, 1800, 64)   # ECG, 64-point level: half-empty column tiles
    assert not fits(32, 768, 1864, 256)                                    # 6 x 2 x 32 = 384 workgroups < 512
    # round 6: a reduction of exactly 512 stays on the MFMA kernel even with one row of tiles -- as the data gradient of
    # (64, 128 -> 512, 384) it takes 0.050 ms against the library's 0.108, (64, 512 -> 512, 384) forward 0.127 against 0.145
    # (tools/bench_conv_passes.py); only (64, 512 -> 192, 384) forward loses 0.015 ms
    assert fits(64, 128, 512, 384) and fits(64, 32, 512, 384) and not fits(64, 128, 516, 384)
    assert fits(64, 512, 128, 384)                                         # the same layers' data gradients


def test_bench_settle_and_model_level_note():
    """bench.py's untimed pre-warm-up of the model-level steps stops once three consecutive steps agree (and gives up at its
    bound); the note that travels in the bench line formats (a '%' in it once cost the line its model-level figures)."""
    import importlib
    import time
    import types
    bench = importlib.import_module("bench")
    durations = iter([0.05, 0.04, 0.03, 0.03, 0.003, 0.003, 0.003, 0.003, 0.003, 0.003])
    calls = []

    def step():
        calls.append(1)
        time.sleep(next(durations, 0.003))

    n = bench.settle(step, max_steps=10)
    assert 7 <= n <= 10 and len(calls) == n
    jitter = iter([0.002, 0.02] * 10)
    assert bench.settle(lambda: time.sleep(next(jitter)), max_steps=8) == 8
    note = bench.model_level_note(types.SimpleNamespace(points=16384, eps=0.004, iters=3000))
    assert "5 %" in note and "16384" in note and "0.004 x 3000" in note
