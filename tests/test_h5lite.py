"""the on-disk formats either side of the completion loop.

The reference reads MVP_*_CP.h5 (completion/dataset.py:21-46: `incomplete_pcds`,
`complete_pcds`, `labels`, partial i pairs with complete i // 26) and writes
`results.h5` / dataset `results` (completion/test.py:57-61) with h5py.  h5py is
absent from this image; completion/h5lite.py covers the subset of HDF5 those
files use.  Pinned against the REAL library in both directions:
  * reader -- fixtures tests/golden/mvp_tiny_*.h5 written by libhdf5 1.10
    (tests/golden/make_h5_fixtures.c/.sh): default property lists (what h5py's
    create_dataset(data=) produces), chunked + shuffle + gzip, libver=latest;
  * writer -- files written here are parsed by the real `h5dump` when the
    image has it (/opt/conda/bin), and always by the reader.
"""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

COMPLETION = os.path.join(ROOT, "completion")
if COMPLETION not in sys.path:
    sys.path.insert(0, COMPLETION)
GOLDEN = os.path.join(ROOT, "tests", "golden")
H5DUMP = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if os.path.exists("/opt/conda/bin/h5dump") else None)


def _expected():
    jk = (np.arange(24, dtype=np.float32) / 64).reshape(1, 8, 3)
    inc = np.arange(52, dtype=np.float32)[:, None, None] + jk
    com = -(np.arange(2, dtype=np.float32)[:, None, None] + jk)
    lab = (np.arange(52) // 26) * 5 + 3
    return inc, com, lab


@pytest.mark.parametrize("variant", ["default", "gzip", "latest"])
def test_reader_on_files_written_by_the_real_library(variant):
    import h5lite
    inc, com, lab = _expected()
    with h5lite.File(os.path.join(GOLDEN, "mvp_tiny_%s.h5" % variant), "r") as f:
        assert sorted(f.keys()) == ["complete_pcds", "incomplete_pcds", "labels"]
        assert f["incomplete_pcds"].shape == (52, 8, 3) and f["incomplete_pcds"].dtype == np.float32
        assert f["labels"].dtype == np.int64
        np.testing.assert_array_equal(f["incomplete_pcds"][()], inc)
        np.testing.assert_array_equal(f["complete_pcds"][()], com)
        np.testing.assert_array_equal(np.array(f["labels"][()]), lab)
        with pytest.raises(KeyError):
            f["results"]


def test_mvp_cp_pairs_partial_i_with_complete_i_div_26():
    """dataset.py:36-46: (label, partial, complete[index // 26]); test split
    yields the partial cloud only."""
    from dataset import MVP_CP
    inc, com, lab = _expected()
    ds = MVP_CP("train", os.path.join(GOLDEN, "mvp_tiny_gzip.h5"))
    assert len(ds) == 52
    for i in (0, 25, 26, 51):
        label, partial, complete = ds[i]
        assert label == lab[i]
        assert torch.equal(partial, torch.from_numpy(inc[i])) and torch.equal(complete, torch.from_numpy(com[i // 26]))
    t = MVP_CP("test", os.path.join(GOLDEN, "mvp_tiny_default.h5"))
    assert torch.equal(t[3], torch.from_numpy(inc[3]))
    with pytest.raises(ValueError):
        MVP_CP("nope")


def test_build_dataset_fails_loudly_without_files(tmp_path, monkeypatch):
    """A missing ./data/*.h5 must raise unless the cfg asks for synthetic data."""
    from dataset import SyntheticMVP, build_dataset, MVP_CP
    from train_utils import AttrDict
    monkeypatch.chdir(tmp_path)
    with pytest.raises(FileNotFoundError):
        build_dataset(AttrDict(num_points=2048), "train")
    assert isinstance(build_dataset(AttrDict(num_points=2048, synthetic=True), "train"), SyntheticMVP)
    os.makedirs(tmp_path / "d")
    shutil.copy(os.path.join(GOLDEN, "mvp_tiny_default.h5"), tmp_path / "d" / "MVP_Test_CP.h5")
    ds = build_dataset(AttrDict(num_points=2048, data_dir=str(tmp_path / "d")), "val")
    assert isinstance(ds, MVP_CP) and len(ds) == 52


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int32, np.int64, np.uint8])
def test_writer_round_trip(tmp_path, dtype):
    import h5lite
    rng = np.random.default_rng(3)
    res = (rng.random((5, 7, 3)) * 100).astype(dtype)
    path = str(tmp_path / "results.h5")
    with h5lite.File(path, "w") as f:
        f.create_dataset("results", data=res)
        with pytest.raises(ValueError):
            f.create_dataset("results", data=res)
    with h5lite.File(path, "r") as f:
        assert f.keys() == ["results"]
        got = f["results"][()]
    assert got.dtype == np.dtype(dtype) and np.array_equal(got, res)


@pytest.mark.skipif(H5DUMP is None, reason="no h5dump in this image")
def test_writer_output_is_read_by_the_real_library(tmp_path):
    """results.h5 as test.py writes it (dataset `results`, float32 (n, 2048, 3))
    plus a three-dataset MVP-layout file: `h5dump` must parse both and print
    the same numbers."""
    import h5lite
    rng = np.random.default_rng(0)
    res = rng.random((3, 2048, 3), dtype=np.float32)
    path = str(tmp_path / "results.h5")
    with h5lite.File(path, "w") as f:
        f.create_dataset("results", data=res)
    head = subprocess.run([H5DUMP, "-H", path], capture_output=True, text=True)
    assert head.returncode == 0, head.stderr
    assert 'DATASET "results"' in head.stdout and "H5T_IEEE_F32LE" in head.stdout
    assert "( 3, 2048, 3 )" in head.stdout
    raw = str(tmp_path / "results.bin")
    out = subprocess.run([H5DUMP, "-d", "/results", "-b", "LE", "-o", raw, path], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    np.testing.assert_array_equal(np.fromfile(raw, dtype="<f4").reshape(res.shape), res)

    inc, com, lab = _expected()
    path2 = str(tmp_path / "mvp.h5")
    with h5lite.File(path2, "w") as f:
        f.create_dataset("incomplete_pcds", data=inc)
        f.create_dataset("complete_pcds", data=com)
        f.create_dataset("labels", data=lab)
    for name, want, dt in (("incomplete_pcds", inc, "<f4"), ("complete_pcds", com, "<f4"), ("labels", lab, "<i8")):
        out = subprocess.run([H5DUMP, "-d", "/" + name, "-b", "LE", "-o", raw, path2], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        np.testing.assert_array_equal(np.fromfile(raw, dtype=dt).reshape(want.shape), want)


def test_not_hdf5_is_rejected(tmp_path):
    import h5lite
    p = tmp_path / "x.h5"
    p.write_bytes(b"not an hdf5 file at all" * 10)
    with pytest.raises(IOError):
        h5lite.File(str(p), "r")
