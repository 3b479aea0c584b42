"""read_hdf5_data (/root/reference/quantization/quantization.py:746-820).  The file reader is checked against what
the real h5py read from the same files (tests/golden/make_golden_hdf5.py, run under /opt/conda/bin/python3.9), the shuffle /
split against what the reference's own function returned (make_golden_hdf5_split.py)."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from golden.make_golden_hdf5_split import synthetic_archive
from quantization_amd import hdf5_data, read_hdf5_data
from quantization_amd.hdf5_data import Hdf5FormatError, MiniHdf5File

G = os.path.join(os.path.dirname(__file__), "golden", "hdf5")
EXP = np.load(os.path.join(G, "hdf5_expected.npz"))


@pytest.mark.parametrize("tag", ["small", "many", "mixed", "chunked"])
def test_builtin_reader_equals_h5py(tag):
    f = MiniHdf5File(os.path.join(G, f"hdf5_{tag}.hdf5"))
    keys = f.keys()
    assert keys == list(EXP[tag + "_keys"])            # h5py's iteration order (by name), which fixes the row order
    dim = f.read(keys[0]).shape[-1]
    frames = np.concatenate([np.ascontiguousarray(f.read(k)).reshape(-1, dim).astype(np.float16) for k in keys])
    assert np.array_equal(frames, EXP[tag + "_frames"])


def test_read_hdf5_data_from_a_file(monkeypatch):
    """(train, valid) of a small archive: float16 CPU tensors, valid = int(5 %) of the frames, rows = one shuffle of the
    concatenated datasets with numpy's global RNG (the reference raises on archives this small: module docstring)."""
    np.random.seed(7)
    train, valid = read_hdf5_data(os.path.join(G, "hdf5_small.hdf5"))
    frames = EXP["small_frames"].copy()
    np.random.seed(7)
    np.random.shuffle(frames)
    nv = int(0.05 * frames.shape[0])
    assert train.dtype == torch.float16 and valid.dtype == torch.float16 and train.device.type == "cpu"
    assert tuple(valid.shape) == (nv, 32) and tuple(train.shape) == (frames.shape[0] - nv, 32)
    assert np.array_equal(valid.numpy(), frames[:nv]) and np.array_equal(train.numpy(), frames[nv:])


def test_split_equals_the_reference(monkeypatch):
    exp = json.load(open(os.path.join(G, "hdf5_split_expected.json")))
    sets = synthetic_archive()
    closed = []

    class FakeArchive:                      # the datasets handed over in memory: the test is about stack / shuffle / split
        def __init__(self, fn):
            self.names = list(sets.keys())

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            closed.append(True)

        def shape(self, k):
            return sets[k].shape

        def rows(self, k, dim):
            return np.ascontiguousarray(sets[k]).reshape(-1, dim)

    monkeypatch.setattr(hdf5_data, "_Archive", FakeArchive)
    np.random.seed(exp["seed"])
    train, valid = read_hdf5_data("in-memory")
    assert closed == [True]                 # the archive is released once the frames are copied out
    assert list(train.shape) == exp["train_shape"] and list(valid.shape) == exp["valid_shape"]
    assert str(train.dtype) == exp["dtype"]
    assert hashlib.sha256(train.contiguous().numpy().tobytes()).hexdigest() == exp["train_sha256"]
    assert hashlib.sha256(valid.contiguous().numpy().tobytes()).hexdigest() == exp["valid_sha256"]


def test_inconsistent_dim_asserts(monkeypatch):
    sets = {"a": np.zeros((4, 8), np.float16), "b": np.zeros((4, 6), np.float16)}
    closed = []

    class FakeArchive:                      # the datasets handed over in memory: the test is about stack / shuffle / split
        def __init__(self, fn):
            self.names = list(sets.keys())

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            closed.append(True)

        def shape(self, k):
            return sets[k].shape

        def rows(self, k, dim):
            return np.ascontiguousarray(sets[k]).reshape(-1, dim)

    monkeypatch.setattr(hdf5_data, "_Archive", FakeArchive)
    with pytest.raises(AssertionError):                 # quantization.py:792
        read_hdf5_data("in-memory")


def test_not_an_hdf5_file(tmp_path):
    p = tmp_path / "x.hdf5"
    p.write_bytes(b"not hdf5" * 100)
    with pytest.raises(Hdf5FormatError):
        MiniHdf5File(str(p))


def test_broken_files_raise_the_format_error(tmp_path):
    from quantization_amd.hdf5_data import Hdf5FormatError, MiniHdf5File
    empty = tmp_path / "empty.h5"
    empty.write_bytes(b"")
    with pytest.raises(Hdf5FormatError):
        MiniHdf5File(str(empty))
    junk = tmp_path / "junk.h5"
    junk.write_bytes(b"x" * 4096)
    with pytest.raises(Hdf5FormatError):
        MiniHdf5File(str(junk))


def test_the_archive_is_closed_after_reading(tmp_path):
    import glob
    from quantization_amd.hdf5_data import MiniHdf5File
    src = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "hdf5", "*.hdf5")) +
                 glob.glob(os.path.join(os.path.dirname(__file__), "golden", "hdf5", "*.h5")))[0]
    with MiniHdf5File(src) as f:
        names = f.keys()
        a = np.array(f.read(names[0]))       # a copy: the mapping can go
    assert f.buf is None and a.size > 0
