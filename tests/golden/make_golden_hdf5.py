"""Writes the HDF5 fixtures of tests/test_hdf5_data.py with the real h5py, the way the reference's
test_write_hdf5.py:25-31 writes its training archive (`hf.create_dataset(f'dataset_{i}', data=x)` of float16
arrays), plus what h5py reads back from them (names in iteration order, the concatenated frames).

The build image's main interpreter has no h5py; /opt/conda/bin/python3.9 has h5py 3.3.0 (HDF5 1.10):
    /opt/conda/bin/python3.9 tests/golden/make_golden_hdf5.py
"""
import os

import h5py
import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hdf5")


def write(name, arrays, **kw):
    path = os.path.join(HERE, name)
    hf = h5py.File(path, "w")
    for key, a in arrays:
        hf.create_dataset(key, data=a, **kw)
    hf.close()
    hf = h5py.File(path, "r")
    keys = list(hf.keys())
    dim = hf[keys[0]].shape[-1]
    frames = np.concatenate([np.ascontiguousarray(hf[k][:]).reshape(-1, dim) for k in keys]).astype(np.float16)
    hf.close()
    return keys, frames


def main():
    rng = np.random.RandomState(5)
    out = {}
    # (a) the reference's layout at small size: float16 (B, dim) datasets named dataset_i, one with extra leading axes
    arrays = [(f"dataset_{i}", rng.randn(48, 32).astype(np.float16)) for i in range(12)]
    arrays.append(("extra_3d", rng.randn(3, 5, 32).astype(np.float16)))
    out["small_keys"], out["small_frames"] = write("hdf5_small.hdf5", arrays)
    # (b) many datasets: the group's B-tree gets internal nodes (the reference's archive has 976 datasets)
    arrays = [(f"dataset_{i}", rng.randn(2, 8).astype(np.float16)) for i in range(300)]
    out["many_keys"], out["many_frames"] = write("hdf5_many.hdf5", arrays)
    # (c) other element types a caller may have stored (converted to float16 like the reference's `ans[...] = array`)
    arrays = [("a_f32", rng.randn(10, 16).astype(np.float32)), ("b_f64", rng.randn(7, 16)), ("c_f16", rng.randn(5, 16).astype(np.float16))]
    out["mixed_keys"], out["mixed_frames"] = write("hdf5_mixed.hdf5", arrays)
    # (d) chunked storage (uncompressed), which create_dataset(chunks=...) produces
    arrays = [(f"d{i}", rng.randn(37, 24).astype(np.float16)) for i in range(3)]
    out["chunked_keys"], out["chunked_frames"] = write("hdf5_chunked.hdf5", arrays, chunks=(8, 24))
    np.savez_compressed(os.path.join(HERE, "hdf5_expected.npz"),
                        **{k: (np.array(v) if k.endswith("_frames") else np.array(v, dtype="U")) for k, v in out.items()})
    for f in ("hdf5_small.hdf5", "hdf5_many.hdf5", "hdf5_mixed.hdf5", "hdf5_chunked.hdf5", "hdf5_expected.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
