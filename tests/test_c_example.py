"""The C ABI used from plain C++ (examples/encode_c_abi.cpp): builds here (no GPU needed to compile),
runs on the MI355X and its codes / reconstruction are checked against the CPU oracle."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "_build", "encode_c_abi")


def _build():
    import __graft_entry__ as g
    g.build()
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    src = os.path.join(ROOT, "examples", "encode_c_abi.cpp")
    lib = os.path.join(ROOT, "quantization_amd", "lib")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(lib, "libmcq_hip.so"))):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", src, "-I" + os.path.join(ROOT, "include"),
                               "-L" + lib, "-lmcq_hip", "-Wl,-rpath," + lib, "-o", EXE])
    return EXE


def test_example_compiles_and_links_against_the_abi():
    assert os.path.exists(_build())


def _lcg_stream(n, seed=12345):
    out = np.empty(n, np.float32)
    s = seed
    mask = (1 << 64) - 1
    for i in range(n):
        s = (s * 6364136223846793005 + 1442695040888963407) & mask
        out[i] = np.float32(((s >> 40) & 0xFFFFFF) / 8388608.0) - np.float32(1.0)
    return out


@pytest.mark.gpu
def test_example_output_matches_oracle():
    from oracle.oracle import OracleQuantizer
    exe = _build()
    N, K, D, B, iters = 4, 256, 96, 300, 3
    with tempfile.TemporaryDirectory() as tmp:
        fc, fo = os.path.join(tmp, "codes.bin"), os.path.join(tmp, "dec.bin")
        subprocess.check_call([exe, fc, fo, str(B)])
        codes = np.fromfile(fc, np.uint8).reshape(B, N)
        dec = np.fromfile(fo, np.float32).reshape(B, D)
    stream = _lcg_stream(N * K * D + N * K + B * D)
    centers = (np.float32(0.5) * stream[:N * K * D]).reshape(N, K, D)
    weight = (np.float32(0.5) * centers).reshape(N * K, D)
    bias = np.float32(0.05) * stream[N * K * D:N * K * D + N * K]
    x = stream[N * K * D + N * K:].reshape(B, D)
    o = OracleQuantizer(centers, 0.0, weight, bias, 0.0)
    assert np.array_equal(codes, o.encode(x, iters))
    assert np.array_equal(dec, o.decode(codes))
