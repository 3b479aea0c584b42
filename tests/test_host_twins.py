"""Host twins of the C ABI (oracle/mcq_host.h): same signatures and argument checks as
include/mcq.h on host pointers.  CPU part: signatures line up, the twins reproduce the reference
fixtures, and rejected argument lists get the same code from both libraries.  GPU part: one
argument list through mcq_* (device pointers) and mcq_*_host (host copies), outputs bit-identical."""
import os
import re

import numpy as np
import pytest

import host_twins as ht
from golden import fixtures
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _protos(path, suffix=""):
    txt = re.sub(r"/\*.*?\*/", " ", open(path).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int|size_t)\s+(mcq_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", txt):
        name = m.group(2)
        if suffix and not name.endswith(suffix):
            continue
        out[name[:len(name) - len(suffix)] if suffix else name] = (m.group(1), re.sub(r"\s+", " ", m.group(3)).strip())
    return out


def test_twin_signatures_equal_the_device_abi():
    dev = _protos(os.path.join(ROOT, "include", "mcq.h"))
    host = _protos(os.path.join(ROOT, "oracle", "mcq_host.h"), "_host")
    assert set(host) == set(ht.TWINS)
    H = ht.lib()
    for name, proto in host.items():
        assert dev[name] == proto, (name, dev[name], proto)
        assert hasattr(H, name + "_host")


def _device_lib():
    import __graft_entry__ as g
    g.build()
    from quantization_amd import _lib
    return _lib.lib()


def test_rejected_arguments_get_the_same_code():
    L, H = _device_lib(), ht.lib()
    enc = [(None, 4, None, 1.0, 8, 8, 64, 1, None, None, None, 0, None),
           (None, 4, None, 1.0, 8, 512, 64, 1, None, None, None, 0, None),
           (None, 4, None, 1.0, 3, 256, 64, 1, None, None, None, 0, None),
           (None, -1, None, 1.0, 8, 256, 64, 1, None, None, None, 0, None),
           (None, 4, None, 1.0, 8, 256, 64, 61, None, 1, None, 0, None),
           (None, 4, None, 1.0, 8, 256, 64, 1, 1, 1, None, 0, None),        # both outputs given
           (None, 4, None, 1.0, 8, 256, 64, 1, 1, None, None, 0, None),     # null x / prepared / workspace
           (None, 0, None, 1.0, 8, 256, 64, 1, 1, None, None, 0, None)]     # empty batch: accepted
    for a in enc:
        assert L.mcq_encode(*a) == H.mcq_encode_host(*a), a
    dec = [(None, 2, 8, 4, None, 8, 256, 64, None, None), (None, 1, 3, 4, None, 8, 256, 64, None, None),
           (None, 1, 8, -1, None, 8, 256, 64, None, None), (None, 1, 8, 4, None, 8, 1024, 64, None, None),
           (None, 8, 8, 0, None, 8, 256, 64, None, None), (None, 1, 8, 4, None, 8, 256, 64, None, None)]
    for a in dec:
        assert L.mcq_decode(*a) == H.mcq_decode_host(*a), a
    prep = [(None, 1.0, None, None, 8, 256, 64, None, None), (None, 1.0, None, None, 8, 4, 64, None, None),
            (1, 1.0, 1, None, 8, 256, 64, 1, None)]
    for a in prep:
        assert L.mcq_prepare(*a) == H.mcq_prepare_host(*a), a


@pytest.mark.parametrize("name", ["trained_d64_b8_p2", "trained_d64_b8_p1", "synth_d30_k32_n4", "k512_d32_n4"])
def test_twins_reproduce_the_reference_fixtures(name):
    fx = fixtures.load(name)
    H = ht.lib()
    D, K, N, B = fx["D"], fx["K"], fx["N"], fx["B"]
    st = fx["state"]
    # (the scale factors of the reference's run, pinned by the fixture: torch's fp32 exp differs in the last bit between CPUs)
    cs_, ls = st.scales_exp if getattr(st, "scales_exp", None) else (oracle.scale_exp(st["centers_scale"]), oracle.scale_exp(st["logits_scale"]))
    blob = ht.prepare(st, cs_)
    ws = np.zeros(max(1, H.mcq_encode_workspace_bytes_host(B, N, K, D)), np.uint8)
    x = np.ascontiguousarray(fx["x"], np.float32)
    it = fx["iters"][-1]
    idx = np.zeros((B, N), np.int64)
    assert H.mcq_encode_host(ht.ptr(x), B, ht.ptr(blob), ls, N, K, D, it, None, ht.ptr(idx), ht.ptr(ws), ws.size, None) == 0
    fixtures.check_codes(fx, it, idx, name)
    pack = 2 if K == 16 else 1
    by = np.zeros((B, N // pack), np.uint8)
    out = np.zeros((B, D), np.float32)
    if K <= 256:
        assert H.mcq_encode_host(ht.ptr(x), B, ht.ptr(blob), ls, N, K, D, it, ht.ptr(by), None, ht.ptr(ws), ws.size, None) == 0
        same = (idx == fx[f"codes_it{it}"]).all(axis=1)
        assert np.array_equal(by[same], fx[f"bytes_it{it}"][same])
        # decode of the reference's own bytes: head rows to 1e-5, every row by checksum
        ref_bytes = np.ascontiguousarray(fx[f"bytes_it{it}"])
        assert H.mcq_decode_host(ht.ptr(ref_bytes), 1, ref_bytes.shape[1], B, ht.ptr(blob), N, K, D, ht.ptr(out), None) == 0
    else:
        # codebooks of 512 / 1,024 entries: no byte form (rejected), int64 indexes decode
        assert H.mcq_encode_host(ht.ptr(x), B, ht.ptr(blob), ls, N, K, D, it, ht.ptr(by), None, ht.ptr(ws), ws.size, None) == -1      # MCQ_EINVAL
        ref_idx = np.ascontiguousarray(fx[f"codes_it{it}"].astype(np.int64))
        assert H.mcq_decode_host(ht.ptr(ref_idx), 8, N, B, ht.ptr(blob), N, K, D, ht.ptr(out), None) == 0
    np.testing.assert_allclose(out[:16], fx["decode_head"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out.astype(np.float64).sum(axis=1), fx["decode_rowsum"], rtol=1e-5, atol=1e-4)
    # refine_indexes twin: one pass from the 0-pass codes gives the 1-pass codes
    if 0 in fx["iters"] and 1 in fx["iters"]:
        start = np.ascontiguousarray(fx["codes_it0"].astype(np.int64))
        nxt = np.zeros_like(start)
        assert H.mcq_refine_indexes_host(ht.ptr(x), B, ht.ptr(blob), N, K, D, 1, ht.ptr(start), ht.ptr(nxt), ht.ptr(ws),
                                         ws.size, None) == 0
        agree = (nxt == fx["codes_it1"]).all(axis=1).mean()
        assert agree > 0.99, agree


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["trained_d64_b8_p2", "trained_d64_b4_p1", "synth_d40_k64_n8"])
def test_one_argument_list_through_both_libraries(name):
    import torch
    L, H = _device_lib(), ht.lib()
    fx = fixtures.load(name)
    D, K, N, B = fx["D"], fx["K"], fx["N"], fx["B"]
    st = fx["state"]
    cs, ls = oracle.scale_exp(st["centers_scale"]), oracle.scale_exp(st["logits_scale"])
    dev = torch.device("cuda:0")
    host = {"c": np.ascontiguousarray(st["centers"], np.float32), "w": np.ascontiguousarray(st["to_logits.weight"], np.float32),
            "b": np.ascontiguousarray(st["to_logits.bias"], np.float32), "x": np.ascontiguousarray(fx["x"], np.float32)}
    d = {k: torch.from_numpy(v).to(dev) for k, v in host.items()}
    pack = 2 if K == 16 else 1
    # -- prepare
    hblob = np.zeros(H.mcq_prepared_bytes_host(N, K, D), np.uint8)
    dblob = torch.zeros(L.mcq_prepared_bytes(N, K, D), dtype=torch.uint8, device=dev)
    args = lambda c, w, b, blob: (c, cs, w, b, N, K, D, blob, None)
    assert H.mcq_prepare_host(*args(ht.ptr(host["c"]), ht.ptr(host["w"]), ht.ptr(host["b"]), ht.ptr(hblob))) == 0
    assert L.mcq_prepare(*args(d["c"].data_ptr(), d["w"].data_ptr(), d["b"].data_ptr(), dblob.data_ptr())) == 0
    # -- encode (bytes), 5 passes
    hws = np.zeros(max(1, H.mcq_encode_workspace_bytes_host(B, N, K, D)), np.uint8)
    dws = torch.zeros(L.mcq_encode_workspace_bytes(B, N, K, D), dtype=torch.uint8, device=dev)
    hout = np.zeros((B, N // pack), np.uint8)
    dout = torch.zeros((B, N // pack), dtype=torch.uint8, device=dev)
    enc = lambda x, blob, out, ws, wsn: (x, B, blob, ls, N, K, D, 5, out, None, ws, wsn, None)
    assert H.mcq_encode_host(*enc(ht.ptr(host["x"]), ht.ptr(hblob), ht.ptr(hout), ht.ptr(hws), hws.size)) == 0
    assert L.mcq_encode(*enc(d["x"].data_ptr(), dblob.data_ptr(), dout.data_ptr(), dws.data_ptr(), dws.numel())) == 0
    torch.cuda.synchronize()
    assert np.array_equal(dout.cpu().numpy(), hout), "codes differ between mcq_encode and mcq_encode_host"
    # -- decode of those bytes
    hy = np.zeros((B, D), np.float32)
    dy = torch.zeros((B, D), dtype=torch.float32, device=dev)
    dec = lambda codes, blob, out: (codes, 1, N // pack, B, blob, N, K, D, out, None)
    assert H.mcq_decode_host(*dec(ht.ptr(hout), ht.ptr(hblob), ht.ptr(hy))) == 0
    assert L.mcq_decode(*dec(dout.data_ptr(), dblob.data_ptr(), dy.data_ptr())) == 0
    torch.cuda.synchronize()
    assert np.array_equal(dy.cpu().numpy().view(np.uint32), hy.view(np.uint32)), "decode not bit-identical"
    # -- logits (first 64 vectors)
    hl = np.zeros((64, N * K), np.float32)
    dl = torch.zeros((64, N * K), dtype=torch.float32, device=dev)
    lws = torch.zeros(L.mcq_logits_workspace_bytes(64, N, D), dtype=torch.uint8, device=dev)
    lg = lambda x, blob, out, ws, wsn: (x, 64, blob, ls, N, K, D, out, ws, wsn, None)
    assert H.mcq_logits_host(*lg(ht.ptr(host["x"]), ht.ptr(hblob), ht.ptr(hl), None, 0)) == 0
    assert L.mcq_logits(*lg(d["x"].data_ptr(), dblob.data_ptr(), dl.data_ptr(), lws.data_ptr(), lws.numel())) == 0
    torch.cuda.synchronize()
    assert np.array_equal(dl.cpu().numpy().view(np.uint32), hl.view(np.uint32)), "logits not bit-identical"
