"""Host-side checks that need no GPU: the C-ABI library loads and exports every
symbol of include/mcq.h; size queries; argument validation that precedes any launch."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    import __graft_entry__ as g
    g.build()
    from quantization_amd import _lib
    return _lib


def test_header_symbols_are_exported():
    _lib_mod = _lib()
    L = _lib_mod.lib()
    hdr = open(os.path.join(ROOT, "include", "mcq.h")).read()
    declared = set(re.findall(r"\b(mcq_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib_mod.SYMBOLS), declared ^ set(_lib_mod.SYMBOLS)
    for s in declared:
        assert hasattr(L, s), s
    assert L.mcq_abi_version() == 7


def test_size_queries():
    L = _lib().lib()
    assert L.mcq_padded_dim(512) == 512 and L.mcq_padded_dim(40) == 48 and L.mcq_padded_dim(1) == 16
    # scaled centers + sumsq + padded weight + bias
    assert L.mcq_prepared_bytes(8, 256, 512) >= 2 * 8 * 256 * 512 * 4 + 2 * 8 * 256 * 4
    a = L.mcq_encode_workspace_bytes(1000, 8, 256, 512)
    b = L.mcq_encode_workspace_bytes(65536, 8, 256, 512)
    c = L.mcq_encode_workspace_bytes(10 ** 7, 8, 256, 512)
    assert a < b and b == c          # the batch is chunked: the workspace stops growing


def test_argument_validation_without_launch():
    m = _lib()
    L = m.lib()
    # unsupported domain and bad arguments are rejected before anything touches the device
    assert L.mcq_encode(None, 4, None, 1.0, 8, 8, 64, 1, None, None, None, 0, None) == m.MCQ_EUNSUPPORTED
    assert L.mcq_encode(None, 4, None, 1.0, 8, 2048, 64, 1, None, None, None, 0, None) == m.MCQ_EUNSUPPORTED
    assert L.mcq_encode(None, 4, None, 1.0, 32, 1024, 64, 1, None, None, None, 0, None) == m.MCQ_EUNSUPPORTED    # Gram matrix past 16,384 rows
    assert L.mcq_encode(None, 4, None, 1.0, 8, 512, 64, 1, None, None, None, 0, None) == m.MCQ_EINVAL            # supported shape, no output array
    assert L.mcq_encode(None, 4, None, 1.0, 3, 256, 64, 1, None, None, None, 0, None) == m.MCQ_EINVAL
    assert L.mcq_encode(None, -1, None, 1.0, 8, 256, 64, 1, None, None, None, 0, None) == m.MCQ_EINVAL
    assert L.mcq_decode(None, 2, 8, 4, None, 8, 256, 64, None, None) == m.MCQ_EINVAL
    assert L.mcq_decode(None, 1, 3, 4, None, 8, 256, 64, None, None) == m.MCQ_EINVAL
    assert L.mcq_prepare(None, 1.0, None, None, 8, 256, 64, None, None) == m.MCQ_EINVAL
    # the trainer's entry points stay at K <= 256 (include/mcq.h): wider codebooks are UNSUPPORTED there, not "invalid"
    assert L.mcq_logits_refine(None, 4, None, 1.0, 4, 512, 64, 1, None, None, None, 0, None, 0) == m.MCQ_EUNSUPPORTED
    assert L.mcq_logits_refine_codes(None, 4, None, 1.0, 4, 512, 64, 1, None, None, None, None, 0, None, 0) == m.MCQ_EUNSUPPORTED
    assert L.mcq_decode_backward_u8(None, None, 4, 4, 512, 64, None, None) == m.MCQ_EUNSUPPORTED
    assert L.mcq_decode_backward_u8_ex(None, None, 4, 4, 512, 64, None, None, None, 1.0, None, None, None) == m.MCQ_EUNSUPPORTED
    assert L.mcq_decode_backward_u8(None, None, 4, 4, 256, 64, None, None) == m.MCQ_EINVAL
    assert L.mcq_loss_fwd(None, None, 4, 4, 512, None, None, None, None, None, 0, None) == m.MCQ_EUNSUPPORTED
    assert L.mcq_profile_encode(None, 4, None, 1.0, 8, 256, 64, 1, None, 0, None, None, None, 0) == m.MCQ_EINVAL     # no output array


def test_module_api_surface_and_state_dict():
    import torch
    from quantization_amd import Quantizer, QuantizerTrainer
    q = Quantizer(dim=64, codebook_size=256, num_codebooks=4)
    sd = q.state_dict()
    shapes = {k: (tuple(v.shape), v.dtype) for k, v in sd.items()}
    assert shapes == {
        "centers": ((4, 256, 64), torch.float32), "logits_scale": ((), torch.float32),
        "centers_scale": ((), torch.float32), "id_buf": ((8,), torch.uint8),
        "to_logits.weight": ((1024, 64), torch.float32), "to_logits.bias": ((1024,), torch.float32)}
    assert torch.equal(q.centers.reshape(1024, 64), q.to_logits.weight)      # quantization.py:41-42
    q2 = Quantizer(64, 256, 4)
    assert q2.get_id() != q.get_id() and len(q.get_id()) == 8
    q2.load_state_dict(sd)
    assert q2.get_id() == q.get_id()                                         # test_train_hdf5.py:54
    assert "codebook_size=256" in q.show_init_invocation()
    for bad in (dict(dim=8, codebook_size=12, num_codebooks=2), dict(dim=8, codebook_size=16, num_codebooks=3)):
        with pytest.raises(AssertionError):
            Quantizer(**bad)
    with pytest.raises(AssertionError):
        QuantizerTrainer(dim=8, bytes_per_frame=3, device=torch.device("cpu"))
    # no CPU fallback: a CPU tensor is an error, not a slow path
    with pytest.raises(Exception):
        q.encode(torch.zeros(2, 64))


def test_product_quantizer_matches_definition():
    import torch
    from quantization_amd import Quantizer
    torch.manual_seed(0)
    q = Quantizer(8, 16, 4)
    with torch.no_grad():
        q.logits_scale.fill_(0.3)
        q.centers_scale.fill_(-0.2)
        q.centers.normal_()
    p = q.get_product_quantizer()
    assert (p.codebook_size, p.num_codebooks) == (256, 2)
    assert float(p.logits_scale) == float(q.logits_scale) and float(p.centers_scale) == float(q.centers_scale)
    for c in range(2):
        for k1 in (0, 5, 15):
            for k2 in (0, 7, 15):
                ko = k1 * 16 + k2                                            # quantization.py:107
                assert torch.equal(p.centers[c, ko], q.centers[2 * c, k1] + q.centers[2 * c + 1, k2])
                assert torch.equal(p.to_logits.weight[256 * c + ko],
                                   q.to_logits.weight[16 * 2 * c + k1] + q.to_logits.weight[16 * (2 * c + 1) + k2])
                assert p.to_logits.bias[256 * c + ko] == q.to_logits.bias[16 * 2 * c + k1] + q.to_logits.bias[16 * (2 * c + 1) + k2]


def test_workspace_and_prepared_sizes_over_the_domain():
    """mcq_encode_workspace_bytes / mcq_prepared_bytes for every (K, N) of the domain: positive, monotone in B up to the
    default chunk, below 4 GB at the largest shapes, and 'slack only' outside the domain."""
    from quantization_amd import _lib as m
    L = m.lib()
    for K in (16, 32, 64, 128, 256):
        for N in (1, 2, 4, 8, 16, 32, 64, 128):
            ok = N <= 64
            big = L.mcq_encode_workspace_bytes(10 ** 7, N, K, 512)
            small = L.mcq_encode_workspace_bytes(100, N, K, 512)
            if not ok:
                assert big == small
                continue
            assert 0 < small < big <= 4 * 2 ** 30, (K, N, small, big)
            nk = N * K
            assert L.mcq_prepared_bytes(N, K, 512) >= 4 * (2 * nk * 512 + nk * nk)
