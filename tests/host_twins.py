"""ctypes view of the host twins (oracle/mcq_host.h) with the argtypes of quantization_amd/_lib.py, so a
test can push ONE argument list through the device entry point and its host twin.  Test helper only."""
import ctypes

import numpy as np

from oracle import oracle as _oracle

TWINS = ("mcq_prepared_bytes", "mcq_prepare", "mcq_encode_workspace_bytes", "mcq_encode", "mcq_refine_indexes",
         "mcq_decode", "mcq_logits")
_h = None


def lib():
    global _h
    if _h is not None:
        return _h
    H = ctypes.CDLL(_oracle.build())
    vp, f32, i32, i64, sz = ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_long, ctypes.c_size_t
    H.mcq_prepared_bytes_host.restype = sz
    H.mcq_prepared_bytes_host.argtypes = [i32, i32, i32]
    H.mcq_prepare_host.argtypes = [vp, f32, vp, vp, i32, i32, i32, vp, vp]
    H.mcq_encode_workspace_bytes_host.restype = sz
    H.mcq_encode_workspace_bytes_host.argtypes = [i64, i32, i32, i32]
    H.mcq_encode_host.argtypes = [vp, i64, vp, f32, i32, i32, i32, i32, vp, vp, vp, sz, vp]
    H.mcq_refine_indexes_host.argtypes = [vp, i64, vp, i32, i32, i32, i32, vp, vp, vp, sz, vp]
    H.mcq_decode_host.argtypes = [vp, i32, i32, i64, vp, i32, i32, i32, vp, vp]
    H.mcq_logits_host.argtypes = [vp, i64, vp, f32, i32, i32, i32, vp, vp, ctypes.c_size_t, vp]
    _h = H
    return H


def ptr(a):
    return a.ctypes.data if a is not None else None


def prepare(state, cscale_exp):
    """host `prepared` blob (numpy uint8) for a state dict of numpy arrays"""
    H = lib()
    N, K, D = state["centers"].shape
    blob = np.zeros(H.mcq_prepared_bytes_host(N, K, D), np.uint8)
    c = np.ascontiguousarray(state["centers"], np.float32)
    w = np.ascontiguousarray(state["to_logits.weight"], np.float32)
    b = np.ascontiguousarray(state["to_logits.bias"], np.float32)
    rc = H.mcq_prepare_host(ptr(c), cscale_exp, ptr(w), ptr(b), N, K, D, ptr(blob), None)
    assert rc == 0, rc
    return blob
