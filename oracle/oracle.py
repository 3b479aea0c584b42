"""ctypes front-end of the CPU oracle (oracle/mcq_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by quantization_amd/.  See the header of
mcq_oracle.c for what it restates and how it is pinned.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmcq_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile mcq_oracle.c with gcc (oracle/Makefile); returns the .so path."""
    newest = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("mcq_oracle.c", "mcq_host.c", "mcq_host.h"))
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < newest:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "_build/libmcq_oracle.so"])
    return _SO


def _load():
    global _lib
    if _lib is not None:
        return _lib
    lib = ctypes.CDLL(build())
    f32p = ctypes.POINTER(ctypes.c_float)
    u16p = ctypes.POINTER(ctypes.c_uint16)      # indexes / entries: uint16 in the C oracle (codebook_size up to 1,024)
    i32p = ctypes.POINTER(ctypes.c_int)
    lib.mcq_oracle_create.restype = ctypes.c_void_p
    lib.mcq_oracle_create.argtypes = [f32p, ctypes.c_float, f32p, f32p, ctypes.c_float,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.mcq_oracle_free.argtypes = [ctypes.c_void_p]
    lib.mcq_oracle_get_centers.argtypes = [ctypes.c_void_p, f32p, f32p]
    lib.mcq_oracle_compute_indexes.restype = ctypes.c_int
    lib.mcq_oracle_compute_indexes.argtypes = [ctypes.c_void_p, f32p, ctypes.c_long, ctypes.c_int, u16p,
                                               ctypes.c_int]
    lib.mcq_oracle_refine_trace.restype = ctypes.c_int
    lib.mcq_oracle_refine_trace.argtypes = [ctypes.c_void_p, f32p, u16p, f32p, f32p, f32p, f32p, i32p, f32p,
                                            f32p]
    lib.mcq_oracle_logits.restype = ctypes.c_int
    lib.mcq_oracle_logits.argtypes = [ctypes.c_void_p, f32p, ctypes.c_long, f32p]
    lib.mcq_oracle_decode.argtypes = [ctypes.c_void_p, u16p, ctypes.c_long, f32p]
    lib.mcq_oracle_ladder.restype = ctypes.c_int
    lib.mcq_oracle_ladder.argtypes = [ctypes.c_int, ctypes.c_int, i32p, i32p, i32p]
    lib.mcq_oracle_dp.restype = ctypes.c_int
    lib.mcq_oracle_dp.argtypes = [ctypes.c_int]
    _lib = lib
    return lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, ty):
    return a.ctypes.data_as(ctypes.POINTER(ty))


def scale_exp(scale: float) -> float:
    """exp(scale * 10) in fp32, the way get_centers()/_logits() form it
    (quantization/quantization.py:78, :278): an fp32 multiply, then torch's fp32 exp."""
    import torch
    return float((torch.tensor(float(scale), dtype=torch.float32) * 10.0).exp().item())


def ladder(N: int, K: int):
    """(first_keep, [(Kin, Kout), ...]) -- the prune/combine ladder of _refine_indexes."""
    lib = _load()
    first = ctypes.c_int(0)
    kin = (ctypes.c_int * 16)()
    kout = (ctypes.c_int * 16)()
    n = lib.mcq_oracle_ladder(N, K, ctypes.byref(first), kin, kout)
    return first.value, [(kin[i], kout[i]) for i in range(n)]


class OracleQuantizer:
    """CPU oracle for one quantizer state (numpy in, numpy out)."""

    def __init__(self, centers, centers_scale, weight=None, bias=None, logits_scale=0.0, scales_exp=None):
        """scales_exp: (exp(10 centers_scale), exp(10 logits_scale)) as fp32 values to use instead of this host's torch exp --
        the factors a fixture's reference run computed with (torch's fp32 exp differs in the last bit between CPUs)."""
        lib = _load()
        centers = _f32(centers)
        self.N, self.K, self.D = centers.shape
        if scales_exp is not None:
            self.cscale_exp, self.lscale_exp = float(np.float32(scales_exp[0])), float(np.float32(scales_exp[1]))
        else:
            self.cscale_exp = scale_exp(centers_scale)
            self.lscale_exp = scale_exp(logits_scale)
        f = ctypes.c_float
        if weight is not None:
            weight = _f32(weight)
            bias = _f32(bias)
            assert weight.shape == (self.N * self.K, self.D) and bias.shape == (self.N * self.K,)
            wp, bp = _ptr(weight, f), _ptr(bias, f)
        else:
            wp = bp = None
        self._h = lib.mcq_oracle_create(_ptr(centers, f), self.cscale_exp, wp, bp, self.lscale_exp,
                                        self.N, self.K, self.D)
        self._lib = lib

    @classmethod
    def from_state_dict(cls, sd):
        g = lambda k: np.asarray(sd[k].detach().cpu().numpy() if hasattr(sd[k], "detach") else sd[k])
        return cls(g("centers"), float(g("centers_scale")), g("to_logits.weight"), g("to_logits.bias"),
                   float(g("logits_scale")), scales_exp=getattr(sd, "scales_exp", None))

    def __del__(self):
        try:
            self._lib.mcq_oracle_free(self._h)
        except Exception:
            pass

    def centers(self):
        """(scaled centers (N,K,D), their sumsq (N,K))"""
        C = np.empty((self.N, self.K, self.D), np.float32)
        Q = np.empty((self.N, self.K), np.float32)
        self._lib.mcq_oracle_get_centers(self._h, _ptr(C, ctypes.c_float), _ptr(Q, ctypes.c_float))
        return C, Q

    def compute_indexes(self, x, refine_indexes_iters=3, nthreads=0):
        x = _f32(x).reshape(-1, self.D)
        idx = np.empty((x.shape[0], self.N), np.uint16)
        rc = self._lib.mcq_oracle_compute_indexes(self._h, _ptr(x, ctypes.c_float), x.shape[0],
                                                  int(refine_indexes_iters), _ptr(idx, ctypes.c_uint16),
                                                  int(nthreads))
        assert rc == 0, f"oracle error {rc}"
        return idx.astype(np.uint8) if self.K <= 256 else idx

    def encode(self, x, refine_indexes_iters=5, as_bytes=True, nthreads=0):
        """Quantizer.encode (quantization/quantization.py:244-275) incl. nibble packing."""
        x = np.asarray(x)
        idx = self.compute_indexes(x, refine_indexes_iters, nthreads).astype(np.int64)
        if as_bytes:
            K = self.K
            while K * K <= 256:
                idx = idx[:, ::2] + K * idx[:, 1::2]
                K = K * K
            assert K <= 256                                            # quantization.py:271
            idx = idx.astype(np.uint8)
        return idx.reshape(*x.shape[:-1], -1)

    def logits(self, x):
        x = _f32(x).reshape(-1, self.D)
        out = np.empty((x.shape[0], self.N * self.K), np.float32)
        rc = self._lib.mcq_oracle_logits(self._h, _ptr(x, ctypes.c_float), x.shape[0], _ptr(out, ctypes.c_float))
        assert rc == 0
        return out

    def separate_indexes(self, codes):
        """_maybe_separate_indexes (quantization/quantization.py:551-573)."""
        codes = np.asarray(codes).reshape(-1, np.asarray(codes).shape[-1]).astype(np.int64)
        n = codes.shape[-1]
        if n != self.N:
            r = self.N // n
            assert r in (2, 4, 8, 16) and self.N == n * r
            codes = (codes[:, :, None] // (self.K ** np.arange(r))[None, None, :]) % self.K
            codes = codes.reshape(-1, self.N)
        return codes

    def decode(self, codes):
        codes = np.asarray(codes)
        lead = codes.shape[:-1]
        idx = np.ascontiguousarray(self.separate_indexes(codes).astype(np.uint16))
        out = np.empty((idx.shape[0], self.D), np.float32)
        self._lib.mcq_oracle_decode(self._h, _ptr(idx, ctypes.c_uint16), idx.shape[0], _ptr(out, ctypes.c_float))
        return out.reshape(*lead, self.D)

    def refine_trace(self, x1, idx1):
        """One refinement pass on one vector; returns a dict of per-stage intermediates."""
        first, lad = ladder(self.N, self.K)
        n_sel = self.N * first
        n_comb = 0
        groups = self.N
        for kin, kout in lad:
            groups //= 2
            n_comb += groups * kin * kin
            n_sel += groups * kout
        Dp = self._lib.mcq_oracle_dp(self.D)
        x1 = _f32(x1).reshape(self.D)
        idx = np.ascontiguousarray(idx1, dtype=np.uint16).reshape(self.N).copy()
        t = dict(xerr=np.zeros(Dp, np.float32), E=np.zeros(1, np.float32), R=np.zeros(self.N, np.float32),
                 S0=np.zeros((self.N, self.K), np.float32), sel_pos=np.zeros(n_sel, np.int32),
                 sel_val=np.zeros(n_sel, np.float32), comb=np.zeros(max(n_comb, 1), np.float32))
        f = ctypes.c_float
        self._lib.mcq_oracle_refine_trace(self._h, _ptr(x1, f), _ptr(idx, ctypes.c_uint16), _ptr(t["xerr"], f),
                                          _ptr(t["E"], f), _ptr(t["R"], f), _ptr(t["S0"], f),
                                          _ptr(t["sel_pos"], ctypes.c_int), _ptr(t["sel_val"], f),
                                          _ptr(t["comb"], f))
        t["idx"] = idx
        return t
