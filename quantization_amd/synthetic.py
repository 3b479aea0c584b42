"""Deterministic inputs and synthetic quantizer states: the bench workload (bench.py), the smoke check and the
test fixtures (tests/golden/gen.py re-exports this module; make_golden.py runs where the reference is importable,
the tests run anywhere).  numpy-only, elementwise-only arithmetic (no BLAS) so the arrays
are bit-identical on every machine with this image; fixtures store a checksum of
each regenerated array so a mismatch is diagnosed as such.
"""
import numpy as np


def checksum(a) -> float:
    a = np.ascontiguousarray(a)
    w = (np.arange(a.size, dtype=np.float64) % 251.0) + 1.0
    return float((a.astype(np.float64).ravel() * w).sum())


def make_x(seed: int, B: int, D: int) -> np.ndarray:
    """Correlated Gaussian frames (B, D) fp32: neighbouring features share a component."""
    rs = np.random.RandomState(seed)
    base = rs.standard_normal((B, D)).astype(np.float32)
    s = (0.6 + 0.8 * rs.random_sample(D)).astype(np.float32)
    t = (0.5 * rs.random_sample(D)).astype(np.float32)
    return (base * s + np.roll(base, 1, axis=1) * t).astype(np.float32)


def make_gaussian(seed: int, B: int, D: int) -> np.ndarray:
    """x ~ N(0,1) fp32, the bench workload."""
    return np.random.RandomState(seed).standard_normal((B, D)).astype(np.float32)


def make_kind(kind: str, seed: int, B: int, D: int) -> np.ndarray:
    """Frames of a named distribution (the `x_kind` of a fixture).  Beyond the zero-mean unit-scale sets: frames with a
    large common offset ("mean10", "mean100"), one dominant feature ("outlier300": feature 0 scaled by 300) and heavy
    tails ("student2": Student-t with two degrees of freedom) -- what log-mel / self-supervised features look like and
    where the table form's cancellation terms are largest."""
    if kind == "gaussian":
        return make_gaussian(seed, B, D)
    if kind == "make_x":
        return make_x(seed, B, D)
    if kind in ("mean10", "mean100"):
        return (make_x(seed, B, D) + np.float32(10.0 if kind == "mean10" else 100.0)).astype(np.float32)
    if kind == "outlier300":
        x = make_x(seed, B, D)
        x[:, 0] = (x[:, 0] * np.float32(300.0)).astype(np.float32)
        return x
    if kind == "student2":
        return np.random.RandomState(seed).standard_t(2.0, size=(B, D)).astype(np.float32)
    raise ValueError(kind)


def synthetic_state(seed: int, D: int, K: int, N: int, centers_scale=0.02, logits_scale=-0.01):
    """A seeded quantizer state with a sensible initial guess: to_logits scores a
    codeword by alpha * (x.c - |c|^2 / 2), i.e. nearest-codeword per codebook."""
    rs = np.random.RandomState(seed)
    centers = (rs.standard_normal((N, K, D)) * (1.0 / np.sqrt(N))).astype(np.float32)
    alpha = np.float32(0.5)
    weight = (centers.reshape(N * K, D) * alpha).astype(np.float32)
    sq = (centers.astype(np.float32) ** 2).sum(axis=2, dtype=np.float32).reshape(N * K)
    bias = (-0.5 * alpha * sq).astype(np.float32)
    bias = (bias + 0.01 * rs.standard_normal(N * K).astype(np.float32)).astype(np.float32)
    return {
        "centers": centers,
        "centers_scale": np.float32(centers_scale),
        "logits_scale": np.float32(logits_scale),
        "to_logits.weight": weight,
        "to_logits.bias": bias,
    }
