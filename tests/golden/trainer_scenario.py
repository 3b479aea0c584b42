"""Inputs of the trainer scenario fixture (trainer_scenario_d256_b4.npz): the reference's integration scenario
(test_quantization.py:11-48) feeds `model(x) + 0.05 * x` with x ~ N(0, 1) and model = Linear-ReLU-Linear-ReLU-
LayerNorm-Linear with random weights.  Here the same network is evaluated in numpy float64 with seeded weights and
rounded to fp32, so generator and test produce the same frames on any machine (up to rare 1-ulp roundings)."""
import numpy as np

DIM, BYTES, BATCH, P1, P2, SEED = 256, 4, 600, 500, 500, 1

_W = None


def _weights():
    global _W
    if _W is None:
        rs = np.random.RandomState(4242)
        bound = 1.0 / np.sqrt(DIM)                      # nn.Linear's default init range
        _W = [(rs.uniform(-bound, bound, (DIM, DIM)), rs.uniform(-bound, bound, DIM)) for _ in range(3)]
    return _W


def scenario_batch(it: int) -> np.ndarray:
    (w1, b1), (w2, b2), (w3, b3) = _weights()
    x = np.random.RandomState(31337 + it).standard_normal((BATCH, DIM))
    h = np.maximum(x @ w1.T + b1, 0.0)
    h = np.maximum(h @ w2.T + b2, 0.0)
    h = (h - h.mean(axis=1, keepdims=True)) / np.sqrt(h.var(axis=1, keepdims=True) + 1e-5)     # LayerNorm
    y = h @ w3.T + b3 + 0.05 * x
    return y.astype(np.float32)
