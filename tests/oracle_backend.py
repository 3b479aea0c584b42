"""Test-only: run quantization_amd's host logic on CPU tensors by routing the two kernel
entry points through the CPU oracle.  The product has no such path (it raises on CPU
tensors); tests use this to check trainer/host logic without a GPU."""
import contextlib

import numpy as np
import torch

from oracle.oracle import OracleQuantizer


@contextlib.contextmanager
def oracle_kernels():
    from quantization_amd.quantizer import Quantizer

    def _oracle(self):
        sd = {k: v.detach().cpu() for k, v in self.state_dict().items()}
        return OracleQuantizer.from_state_dict(sd)

    def _search(self, x2d, iters, as_bytes):
        self._check_domain()
        out = _oracle(self).encode(x2d.detach().cpu().numpy().astype(np.float32), iters, as_bytes)
        return torch.from_numpy(np.ascontiguousarray(out)).to(x2d.device)

    def _decode_kernel(self, flat):
        y = _oracle(self).decode(flat.detach().cpu().numpy())
        return torch.from_numpy(y).to(flat.device)

    saved = (Quantizer._search, Quantizer._decode_kernel)
    Quantizer._search, Quantizer._decode_kernel = _search, _decode_kernel
    try:
        yield
    finally:
        Quantizer._search, Quantizer._decode_kernel = saved
