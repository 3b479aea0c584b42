"""quantization_amd -- MI355X-native multi-codebook vector quantizer.

Drop-in for the hot path of danpovey/quantization (`Quantizer.encode/decode`,
`QuantizerTrainer.step`): same class names, arguments and state-dict layout as
/root/reference/quantization/__init__.py:1-2 exports for this path.
"""
from .quantizer import Quantizer  # noqa: F401
from .trainer import QuantizerTrainer  # noqa: F401
from .prediction import JointCodebookLoss  # noqa: F401   (the consumer of the codes, quantization/__init__.py:4)
from .hdf5_data import read_hdf5_data  # noqa: F401      (the data helper, quantization/__init__.py:3)

__all__ = ["Quantizer", "QuantizerTrainer", "JointCodebookLoss", "read_hdf5_data"]
