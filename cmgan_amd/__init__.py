"""cmgan_amd - MI355X-native CMGAN generator forward path (hand-written HIP behind a C ABI).

    from cmgan_amd import TSCNet, ConformerBlock           # mirrors of the reference classes
    from cmgan_amd.utils import power_compress, power_uncompress, stft_compress, uncompress_istft
    from cmgan_amd.evaluation import enhance_one_track, evaluation   # src/evaluation.py
    from cmgan_amd.metrics import compute_metrics                    # src/tools/compute_metrics.py (CPU, numpy)

Importing the package is cheap and GPU-free; the shared library is loaded (and must
exist) the moment an Engine / model is constructed.
"""
from .conformer import ConformerBlock        # noqa: F401
from .generator import TSCNet                # noqa: F401

__all__ = ["TSCNet", "ConformerBlock"]
