"""padt_amd — MI355X-native (gfx950 HIP) implementation of PaDT's generate-with-Visual-Reference-Tokens hot path.

Drop-in names of the reference package (src/PaDT/__init__.py:1, src/PaDT/models/__init__.py:1-3):
``PaDTForConditionalGeneration``, ``PaDTDecoder``, ``VisonTextProcessingClass``, ``parseVRTintoCompletion``.
Importing the package needs only PyTorch; constructing a model loads libpadt_hip.so and fails loudly without it.
"""
import os as _os

# HIP maps streams onto HSA hardware queues (4 by default).  The throughput runner keeps the caller's stream, a prefill stream and two decode
# lanes busy; any further stream (post-processing, RCCL, the result exchange's side stream) would share a queue with one of them and wait for
# its backlog (≈90 ms stalls measured, tools/diag/to_rle_timing.py).  Only effective when set before the ROCm runtime initialises, i.e. when
# padt_amd is imported before the first torch.cuda call; an explicit setting of the user wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .config import PaDTConfig, VisionConfig, padt_pro_3b, padt_pro_7b, small_test_config
from .processor import VisonTextProcessingClass, parseVRTintoCompletion

__all__ = ["PaDTForConditionalGeneration", "PaDTDecoder", "VisonTextProcessingClass", "parseVRTintoCompletion",
           "PaDTConfig", "VisionConfig", "padt_pro_3b", "padt_pro_7b", "small_test_config"]


def __getattr__(name):
    if name == "PaDTForConditionalGeneration":
        from .modeling import PaDTForConditionalGeneration
        return PaDTForConditionalGeneration
    if name == "PaDTDecoder":
        from .decoder import PaDTDecoder
        return PaDTDecoder
    raise AttributeError(name)
