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
# THIS IS A PROCESS-WIDE SIDE EFFECT OF `import padt_amd` (it changes HSA queue allocation for every GPU user of the process): PADT_NO_ENV=1
# turns it off; `hw_queue_note` says when the default came too late to matter (the runtime was already up) — pipeline.PipelinedRunner warns then.
import torch as _torch

hw_queue_note = None
if _os.environ.get("PADT_NO_ENV") != "1":
    if "GPU_MAX_HW_QUEUES" not in _os.environ and _torch.cuda.is_initialized():
        hw_queue_note = ("padt_amd was imported after the ROCm runtime initialised: GPU_MAX_HW_QUEUES=8 could not take effect (HIP keeps its 4 hardware "
                         "queues; a fifth busy stream will share one). Import padt_amd — or export GPU_MAX_HW_QUEUES=8 — before the first torch.cuda call.")
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
