"""MI355X-native wakeword training inner loop (drop-in for the reference's
``src/training`` Trainer + ``src/models`` factory path).

The package layout mirrors the reference's ``src/`` tree for the hot path only:
``config`` (dataclasses the Trainer reads), ``data`` (FeatureExtractor / SpecAugment),
``models`` (create_model / create_loss_function), ``training`` (Trainer and its glue).
All device work goes through ``_native`` -> ``csrc/libwwhip.so`` (hand-written HIP, gfx950).
"""
import os as _os

# The step keeps up to three HIP streams busy (conv stack, input stage one batch ahead, RCCL).  The HIP runtime multiplexes
# streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and RCCL's communicator takes some: with the default, the
# input-stage stream and the main stream end up on ONE queue in data-parallel runs and the log-mel kernel stops overlapping
# the conv kernels (measured: 1.48 vs 1.39 ms per step, gpurun_out r02c).  Read when the HIP runtime initialises, so it is
# set at import time; an explicit setting in the environment wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.2.0"
