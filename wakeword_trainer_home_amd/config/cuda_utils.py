"""GPU gate.  The reference refuses to train without CUDA: ``enforce_cuda()`` prints and
``sys.exit(1)`` (``src/config/cuda_utils.py:210-223``), called first thing in ``Trainer.__init__``
(``src/training/trainer.py:63``).  Same contract here; on ROCm ``torch.cuda.is_available()`` is the
HIP device check, and the device must be gfx950 for the native kernels."""
import sys

import torch


def validate():
    if not torch.cuda.is_available():
        return False, ("CUDA/HIP device is not available. GPU is MANDATORY for this training path "
                       "(the HIP hot path has no CPU fallback).")
    n = torch.cuda.device_count()
    if n == 0:
        return False, "No GPUs detected."
    return True, f"GPU validation passed: {n} device(s), {torch.cuda.get_device_name(0)}"


def enforce_cuda():
    ok, msg = validate()
    print(msg)
    if not ok:
        print("\n" + "=" * 60 + "\nCUDA VALIDATION FAILED - EXITING\n" + "=" * 60)
        sys.exit(1)
    return True
