"""MI355X-native wakeword training inner loop (drop-in for the reference's
``src/training`` Trainer + ``src/models`` factory path).

The package layout mirrors the reference's ``src/`` tree for the hot path only:
``config`` (dataclasses the Trainer reads), ``data`` (FeatureExtractor / SpecAugment),
``models`` (create_model / create_loss_function), ``training`` (Trainer and its glue).
All device work goes through ``_native`` -> ``csrc/libwwhip.so`` (hand-written HIP, gfx950).
"""
__version__ = "0.1.0"
