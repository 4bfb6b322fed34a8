"""CPU oracle for the wakeword training inner loop -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product package
(``wakeword_trainer_home_amd``) never imports this package and fails loudly
when its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * losses / metrics / schedulers / Trainer step trace: PINNED against fixtures
    generated in the build container by importing the reference's own modules
    (tests/golden/make_golden.py -> tests/golden/*.npz|json).
  * Philox4x32-10: PINNED against the Random123 known-answer vectors.
  * log-mel / MFCC / SpecAugment / cnn_small: the reference ships no source
    for ``src/data`` and has no ``cnn_small``, so these
    follow the build's own written spec -- **parity unpinned** with respect to
    the reference; they are pinned only against independent formulations
    (float64 DFT by definition, torch.stft, torch.nn).
"""
