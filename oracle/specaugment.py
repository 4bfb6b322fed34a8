"""SpecAugment oracle (integer-defined masks from Philox4x32-10).

The reference's ``SpecAugment`` source is ABSENT; only its
constructor and shape contract are pinned by
``tests/test_training_pipeline.py:252-262`` (``SpecAugment(freq_mask_param=15,
time_mask_param=35, n_freq_masks=2, n_time_masks=2)`` on ``(1,64,50)`` -> same
shape) and the application probabilities by ``src/config/defaults.py:90-91``.
The index law below is the BUILD'S OWN SPEC -- **parity
unpinned** w.r.t. the reference; the HIP kernel must match it bit-exactly.

For sample b and mask k (freq masks k=0..n_f-1, then time masks k=n_f..n_f+n_t-1):
    r      = philox4x32_10(ctr=(step_lo, step_hi, b, TAG_SPECAUG<<24 | k), key=seed)
    dim    = F (freq) or T (time);  pmax = min(param, dim)
    w      = r[0] mod (pmax+1)
    s      = r[1] mod (dim - w + 1)
    apply  = uint64(r[2]) < floor(prob * 2**32)
    index row = (s, w if apply else 0);   x[b, :, s:s+w, :] = 0  (resp. [..., s:s+w])
"""
import numpy as np
from .philox import philox4x32_10, make_ctr, make_key, prob_threshold, TAG_SPECAUG


def specaug_indices(B, F, T, freq_mask_param, time_mask_param, n_freq_masks, n_time_masks,
                    freq_mask_prob=1.0, time_mask_prob=1.0, seed=0, step=0, sample_offset=0):
    """-> int32 (B, n_f+n_t, 2) rows (start, width)."""
    K = n_freq_masks + n_time_masks
    out = np.zeros((B, K, 2), dtype=np.int32)
    if K == 0:
        return out
    b = (np.arange(B, dtype=np.uint64) + np.uint64(sample_offset))[:, None]
    k = np.arange(K, dtype=np.uint64)[None, :]
    r = philox4x32_10(make_ctr(step, b, TAG_SPECAUG, k), make_key(seed)).astype(np.uint64)
    is_f = (np.arange(K) < n_freq_masks)[None, :]
    dim = np.where(is_f, F, T).astype(np.uint64)
    param = np.where(is_f, freq_mask_param, time_mask_param).astype(np.uint64)
    pmax = np.minimum(param, dim)
    w = r[..., 0] % (pmax + np.uint64(1))
    s = r[..., 1] % (dim - w + np.uint64(1))
    thr = np.where(is_f, prob_threshold(freq_mask_prob), prob_threshold(time_mask_prob)).astype(np.uint64)
    apply = r[..., 2] < thr
    out[..., 0] = s.astype(np.int32)
    out[..., 1] = np.where(apply, w, 0).astype(np.int32)
    return out


def specaug_apply(x, idx, n_freq_masks):
    """x (B,1,F,T) -> masked copy, idx from specaug_indices."""
    x = np.array(x, copy=True)
    B = x.shape[0]
    for b in range(B):
        for k in range(idx.shape[1]):
            s, w = int(idx[b, k, 0]), int(idx[b, k, 1])
            if w == 0:
                continue
            if k < n_freq_masks:
                x[b, :, s:s + w, :] = 0
            else:
                x[b, :, :, s:s + w] = 0
    return x
