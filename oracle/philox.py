"""Philox4x32-10 counter-based RNG (Salmon et al., SC'11; Random123 v1.14).

Oracle for the build-defined SpecAugment / dropout randomness (SURVEY.md §8a-S:
the reference's ``src/data/augmentation.py`` is absent, so the index spec is the
build's own).  Pure integer arithmetic => bit-exact between numpy and HIP.
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)
SH = np.uint64(32)


def philox4x32_10(ctr, key):
    """ctr: (..., 4) uint32-valued, key: (..., 2) uint32-valued -> (..., 4) uint32."""
    c = [np.asarray(ctr)[..., i].astype(np.uint64) & MASK for i in range(4)]
    key = np.asarray(key)
    k0 = key[..., 0].astype(np.uint64) & MASK
    k1 = key[..., 1].astype(np.uint64) & MASK
    for r in range(10):
        p0 = M0 * c[0]
        p1 = M1 * c[2]
        hi0, lo0 = p0 >> SH, p0 & MASK
        hi1, lo1 = p1 >> SH, p1 & MASK
        c = [(hi1 ^ c[1] ^ k0) & MASK, lo1, (hi0 ^ c[3] ^ k1) & MASK, lo0]
        k0 = (k0 + np.uint64(W0)) & MASK
        k1 = (k1 + np.uint64(W1)) & MASK
    return np.stack(c, axis=-1).astype(np.uint32)


def prob_threshold(p: float) -> int:
    """Integer threshold T such that (u32 draw < T) happens with probability p.

    T = floor(float32(p) * 2**32) clamped to [0, 2**32]; compared as uint64 so p=1.0
    is always true.  p crosses the C-ABI as a float32 (ww_specaug_cfg / dropout_p), so the
    spec takes the float32 value of p; the product itself is exact in double
    (ww_prob_threshold on the C side).
    """
    t = int(np.floor(float(np.float32(p)) * 4294967296.0))
    return max(0, min(t, 1 << 32))


# stream tags placed in the top byte of ctr[3]
TAG_SPECAUG = 0
TAG_DROPOUT = 1


def make_ctr(step: int, sample, tag: int, idx):
    sample = np.asarray(sample, dtype=np.uint64)
    idx = np.asarray(idx, dtype=np.uint64)
    sample, idx = np.broadcast_arrays(sample, idx)
    ctr = np.empty(sample.shape + (4,), dtype=np.uint64)
    ctr[..., 0] = step & 0xFFFFFFFF
    ctr[..., 1] = (step >> 32) & 0xFFFFFFFF
    ctr[..., 2] = sample
    ctr[..., 3] = (np.uint64(tag) << np.uint64(24)) | idx
    return ctr


def make_key(seed: int):
    return np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint64)
