"""numpy restatement of the counter hash behind the dropout of the strain embedding's training path
(posteriflow_amd/csrc/pf_dense.h: enc_drop_hash).  TEST INFRASTRUCTURE (see oracle/__init__.py): lets a test rebuild the
exact 0 | 1/(1-p) factors the kernels applied, so that the oracle's forward and autograd gradients can be evaluated with
nn.Dropout replaced by a multiplication with those factors (nn.Dropout's own random stream is not reproducible across
devices)."""
import numpy as np


def _u32(x):
    return np.uint32(x & 0xFFFFFFFF)


def drop_key(seed: int, site: int) -> int:
    k = (seed * 0x9E3779B9 + site * 0x85EBCA6B + 0x7F4A7C15) & 0xFFFFFFFF
    k ^= k >> 15
    k = (k * 0x2C1B3C6D) & 0xFFFFFFFF
    k ^= k >> 13
    return k


def drop_hash(seed: int, site: int, idx: np.ndarray) -> np.ndarray:
    """16-bit hash of (seed, site, element index): elements 2 j and 2 j + 1 take the low / high half of
    lowbias32(j ^ key); uint32 arithmetic throughout"""
    idx = np.asarray(idx, dtype=np.uint32)
    with np.errstate(over="ignore"):
        x = (idx >> np.uint32(1)) ^ _u32(drop_key(seed & 0xFFFFFFFF, site))
        x = x ^ (x >> np.uint32(16)); x = x * np.uint32(0x7FEB352D)
        x = x ^ (x >> np.uint32(15)); x = x * np.uint32(0x846CA68B)
        x = x ^ (x >> np.uint32(16))
    return np.where(idx & np.uint32(1), x >> np.uint32(16), x & np.uint32(0xFFFF)).astype(np.uint32)


def seed32(seed64: int) -> int:
    """the 32-bit seed the kernels derive from the 64-bit dropout_seed of PfEmbedTrainDesc"""
    return (seed64 ^ (seed64 >> 32)) & 0xFFFFFFFF


def factors(p: float, seed32_: int, site: int, n: int, start: int = 0) -> np.ndarray:
    """float32 [n]: the factor of elements start .. start + n - 1 of dropout layer `site`"""
    if not p > 0.0:
        return np.ones(n, dtype=np.float32)
    thr = np.uint32(int(np.float32(p) * np.float32(65536.0) + np.float32(0.5)))
    idx = (np.arange(n, dtype=np.uint64) + np.uint64(start)).astype(np.uint32)
    keep = drop_hash(seed32_, site, idx) >= thr
    return np.where(keep, np.float32(1.0) / (np.float32(1.0) - np.float32(p)), np.float32(0.0)).astype(np.float32)
