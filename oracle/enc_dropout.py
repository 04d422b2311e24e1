"""numpy restatement of the counter hash behind the dropout of the strain embedding's training path
(posteriflow_amd/csrc/pf_dense.h: enc_drop_hash).  TEST INFRASTRUCTURE (see oracle/__init__.py): lets a test rebuild the
exact 0 | 1/(1-p) factors the kernels applied, so that the oracle's forward and autograd gradients can be evaluated with
nn.Dropout replaced by a multiplication with those factors (nn.Dropout's own random stream is not reproducible across
devices)."""
import numpy as np


def drop_hash(seed: int, site: int, idx: np.ndarray) -> np.ndarray:
    """24-bit hash of (seed, site, element index); uint32 arithmetic throughout"""
    idx = np.asarray(idx, dtype=np.uint32)
    with np.errstate(over="ignore"):
        h = np.uint32(seed & 0xFFFFFFFF) ^ np.uint32((site * 0x9E3779B9) & 0xFFFFFFFF)
        h = np.uint32(h)
        h = h ^ (idx + np.uint32(0x7F4A7C15) + np.uint32((int(h) << 6) & 0xFFFFFFFF) + np.uint32(int(h) >> 2))
        h = h * np.uint32(0x85EBCA6B); h = h ^ (h >> np.uint32(13)); h = h * np.uint32(0xC2B2AE35); h = h ^ (h >> np.uint32(16))
        h = h + idx * np.uint32(0x27D4EB2F); h = h ^ (h >> np.uint32(15)); h = h * np.uint32(0x2C1B3C6D); h = h ^ (h >> np.uint32(12))
    return h >> np.uint32(8)


def seed32(seed64: int) -> int:
    """the 32-bit seed the kernels derive from the 64-bit dropout_seed of PfEmbedTrainDesc"""
    return (seed64 ^ (seed64 >> 32)) & 0xFFFFFFFF


def factors(p: float, seed32_: int, site: int, n: int, start: int = 0) -> np.ndarray:
    """float32 [n]: the factor of elements start .. start + n - 1 of dropout layer `site`"""
    if not p > 0.0:
        return np.ones(n, dtype=np.float32)
    thr = np.uint32(int(np.float32(p) * np.float32(16777216.0) + np.float32(0.5)))
    idx = (np.arange(n, dtype=np.uint64) + np.uint64(start)).astype(np.uint32)
    keep = drop_hash(seed32_, site, idx) >= thr
    return np.where(keep, np.float32(1.0) / (np.float32(1.0) - np.float32(p)), np.float32(0.0)).astype(np.float32)
