"""Numpy restatement of the counter-based noise generator behind ``bg_philox_randn`` (csrc/rng.hip).

TEST INFRASTRUCTURE.  Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the
generator torch/cuRAND/rocRAND use on the device) -- PINNED against the Random123 known-answer vectors in
``tests/test_oracle_philox.py``.  The reference draws its ancestral noise from torch's global device generator
(sample.py:153 -> diffusers ``randn_tensor(device=...)``): the *stream* is not reproducible across devices or batch
shardings even upstream, only the distribution is part of the contract; the kernel's counter layout (one counter per
(element block, global sample, draw)) is what makes sharded runs reproducible here.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter: uint32 array [..., 4]; key: (k0, k1) python ints.  -> uint32 [..., 4]."""
    c = [counter[..., i].astype(np.uint64) for i in range(4)]
    k0, k1 = key
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        n0 = (p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0)
        n2 = (p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1)
        c = [n0 & MASK, p1 & MASK, n2 & MASK, p0 & MASK]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return np.stack(c, -1).astype(np.uint32)


def counters(n_samples, per_sample, draw_id, first_sample):
    blocks = (per_sample + 3) // 4
    gs = (first_sample + np.arange(n_samples, dtype=np.uint64))[:, None].repeat(blocks, 1)
    c = np.zeros((n_samples, blocks, 4), dtype=np.uint32)
    c[..., 0] = np.arange(blocks, dtype=np.uint32)[None]
    c[..., 1] = (gs & MASK).astype(np.uint32)
    c[..., 2] = np.uint32(draw_id)
    c[..., 3] = (np.uint64(0xB9E50000) | ((gs >> np.uint64(32)) & np.uint64(0xFFFF))).astype(np.uint32)
    return c


def raw_bits(n_samples, per_sample, seed, draw_id, first_sample=0):
    bits = philox4x32_10(counters(n_samples, per_sample, draw_id, first_sample), (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    return bits.reshape(n_samples, -1)[:, :per_sample]


def randn(n_samples, per_sample, seed, draw_id, first_sample=0):
    """float32 [n_samples, per_sample]: Box-Muller on the four words of each counter (fp32 arithmetic like the kernel)."""
    bits = philox4x32_10(counters(n_samples, per_sample, draw_id, first_sample), (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    u = ((bits >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 8388608.0)   # 23 bits + 1/2: exact in fp32
    r0 = np.sqrt(np.float32(-2.0) * np.log(u[..., 0]))
    r1 = np.sqrt(np.float32(-2.0) * np.log(u[..., 2]))
    a0 = np.float32(6.28318530717958647692) * u[..., 1]
    a1 = np.float32(6.28318530717958647692) * u[..., 3]
    v = np.stack([r0 * np.cos(a0), r0 * np.sin(a0), r1 * np.cos(a1), r1 * np.sin(a1)], -1).astype(np.float32)
    return v.reshape(n_samples, -1)[:, :per_sample]
