"""oracle/philox.py pinned to the published Philox4x32-10 known-answer vectors (Random123 kat_vectors), plus the
sharding property the cascade relies on: rows drawn for global samples [lo, hi) equal those rows of the full draw."""
import numpy as np

from oracle import philox as ph

KAT = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
       ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
       ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
        (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]


def test_philox_known_answers():
    for ctr, key, want in KAT:
        assert ph.philox4x32_10(np.array(ctr, dtype=np.uint32), key).tolist() == list(want)


def test_rows_depend_only_on_global_sample_index():
    full = ph.randn(10, 50, seed=0x1234567890, draw_id=3)
    part = ph.randn(4, 50, seed=0x1234567890, draw_id=3, first_sample=5)
    assert np.array_equal(full[5:9], part)
    assert not np.array_equal(ph.randn(4, 50, 0x1234567890, 4, 5), part)       # another draw is another stream


def test_moments():
    z = ph.randn(4000, 48, seed=7, draw_id=1).astype(np.float64)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01
    assert abs((z ** 3).mean()) < 0.03 and abs((z ** 4).mean() - 3.0) < 0.1
    assert np.isfinite(z).all()


def test_uniform_grid_is_exact_in_fp32():
    """u01 = (top 23 bits + 1/2) * 2^-23 (csrc/rng.hip): every grid point is exactly representable, strictly inside (0, 1)
    and equally spaced -- the 24-bit variant rounds k + 1/2 to even above 2^23 and reaches 1.0."""
    k = np.array([0, 1, 2 ** 22, 2 ** 23 - 2, 2 ** 23 - 1], dtype=np.uint32)
    u = (k.astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 8388608.0)
    exact = (k.astype(np.float64) + 0.5) / 8388608.0
    assert np.array_equal(u.astype(np.float64), exact) and u.min() > 0 and u.max() < 1
    bad = (np.float32(2 ** 24 - 1) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
    assert bad == np.float32(1.0)                                          # what the old formula did
