"""Pins oracle/philox_ref.py against the Random123 known-answer vectors for philox4x32-10 (kat_vectors of the
Random123 distribution: counter/key all zeros, all ones, and the pi digits)."""
import numpy as np

from oracle import philox_ref as P


def test_random123_known_answers():
    z = P.philox4x32_10(np.zeros(4, np.uint32), np.zeros(2, np.uint32))
    assert [hex(int(v)) for v in z] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    f = P.philox4x32_10(np.full(4, 0xFFFFFFFF, np.uint32), np.full(2, 0xFFFFFFFF, np.uint32))
    assert [hex(int(v)) for v in f] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    p = P.philox4x32_10(np.array([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], np.uint32),
                        np.array([0xa4093822, 0x299f31d0], np.uint32))
    assert [hex(int(v)) for v in p] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_episode_uniforms_shape_and_range():
    U = P.episode_uniforms(12345, np.arange(1000), np.full(1000, 3))
    assert U.shape == (1000, 12) and U.min() >= 0 and U.max() < 1
    assert abs(U.mean() - 0.5) < 0.01
    U2 = P.episode_uniforms(12345, np.arange(1000), np.full(1000, 4))
    assert not np.allclose(U, U2)


def test_philox_normal_is_standard_normal():
    z = P.philox_normal(12345, np.arange(200000), np.full(200000, 3), np.arange(200000) % 1000)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01 and np.isfinite(z).all()
    assert abs(((z > 1.0).mean()) - 0.158655) < 0.004
    # a different step or env gives a different draw, the same counter the same one
    a = P.philox_normal(7, np.array([5]), np.array([2]), np.array([9]))
    assert a == P.philox_normal(7, np.array([5]), np.array([2]), np.array([9]))
    assert a != P.philox_normal(7, np.array([5]), np.array([2]), np.array([10]))
