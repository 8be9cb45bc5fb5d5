"""Pins the oracle's Morton code against the reference's own known-answer vectors (src/sph/morton.rs:184-253)."""
import numpy as np


def test_encode_lookup_examples(oracle_lib):  # morton.rs:190-198
    L = oracle_lib
    assert L.orc_morton_encode_lookup(2, 2) == 12
    assert L.orc_morton_encode_lookup(3, 6) == 45
    assert L.orc_morton_encode_lookup(4, 0) == 16
    assert L.orc_morton_encode_lookup(0b1111000100100000, 0b1001110110001100) == 0b11010111101000111000010010100000


def test_encode_bitfiddle_examples(oracle_lib):  # morton.rs:200-209
    L = oracle_lib
    assert L.orc_morton_encode_bitfiddle(2, 2) == 12
    assert L.orc_morton_encode_bitfiddle(3, 6) == 45
    assert L.orc_morton_encode_bitfiddle(4, 0) == 16
    assert L.orc_morton_encode_bitfiddle(0b1111000100100000, 0b1001110110001100) == 0b11010111101000111000010010100000


def test_decode_examples(oracle_lib):  # morton.rs:215-228
    L = oracle_lib
    assert (L.orc_morton_decode_x(12), L.orc_morton_decode_y(12)) == (2, 2)
    assert (L.orc_morton_decode_x(45), L.orc_morton_decode_y(45)) == (3, 6)
    assert (L.orc_morton_decode_x(16), L.orc_morton_decode_y(16)) == (4, 0)
    m = 0b11010111101000111000010010100000
    assert L.orc_morton_decode_x(m) == 0b1111000100100000
    assert L.orc_morton_decode_y(m) == 0b1001110110001100


def test_find_bigmin_examples(oracle_lib):  # morton.rs:234-251
    L = oracle_lib
    for cur in (16, 19, 29, 35):
        assert L.orc_morton_find_bigmin(cur, 12, 45) == 36
    assert L.orc_morton_find_bigmin(14, 12, 45) == 15
    assert L.orc_morton_find_bigmin(15, 12, 45) == 36


def test_lookup_equals_bitfiddle_and_roundtrip(oracle_lib):
    L = oracle_lib
    rng = np.random.default_rng(1)
    xs = rng.integers(0, 65536, 5000)
    ys = rng.integers(0, 65536, 5000)
    for x, y in zip(xs.tolist() + [0, 65535, 0, 65535], ys.tolist() + [0, 65535, 65535, 0]):
        a = L.orc_morton_encode_lookup(x, y)
        assert a == L.orc_morton_encode_bitfiddle(x, y)
        assert (L.orc_morton_decode_x(a), L.orc_morton_decode_y(a)) == (x, y)


def test_find_bigmin_is_next_code_in_rect(oracle_lib):
    """Property behind morton.rs:151-182: BIGMIN = smallest code > cur that lies in the rectangle."""
    L = oracle_lib
    rng = np.random.default_rng(2)
    for _ in range(300):
        x0, y0 = rng.integers(0, 40, 2)
        x1, y1 = x0 + rng.integers(0, 12), y0 + rng.integers(0, 12)
        mn, mx = L.orc_morton_encode_lookup(int(x0), int(y0)), L.orc_morton_encode_lookup(int(x1), int(y1))
        cur = int(rng.integers(mn, mx + 1))
        if L.orc_morton_is_in_rect(cur, mn, mx):
            continue  # the reference only calls it on codes outside the rect (neighborhood_search.rs:215-221)
        expect = next((c for c in range(cur + 1, mx + 1) if L.orc_morton_is_in_rect(c, mn, mx)), None)
        if expect is not None:
            assert L.orc_morton_find_bigmin(cur, mn, mx) == expect, (cur, mn, mx)
