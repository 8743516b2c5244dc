"""Pins the decoder restatement in oracle/bz2_oracle.c (src/bzip2/decoder.rs) against the
reference's decoder fixtures (src/bzip2/mod.rs:84-148) and libbzip2."""
import bz2
import os
import random

import pytest

from conftest import GOLDEN, sample

E_DATA, E_MAGIC_FIRST, E_MAGIC = -1, -4, -5


@pytest.mark.parametrize("i", [1, 2, 3, 4])
def test_sample_fixtures_decode(oracle, i):
    with open(os.path.join(GOLDEN, "sample%d.bz2" % i), "rb") as f:
        z = f.read()
    out, st = oracle.decode(z)
    assert st == 0 and out == sample(i)  # sample4.bz2 is two concatenated streams (mod.rs:141-148)


@pytest.mark.parametrize("level", [1, 5, 9])
def test_roundtrip_own_encoder_and_libbzip2(oracle, level):
    d = sample(2) + b"a" * 70000 + sample(1)[:30000]
    assert oracle.decode(oracle.encode(d, level)) == (d, 0)
    assert oracle.decode(bz2.compress(d, level)) == (d, 0)


def test_small_and_empty(oracle):
    for d in (b"", b"a", b"a\n", b"ab" * 500, b"a" * 1000, bytes(range(256)) * 3):
        assert oracle.decode(oracle.encode(d, 9)) == (d, 0)
    assert oracle.decode(oracle.encode(b"", 9) + oracle.encode(b"xyz", 1)) == (b"xyz", 0)


def test_errors(oracle):
    d = sample(1)[:60000]
    z = oracle.encode(d, 9)
    assert oracle.decode(b"")[1] == E_MAGIC_FIRST
    assert oracle.decode(b"BZh0")[1] == E_MAGIC_FIRST
    assert oracle.decode(b"XYh9" + z[4:]) == (d, 0)          # 'B','Z','h' are read, not compared (decoder.rs:175-180)
    bad = bytearray(z)
    bad[len(z) // 2] ^= 0x10
    out, st = oracle.decode(bytes(bad))
    assert st == E_DATA
    assert oracle.decode(z[:len(z) // 2])[1] == E_DATA       # a short read returns fewer bits, never Eof
    assert oracle.decode(z + b"garbage!") == (d, E_MAGIC)    # a second "stream" with a bad level byte
    crc_bad = bytearray(z)
    crc_bad[12] ^= 1                                          # stored block CRC
    out, st = oracle.decode(bytes(crc_bad))
    assert st == E_DATA and out == d                          # bytes of the block are yielded before the check


def test_random_roundtrips(oracle):
    rng = random.Random(4)
    for _ in range(20):
        n = rng.randint(0, 5000)
        d = bytes(rng.randrange(rng.choice([2, 4, 256])) for _ in range(n))
        assert oracle.decode(bz2.compress(d, 1)) == (d, 0)
        assert oracle.decode(oracle.encode(d, 1)) == (d, 0)
