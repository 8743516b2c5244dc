"""The Deflate oracle (oracle/deflate_oracle.c) against every known-answer vector the reference's
tests hold for the path (tests/golden/deflate_vectors.json, written by make_deflate_vectors.py),
and against zlib as an independent decoder."""
import json
import os
import random
import zlib

import pytest

from oracle import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
VEC = json.load(open(os.path.join(HERE, "golden", "deflate_vectors.json")))


def expand(spec):
    out = bytearray()
    for s in spec or []:
        if s[0] == "bytes":
            out += bytes(s[1])
        elif s[0] == "cycle":
            lo, hi, n = s[1:]
            out += bytes((lo + i % (hi - lo)) & 0xFF for i in range(n))
        elif s[0] == "repeat":
            out += bytes([s[1]]) * s[2]
    return bytes(out)


def pack_lsb(items):
    acc, cnt, out = 0, 0, bytearray()
    for v, n in items:
        acc |= (v & ((1 << n) - 1)) << cnt
        cnt += n
        while cnt >= 8:
            out.append(acc & 0xFF)
            acc >>= 8
            cnt -= 8
    if cnt:
        out.append(acc & 0xFF)
    return bytes(out)


@pytest.mark.parametrize("v", VEC["lzss"], ids=lambda v: v["name"])
def test_lzss_token_vectors(v):
    want = []
    for tok, count in v["tokens"]:
        want += [tuple(tok)] * count
    got = oracle.lzss_tokens(expand(v["input"]), expand(v.get("dict")), v["comparison"], v["window"], v["max_match"],
                             v["min_match"], v["lazy"])
    assert got == want


@pytest.mark.parametrize("v", VEC["deflate"], ids=lambda v: v["name"])
def test_deflate_vectors(v):
    want = bytes(v["bytes"]) if "bytes" in v else pack_lsb(v["bits"])
    data = expand(v["input"])
    got = oracle.deflate_encode(data)
    assert got == want
    assert zlib.decompress(got, -15) == data


def test_code_tables():
    for ln, pos, lcode, lext, lbits, dcode, dext, dbits in VEC["codes"]:
        c, e, b = oracle.deflate_convert(0, ln - 3)
        assert (c + 257, e, b) == (lcode, lext, lbits)
        c, e, b = oracle.deflate_convert(1, pos)
        assert (c, e, b) == (dcode, dext, dbits)


@pytest.mark.parametrize("v", VEC["containers"], ids=lambda v: v["name"])
def test_container_vectors(v):
    kind = oracle.ZLIB if v["kind"] == "zlib" else oracle.GZIP
    got = oracle.deflate_encode(expand(v["input"]), kind, expand(v.get("dict")))
    assert got == bytes(v["bytes"])


def test_checksums():
    c = VEC["checksums"]["crc32_ieee_reverse"]
    assert oracle.crc32_ieee(expand(c["input"])) == c["value"]
    rnd = random.Random(5)
    for n in (0, 1, 5549, 5550, 5551, 70000):
        d = bytes(rnd.getrandbits(8) for _ in range(n))
        assert oracle.adler32(d) == zlib.adler32(d)
        assert oracle.crc32_ieee(d) == zlib.crc32(d)


def _inputs():
    rnd = random.Random(11)
    words = [bytes(rnd.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rnd.randint(1, 9))) for _ in range(300)]
    text = b" ".join(rnd.choice(words) for _ in range(60000))
    yield "text", text
    yield "random", bytes(rnd.getrandbits(8) for _ in range(200000))
    yield "runs", b"".join(bytes([rnd.randrange(4)]) * rnd.randint(1, 700) for _ in range(2000))
    yield "dna", bytes(rnd.choice(b"ACGT") for _ in range(150000))
    yield "period", bytes(range(256)) * 600
    yield "short", b"hello hello hello"
    yield "mixed", text[:70000] + bytes(rnd.getrandbits(8) for _ in range(70000)) + text[:70000]


@pytest.mark.parametrize("name,data", list(_inputs()), ids=[n for n, _ in _inputs()])
def test_streams_decode_with_zlib(name, data):
    z = oracle.deflate_encode(data)
    assert zlib.decompress(z, -15) == data
    assert zlib.decompress(oracle.deflate_encode(data, oracle.ZLIB)) == data
    import gzip
    assert gzip.decompress(oracle.deflate_encode(data, oracle.GZIP)) == data
    # the token stream re-expands to the input
    out = bytearray()
    for t in oracle.lzss_tokens(data):
        if t[0] == "sym":
            out.append(t[1])
        else:
            for _ in range(t[1]):
                out.append(out[-1 - t[2]])
    assert bytes(out) == data


def test_blocks_and_segments():
    rnd = random.Random(3)
    data = bytes(rnd.choice(b"ab \n") for _ in range(300000))
    e = oracle.DeflateEncoder()
    e.feed(data[:100000], oracle.ACTION_RUN)
    e.feed(data[100000:], oracle.ACTION_FINISH)
    assert e.output() == oracle.deflate_encode(data)
    blocks = e.blocks()
    assert sum(b[1] for b in blocks) == len(data) and all(b[1] <= 0xFFFF for b in blocks)
    assert sum(b[3] for b in blocks) <= 8 * len(e.output())


def test_reference_quirk_match_free_dynamic_block():
    """A dynamic block without any match: the reference writes HDIST = 0 and no distance code length at all
    (deflate/encoder.rs:431-436 `unwrap_or((0, &0))`, :449-451 with an empty offset list), which RFC 1951
    decoders reject.  The restatement keeps it: the stream is the reference's."""
    d = bytes(b for i in range(32) for j in range(32) for b in (i, 32 + j))  # 64 symbols, no trigram twice
    assert all(t[0] == "sym" for t in oracle.lzss_tokens(d))
    e = oracle.DeflateEncoder()
    e.feed(d, oracle.ACTION_FINISH)
    assert [b[2] for b in e.blocks()] == [2]
    with pytest.raises(zlib.error):
        zlib.decompress(e.output(), -15)


def test_wrapper_iterator_level_restatement():
    """ZlibEncoder::next / GZipEncoder::next (zlib/encoder.rs:118-152, gzip/encoder.rs:88-135) at the iterator
    level: with Action::Finish the bytes are the reference's container vectors; with Run / Flush the container
    still ends at the inner encoder's first None: header + the Inflater's bytes so far + trailer, then nothing."""
    import struct
    with open(os.path.join(HERE, "golden", "sample1.ref"), "rb") as f:
        text = f.read()
    for kind in (oracle.ZLIB, oracle.GZIP):
        for d in (b"", b"a", text[:70000], text[:200000]):
            w = oracle.WrapperEncoder(kind)
            assert w.encode_iter(d, oracle.ACTION_FINISH) == oracle.deflate_encode(d, kind)
            assert w.pulled == len(d)
            assert w.encode_iter(b"abc", oracle.ACTION_FINISH) == b"" and w.pulled == 0
    hdr = {oracle.ZLIB: 2, oracle.GZIP: 10}
    for kind in (oracle.ZLIB, oracle.GZIP):
        for d in (b"", text[:300], text[:70000], text[:200000]):
            # Run: the Inflater alone, fed the same bytes, has handed out exactly these bytes
            inner = oracle.DeflateEncoder()
            inner.feed(d, oracle.ACTION_RUN)
            body = inner.output()
            got = oracle.WrapperEncoder(kind).encode_iter(d, oracle.ACTION_RUN)
            trailer = struct.pack(">I", zlib.adler32(d)) if kind == oracle.ZLIB else struct.pack("<II", zlib.crc32(d), len(d))
            assert got[hdr[kind]:] == body + trailer
            if len(d) > 70000:
                assert len(body) > 1000     # closed blocks did come out under Run
            elif len(d) < 65536:
                assert body == b""
            # Flush: the flushed segment (non-final block, byte aligned), then the trailer
            inner = oracle.DeflateEncoder()
            inner.feed(d, oracle.ACTION_FLUSH)
            got = oracle.WrapperEncoder(kind).encode_iter(d, oracle.ACTION_FLUSH)
            assert got[hdr[kind]:] == inner.output() + trailer
            z = zlib.decompressobj(-15)
            assert z.decompress(inner.output()) == d   # a flushed segment inflates to its input
