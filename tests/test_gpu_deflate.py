"""GPU parity tests of the Deflate / zlib / gzip ENCODE path (rows f-2, f-3; run with -m gpu on an
MI355X): df_encode_buffer / df_enc_* / df_gpu_encode_device through the C ABI against the CPU
oracle (oracle/deflate_oracle.c) -- LZSS codes, block decisions and the stream, bit for bit -- on
the reference's own vectors (tests/golden/deflate_vectors.json) and on seeded inputs."""
import gzip
import json
import os
import random
import zlib

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, sample
from test_oracle_deflate import VEC, expand, pack_lsb

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(pkg):
    e = pkg.GpuEngine(0, 1)
    yield e
    e.close()


def dev_encode(pkg, eng, data, kind=0):
    import torch
    n = len(data)
    tin = torch.frombuffer(bytearray(data) + bytearray(16), dtype=torch.uint8).cuda()
    cap = pkg.deflate_bound(n)
    tout = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    k = eng.deflate_encode_device(kind, tin.data_ptr(), n, tout.data_ptr(), cap)
    return bytes(tout[:k].cpu().numpy()), tin


def check(pkg, oracle, eng, data, codes=True):
    got, tin = dev_encode(pkg, eng, data)
    if codes:
        want_codes = oracle.lzss_tokens_raw(data)
        got_codes = eng.deflate_debug_codes(tin.data_ptr(), len(data))
        assert got_codes.shape == want_codes.shape
        bad = np.nonzero((got_codes != want_codes).any(axis=1))[0]
        assert bad.size == 0, (int(bad[0]), got_codes[bad[0]].tolist(), want_codes[bad[0]].tolist())
        e = oracle.DeflateEncoder()
        e.feed(data, oracle.ACTION_FINISH)
        want_blocks = [(b[1], b[2], b[3]) for b in e.blocks()]
        got_blocks = [(b[1], b[2], b[3]) for b in eng.deflate_debug_blocks()]
        assert got_blocks == want_blocks
    want = oracle.deflate_encode(data)
    assert got == want
    assert zlib.decompress(got, -15) == data
    return got


@pytest.mark.parametrize("v", VEC["deflate"], ids=lambda v: v["name"])
def test_reference_deflate_vectors(pkg, oracle, eng, v):
    want = bytes(v["bytes"]) if "bytes" in v else pack_lsb(v["bits"])
    data = expand(v["input"])
    assert pkg.deflate_compress(data) == want
    assert check(pkg, oracle, eng, data) == want


@pytest.mark.parametrize("v", VEC["containers"], ids=lambda v: v["name"])
def test_reference_container_vectors(pkg, v):
    kind = pkg.ZLIB if v["kind"] == "zlib" else pkg.GZIP
    assert pkg.deflate_compress(expand(v["input"]), kind, dict_=expand(v.get("dict"))) == bytes(v["bytes"])


@pytest.mark.parametrize("v", [v for v in VEC["lzss"] if "dict" not in v and v["name"] not in ("test_7", "test_11")],
                         ids=lambda v: v["name"])
def test_lzss_inputs_with_deflate_parameters(pkg, oracle, eng, v):
    """the inputs of the reference's LZSS tests under Inflater's parameters (window 0x8000, 258)"""
    check(pkg, oracle, eng, expand(v["input"]))


def _inputs():
    rnd = random.Random(11)
    words = [bytes(rnd.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rnd.randint(1, 9))) for _ in range(300)]
    text = b" ".join(rnd.choice(words) for _ in range(60000))
    yield "text", text
    yield "random", bytes(rnd.getrandbits(8) for _ in range(200000))
    yield "runs", b"".join(bytes([rnd.randrange(4)]) * rnd.randint(1, 700) for _ in range(2000))
    yield "dna", bytes(rnd.choice(b"ACGT") for _ in range(150000))
    yield "period256", bytes(range(256)) * 600
    yield "zeros", bytes(300000)
    yield "far", b"abc" + b"d" * (0x8000 - 3) + b"abc" + b"e" * 40000 + b"abcd"
    yield "mixed", text[:70000] + bytes(rnd.getrandbits(8) for _ in range(70000)) + text[:70000]
    yield "skewed", bytes(min(255, int(rnd.expovariate(0.08))) for _ in range(120000))


@pytest.mark.parametrize("name,data", list(_inputs()), ids=[n for n, _ in _inputs()])
def test_seeded_inputs(pkg, oracle, eng, name, data):
    check(pkg, oracle, eng, data)


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 5, 257, 258, 259, 260, 261, 262, 4095, 4096, 4097, 8191, 8192, 8193,
                               32767, 32768, 32769, 65534, 65535, 65536, 65537, 131070, 131071])
def test_sizes(pkg, oracle, eng, n):
    rnd = random.Random(n)
    check(pkg, oracle, eng, bytes(rnd.choice(b"ab") for _ in range(n)))
    check(pkg, oracle, eng, bytes(rnd.choice(b"abcdefgh \n") for _ in range(n)))


@pytest.mark.parametrize("i", [1, 2, 3])
def test_samples(pkg, oracle, eng, i):
    check(pkg, oracle, eng, sample(i)[:3000000])


def test_containers_and_mirrors(pkg, oracle):
    d = sample(1)[:400000]
    assert pkg.deflate_compress(d, pkg.ZLIB) == oracle.deflate_encode(d, oracle.ZLIB)
    assert pkg.deflate_compress(d, pkg.GZIP) == oracle.deflate_encode(d, oracle.GZIP)
    assert zlib.decompress(pkg.deflate_compress(d, pkg.ZLIB)) == d
    assert gzip.decompress(pkg.deflate_compress(d, pkg.GZIP)) == d
    # `data.encode(&mut Inflater::new(), Action::Finish).collect()`
    enc = pkg.Inflater()
    assert bytes(pkg.encode(iter(d[:5000]), enc, pkg.Action.FINISH)) == oracle.deflate_encode(d[:5000])
    # Inflater: Run then Finish == one iterator (the bytes the reference hands out under Run come with the Finish)
    enc = pkg.Inflater()
    enc.write(d[:100000]); enc.end(pkg.Action.RUN)
    enc.write(d[100000:]); enc.end(pkg.Action.FINISH)
    assert enc.read_all() == oracle.deflate_encode(d, oracle.DEFLATE)
    # GZipEncoder: Run ends the container (gzip/encoder.rs:120-133); the Finish behind it yields nothing
    enc, ref = pkg.GZipEncoder(), oracle.WrapperEncoder(oracle.GZIP)
    enc.write(d[:100000]); enc.end(pkg.Action.RUN)
    enc.write(d[100000:]); enc.end(pkg.Action.FINISH)
    assert enc.read_all() == ref.encode_iter(d[:100000], oracle.ACTION_RUN) + ref.encode_iter(d[100000:], oracle.ACTION_FINISH)


def test_big_corpus(pkg, oracle, eng):
    """64 MiB of the bench corpus: stream equals the oracle's, inflates back with zlib"""
    import corpus
    d = corpus.corpus_bytes(64 << 20)
    got, _ = dev_encode(pkg, eng, d)
    assert zlib.decompress(got, -15) == d
    want = oracle.deflate_encode(d)
    assert got == want


def test_reference_quirk_match_free_dynamic_block(pkg, oracle, eng):
    """match-free dynamic blocks: HDIST = 0 and no distance length, exactly as the reference writes them"""
    d = bytes(b for i in range(32) for j in range(32) for b in (i, 32 + j))  # 64 symbols, no trigram twice
    got, _ = dev_encode(pkg, eng, d)
    assert got == oracle.deflate_encode(d)
    assert eng.deflate_stats()["dynamic_without_distances"] == 1
    with pytest.raises(zlib.error):
        zlib.decompress(got, -15)


@pytest.mark.parametrize("dn", [1, 2, 3, 100, 32767, 32768, 32769, 100000])
def test_with_dict(pkg, oracle, dn):
    """Inflater::with_dict / ZlibEncoder::with_dict: the last 0x8000 bytes of the dictionary are the window"""
    rnd = random.Random(dn)
    words = [bytes(rnd.choice(b"abcdefgh") for _ in range(rnd.randint(2, 7))) for _ in range(40)]
    text = b" ".join(rnd.choice(words) for _ in range(30000))
    dict_, data = text[:dn], text[50000:50000 + 90000]
    for kind, okind in ((pkg.DEFLATE, oracle.DEFLATE), (pkg.ZLIB, oracle.ZLIB)):
        got = pkg.deflate_compress(data, kind, dict_=dict_)
        assert got == oracle.deflate_encode(data, okind, dict_)
    z = pkg.deflate_compress(data, pkg.ZLIB, dict_=dict_)
    do = zlib.decompressobj(zdict=dict_)
    assert do.decompress(z) == data
    enc = pkg.ZlibEncoder.with_dict(dict_)
    assert enc.encode_all(data) == z
    with pytest.raises(pkg.CompressionError):
        pkg.deflate_compress(data, pkg.GZIP, dict_=dict_)


def test_length_limited_tables(pkg, oracle, eng):
    """a block whose Huffman tree is deeper than the limit: make_table's package-merge path
    (huffman/cano_huff_table.rs:58-151) on the GPU, same lengths as the oracle's"""
    hits = 0
    cases = []
    for seed in (23, 99, 235):
        rnd = random.Random(seed)
        n = rnd.choice([300, 2000, 20000, 70000])
        cases.append(bytes((rnd.getrandbits(8) & rnd.getrandbits(8) & rnd.getrandbits(8)) for _ in range(n)))
    for data in cases:
        got, _ = dev_encode(pkg, eng, data)
        assert got == oracle.deflate_encode(data)
        hits += eng.deflate_stats()["limited_tables"]
    assert hits >= 1


@pytest.mark.parametrize("n", [524286, 524287, 524288, 524289, 524290, 524291, 557055, 557056, 557057, 1048575, 1048576,
                               1048577, 1048578, 1081343, 1081344, 1081345])
def test_sort_chunk_boundaries(pkg, oracle, eng, n):
    """sizes around the 512 Ki-position sort chunks (and chunk + window): chains must run across the seams"""
    rnd = random.Random(n)
    words = [bytes(rnd.choice(b"abcdefghij") for _ in range(rnd.randint(1, 6))) for _ in range(60)]
    d = b" ".join(rnd.choice(words) for _ in range(n // 3))[:n]
    assert len(d) == n
    check(pkg, oracle, eng, d)


def _hash16(t):
    return ((((t[0] << 16) | (t[1] << 8) | t[2]) * 0x7A7C4F9F7A7C4F9F) & 0xFFFFFFFFFFFFFFFF) >> 48


def _match_cases():
    rnd = random.Random(77)
    # repeats of every length around the 16 bytes a step of k_df_match2 compares and around the 258-byte limit,
    # each one behind a different filler so that the candidates of one trigram differ in length
    para = bytes(rnd.choice(b"abcdefghijklmnopqrstuvwxyz ") for _ in range(600))
    parts = []
    for ln in list(range(3, 36)) + [63, 64, 65, 255, 256, 257, 258, 259, 260, 300, 520]:
        parts.append(para[:ln] + bytes([rnd.randrange(128, 256)]) * rnd.randint(1, 5))
    rnd.shuffle(parts)
    yield "lengths", b"".join(parts) * 3
    # the same, ending inside a repeat: limits below 16, between 16 and 272, and a match that runs to the last byte
    body = b"".join(parts)
    for cut in (1, 2, 3, 5, 15, 16, 17, 31, 100, 271, 272, 273):
        yield "tail%d" % cut, body + para[:300] + b"#" + para[:cut]
    # more than 255 candidates, the long ones beyond the first 255 (they must not be found) and just inside
    short = b"".join(b"xyz" + bytes([65 + (i % 23), 97 + (i % 19)]) for i in range(400))
    yield "chain255", b"xyz0123456789ABCDEFGHIJ" + short[:5 * 254] + b"xyz0123456789ABCDEFGHIJ" + short + b"xyz0123456789ABCDEFGHIJ"
    # equally long candidates: the nearest one wins
    yield "ties", (b"hello world, this is it:" + b"A") + (b"hello world, this is it:" + b"B") * 5 + b"hello world, this is it:C" * 2
    # two trigrams with the same 16-bit hash share a chain
    seen = {}
    pair = None
    for t in range(1 << 24):
        tri = (t * 2654435761) & 0xFFFFFF
        b3 = bytes([tri >> 16, (tri >> 8) & 255, tri & 255])
        hh = _hash16(b3)
        if hh in seen and seen[hh] != b3:
            pair = (seen[hh], b3)
            break
        seen[hh] = b3
    assert pair is not None
    x, y = pair
    yield "collide", (x + b"0123456789abcdefgh" + y + b"0123456789abcdefgh") * 40 + x + b"0123456789abcdefXY" + y + b"0123"
    # codes of one length in a row (orbits of the parse that start one byte apart stay apart for long stretches)
    w3 = [bytes(rnd.choice(b"abcdefgh") for _ in range(3)) for _ in range(40)]
    yield "words3", b"".join(rnd.choice(w3) for _ in range(30000))
    w5 = [bytes(rnd.choice(b"abcdefgh") for _ in range(5)) for _ in range(200)]
    yield "words5", b"".join(rnd.choice(w5) for _ in range(20000))
    # a paragraph repeated (every candidate agrees to the limit) and four symbols at random (full chains, short matches)
    yield "deep", para[:400] * 200
    yield "dna", bytes(rnd.choice(b"ACGT") for _ in range(90000))


@pytest.mark.parametrize("name,data", list(_match_cases()), ids=[n for n, _ in _match_cases()])
def test_match_kernels_on_long_and_tied_candidates(pkg, oracle, eng, name, data, monkeypatch):
    """Both match kernels -- candidates read off the sorted order (k_df_match2, the default) and the chain walk of
    rounds 1 and 2 (BZ_DF_MATCH=walk) -- give the oracle's LZSS codes on inputs made for the places where they differ:
    candidates that agree on exactly 15, 16, 17 ... bytes, matches cut by the end of the text, chains of more than
    255, ties, hash collisions."""
    check(pkg, oracle, eng, data)
    monkeypatch.setenv("BZ_DF_MATCH", "walk")
    monkeypatch.setenv("BZ_DF_PARSE", "doubling")  # (and the parse by pointer doubling instead of canonical orbits)
    check(pkg, oracle, eng, data)
    monkeypatch.delenv("BZ_DF_MATCH")
    monkeypatch.delenv("BZ_DF_PARSE")
    monkeypatch.setenv("BZ_DF_CUTS", "after")  # (the block chain behind the marking kernel instead of beside it)
    check(pkg, oracle, eng, data)


@pytest.mark.parametrize("n", [1 << 20, (1 << 20) + 4096 * 3 + 17, 3 * (1 << 20) + 1234])
def test_block_chain_in_pieces(pkg, oracle, eng, n, monkeypatch):
    """from 256 tiles on the block chain runs in four pieces beside the marking kernel (k_df_cuts resumes where the
    bits are final): same blocks as the chain in one piece behind it"""
    rnd = random.Random(n)
    words = [bytes(rnd.choice(b"abcdefghijklmnop") for _ in range(rnd.randint(1, 7))) for _ in range(200)]
    d = b" ".join(rnd.choice(words) for _ in range(n // 3))[:n]
    d = d[:n // 2] + bytes(rnd.getrandbits(8) for _ in range(70000)) + d[n // 2 + 70000:]  # (stored blocks in the middle)
    assert len(d) == n
    got = check(pkg, oracle, eng, d)
    monkeypatch.setenv("BZ_DF_CUTS", "after")
    assert check(pkg, oracle, eng, d) == got


def test_empty_inputs_in_containers(pkg, oracle):
    for kind, okind in ((pkg.DEFLATE, oracle.DEFLATE), (pkg.ZLIB, oracle.ZLIB), (pkg.GZIP, oracle.GZIP)):
        assert pkg.deflate_compress(b"", kind) == oracle.deflate_encode(b"", okind)
    assert pkg.deflate_compress(b"", pkg.ZLIB, dict_=b"abc") == oracle.deflate_encode(b"", oracle.ZLIB, b"abc")


# ---- Action::Flush inside a stream (deflate/encoder.rs:170-195, :227-235, :638-647): byte-aligned segments,
# ---- the window and decompress_len carry over; against the oracle's Inflater fed iterator by iterator
def _flush_case(pkg, oracle, pieces, dict_=b""):
    """pieces: [(bytes, action)]; returns (gpu stream, oracle stream)"""
    enc = pkg.Inflater(dict_=dict_)
    ref = oracle.DeflateEncoder(dict_)
    got = bytearray()
    for data, act in pieces:
        enc.write(data)
        enc.end(act)
        got += enc.read_all()
        ref.feed(data, int(act))
    return bytes(got), ref.output()


def test_flush_segments(pkg, oracle):
    A = pkg.Action
    text = sample(1)
    rnd = random.Random(11)
    noise = bytes(rnd.randrange(256) for _ in range(200000))
    cases = {
        "two_segments": [(text[:50000], A.FLUSH), (text[50000:120000], A.FINISH)],
        "window_carries_over": [(text[:40000], A.FLUSH), (text[:40000], A.FLUSH), (text[:40000], A.FINISH)],
        "flush_at_start_and_twice": [(b"", A.FLUSH), (text[:1000], A.FLUSH), (b"", A.FLUSH), (text[1000:3000], A.FINISH)],
        "flush_then_empty_finish": [(text[:70000], A.FLUSH), (b"", A.FINISH)],
        "run_between": [(text[:10], A.RUN), (text[10:30000], A.FLUSH), (text[30000:30001], A.RUN), (text[30001:90000], A.FINISH)],
        "single_bytes": [(b"a", A.FLUSH), (b"a", A.FLUSH), (b"b", A.FLUSH), (b"", A.FINISH)],
        # decompress_len carried over a flush: the first block behind it closes early ...
        "carry_closes_block_early": [(text[:60000], A.FLUSH), (text[60000:140000], A.FINISH)],
        # ... at once, empty, when the carry is within one code of 0xFFFF ...
        "carry_0xFFFF_empty_first_block": [(text[:0xFFFF], A.FLUSH), (text[0xFFFF:0xFFFF + 5000], A.FINISH)],
        "carry_0xFFFE": [(noise[:0xFFFE], A.FLUSH), (b"\0" * 600, A.FINISH)],
        # ... and a stored block behind a flush repeats the carried bytes (incompressible input)
        "stored_blocks_repeat_bytes": [(noise[:30000], A.FLUSH), (noise[30000:70000], A.FLUSH), (noise[70000:200000], A.FINISH)],
        "finish_then_more": [(text[:5000], A.FINISH), (text[:100], A.FLUSH), (b"", A.FINISH)],
        "many_small": [(text[i * 997:(i + 1) * 997], A.FLUSH) for i in range(40)] + [(b"", A.FINISH)],
    }
    for name, pieces in cases.items():
        got, want = _flush_case(pkg, oracle, pieces)
        assert got == want, name
    got, want = _flush_case(pkg, oracle, [(text[:20000], A.FLUSH), (text[20000:50000], A.FINISH)], dict_=text[60000:100000])
    assert got == want, "with_dict"
    # a flushed stream whose padding happens to be empty inflates (a sanity check of the bytes, not of parity)
    rng = random.Random(5)
    for trial in range(30):
        pieces, pos = [], 0
        while pos < 300000:
            k = rng.choice([0, 1, 17, 4000, 65535, 70000])
            pieces.append((text[pos:pos + k], rng.choice([A.RUN, A.FLUSH, A.FLUSH])))
            pos += k
        pieces.append((b"", A.FINISH))
        got, want = _flush_case(pkg, oracle, pieces)
        assert got == want, ("random", trial)


def test_wrappers_end_their_container_at_the_first_none(pkg, oracle):
    """ZlibEncoder / GZipEncoder under Action::Run and Action::Flush (zlib/encoder.rs:118-152,
    gzip/encoder.rs:88-135): header + what the inner Inflater yields under that action + trailer, piece by piece
    against the oracle's iterator-level restatement; afterwards nothing, and the caller's iterator is not pulled."""
    A = pkg.Action
    text = sample(1)
    rnd = random.Random(21)
    noise = bytes(rnd.randrange(256) for _ in range(150000))
    inputs = [b"", b"a", text[:260], text[:261], text[:262], text[:300], text[:65535], text[:65536], text[:65536 + 261],
              text[:65536 + 600], text[:140000], noise[:70000], noise, b"\0" * 200000, text[:30000] + noise[:66000] + text[:9000]]
    for kind, cls in ((oracle.ZLIB, pkg.ZlibEncoder), (oracle.GZIP, pkg.GZipEncoder)):
        for act in (A.RUN, A.FLUSH, A.FINISH):
            for i, d in enumerate(inputs):
                enc, ref = cls(), oracle.WrapperEncoder(kind)
                got = enc.encode_all(d, act)
                assert got == ref.encode_iter(d, int(act)), (kind, int(act), i, len(d))
                # finished: a second iterator yields nothing and is left alone
                it = iter(b"more input")
                assert enc.next(it, A.FINISH) is None and bytes(it) == b"more input"
                assert ref.encode_iter(b"more input", int(A.FINISH)) == b"" and ref.pulled == 0
                assert enc.encode_all(b"xyz", A.FINISH) == b""
    # Run over more than one closed block, with a dictionary (zlib only), through the byte iterator
    d = text[:100000] + noise[:50000]
    enc, ref = pkg.ZlibEncoder.with_dict(text[200000:240000]), oracle.WrapperEncoder(oracle.ZLIB, text[200000:240000])
    assert bytes(pkg.encode(d, enc, A.RUN)) == ref.encode_iter(d, int(A.RUN))
    # the Finish stream is the one-shot stream
    assert pkg.GZipEncoder().encode_all(d, A.FINISH) == oracle.deflate_encode(d, oracle.GZIP)


def test_long_streams_in_parts(oracle):
    """A segment longer than BZ_DF_PART_MIB (default 1 GiB; 1 MiB here, read once per process) is encoded in
    parts: each but the last with look-ahead, keeping the blocks that cannot change, the next one starting at
    the first block left out, inside the byte the previous one ended in.  Same bytes as one pass (oracle)."""
    import subprocess
    import sys
    from conftest import ROOT
    code = r'''
import importlib, random, sys, zlib
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
pkg = importlib.import_module("rust-compression_amd")
from oracle import oracle
from conftest import sample
rng = random.Random(3)
text = (sample(1) + sample(2)) * 8
noise = bytes(rng.randrange(256) for _ in range(1 << 20))
inputs = {
    "text": text[:5_500_000],
    "zeros": b"\0" * 3_300_000,
    "noise_text": noise + text[:1_500_000] + noise[:700_000] + text[:900_000],
    "just_over": text[:(1 << 20) + 0x10000 + 1025],
    "exactly_guard": text[:(1 << 20) + 0x10000 + 1024],
}
for name, d in inputs.items():
    for kind in (0, 1, 2):
        got = pkg.deflate_compress(d, kind)
        assert got == oracle.deflate_encode(d, kind), (name, kind, len(got))
    assert zlib.decompress(pkg.deflate_compress(d, 1)) == d, name
d = inputs["text"]
assert pkg.deflate_compress(d[100000:], 0, dict_=d[:100000]) == oracle.deflate_encode(d[100000:], 0, d[:100000])
# flushed segments that are themselves longer than a part
A = pkg.Action
enc, ref = pkg.Inflater(), oracle.DeflateEncoder()
got = bytearray()
for piece, act in [(d[:2_600_000], A.FLUSH), (d[2_600_000:2_600_010], A.FLUSH), (d[2_600_010:], A.FINISH)]:
    enc.write(piece); enc.end(act); got += enc.read_all(); ref.feed(piece, int(act))
assert bytes(got) == ref.output()
print("ok")
''' % (ROOT, ROOT)
    e = dict(os.environ, BZ_DF_PART_MIB="1")
    out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-500:] + out.stderr[-3000:]


@pytest.mark.parametrize("part_mib", [2, 3])
def test_many_part_seams_equal_oracle_golden(part_mib):
    """64 MiB of the bench corpus in parts of 2 / 3 MiB: some thirty seams, several of them at a block that starts
    between the literals and the reference of one step of the lazy parse (the next part then has to take the
    parse up at the step, not at the block).  SHA-256 against the oracle's stream (corpus_hashes.json)."""
    import subprocess
    import sys
    gold = json.load(open(os.path.join(GOLDEN, "corpus_hashes.json")))["deflate_text_64mib"]
    code = (
        "import importlib,sys,hashlib;sys.path.insert(0,%r);pkg=importlib.import_module('rust-compression_amd');"
        "import torch,corpus;n=64<<20;eng=pkg.GpuEngine(0,1);d=corpus.corpus_on_device(n,torch.device('cuda',0));"
        "cap=(pkg.lib().df_encode_bound(n)+15)&~15;o=torch.empty(cap,dtype=torch.uint8,device='cuda');torch.cuda.synchronize();"
        "k=eng.deflate_encode_device(pkg.DEFLATE,d.data_ptr(),n,o.data_ptr(),cap);"
        "print(k,hashlib.sha256(bytes(o[:k].cpu().numpy())).hexdigest())" % ROOT)
    e = dict(os.environ, BZ_DF_PART_MIB=str(part_mib))
    out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == "%d %s" % (gold["bytes"], gold["sha256"]), out.stdout[-300:]


def test_two_gib_stream_equals_oracle_golden(pkg, eng):
    """2 GiB of the bench corpus through df_gpu_encode_device: more positions than one part holds (1 GiB +
    look-ahead), i.e. two parts with a bit-level seam; SHA-256 and length against the oracle's stream
    (tests/golden/corpus_hashes.json, made by tests/golden/make_corpus_hashes.py)."""
    import hashlib
    import torch
    import corpus
    gold = json.load(open(os.path.join(GOLDEN, "corpus_hashes.json"))).get("deflate_text_2gib")
    if gold is None:
        pytest.skip("no golden hash for the 2 GiB Deflate stream")
    n = 2 << 30
    d_in = corpus.corpus_on_device(n, torch.device("cuda", 0))
    cap = (pkg.lib().df_encode_bound(n) + 15) & ~15
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    k = eng.deflate_encode_device(pkg.DEFLATE, d_in.data_ptr(), n, d_out.data_ptr(), cap)
    assert k == gold["bytes"]
    h = hashlib.sha256()
    for off in range(0, k, 256 << 20):
        h.update(bytes(d_out[off:min(k, off + (256 << 20))].cpu().numpy()))
    assert h.hexdigest() == gold["sha256"]
