"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI,
against the CPU oracle on the same inputs -- bit-exact."""
import bz2
import importlib
import hashlib
import json
import os
import random

import pytest

from conftest import GOLDEN, ROOT, product, sample

pytestmark = pytest.mark.gpu

with open(os.path.join(GOLDEN, "reference_vectors.json")) as f:
    V = json.load(f)


@pytest.fixture(scope="module")
def eng(pkg):
    e = pkg.GpuEngine(0, 16)
    yield e
    e.close()


def test_library_loaded_and_device(pkg):
    assert pkg.device_count() >= 1


def test_known_answer_stream(pkg):
    v = V["stream_a_nl_level9"]
    assert pkg.compress(bytes.fromhex(v["input_hex"]), 9).hex() == v["output_hex"]


@pytest.mark.parametrize("v", V["bwt_pos"], ids=lambda v: v["src_hex"][:16])
def test_bwt_reference_vectors(eng, v):
    assert eng.debug_bwt(bytes.fromhex(v["src_hex"])) == v["pos"]


@pytest.mark.parametrize("v", V["bwt_L"], ids=lambda v: v["src"][:12])
def test_bwt_L_vectors(eng, v):
    src = v["src"].encode()
    sa = eng.debug_bwt(src)
    assert bytes(src[(s - 1) % len(src)] for s in sa) == v["L"].encode()


def test_bwt_random_vs_oracle(eng, oracle):
    rng = random.Random(11)
    for _ in range(60):
        n = rng.randint(1, 3000)
        k = rng.choice([1, 2, 3, 4, 16, 256])
        s = bytes(rng.randrange(k) for _ in range(n))
        assert eng.debug_bwt(s) == oracle.bwt(s), (n, k)


def test_bwt_periodic_tie_rule(eng, oracle):
    rng = random.Random(12)
    cases = [b"aaaa", b"abab", b"abcabcabc", b"cabcabcab", b"a", b"aa", b"ab" * 500, b"aaaa\xfb" * 200]
    for _ in range(40):
        p = rng.randint(1, 9)
        u = bytes(rng.randrange(3) for _ in range(p))
        cases.append(u * rng.randint(2, 300))
    for s in cases:
        assert eng.debug_bwt(s) == oracle.bwt(s), s[:20]


def test_bwt_deep_lcp(eng, oracle):
    para = bytes(random.Random(5).randrange(97, 123) for _ in range(997))
    s = (para * 40)[:35000]
    assert eng.debug_bwt(s) == oracle.bwt(s)


def test_code_lengths_vs_oracle(eng, oracle):
    rng = random.Random(3)
    fib = [1, 1]
    while len(fib) < 25:
        fib.append(fib[-1] + fib[-2])
    tables = [fib[:20], fib[:25], [5] * 7, [1, 1, 1, 1, 2], [0] * 6, [3, 3, 2, 2, 1, 1, 1], [0, 0, 0]]
    for _ in range(60):
        n = rng.randint(3, 258)
        r = rng.uniform(0.35, 0.9)
        f = [int(900000 * (1 - r) * r ** i * rng.uniform(0.7, 1.3)) for i in range(n)]
        rng.shuffle(f)
        tables.append(f)
    # ties everywhere: the heap's tie-breaks decide who gets which length, and the pipelined heap of
    # k_huff_tables has several sift-downs in flight at once.  (Totals stay below 2^20 occurrences, the probe's domain:
    # a block holds 900 001 symbols at most, and the device's 32-bit package weights equal the reference's usize ones
    # below 2^24 occurrences per package = 16 levels x 2^20, k_huff.hip "DOMAIN".)
    for k in range(240):
        n = rng.choice([2, 3, 4, 5, 7, 8, 9, 16, 17, 31, 33, 64, 100, 129, 200, 257, 258])
        mode = k % 5
        if mode == 0:
            f = [rng.randint(0, 3) for _ in range(n)]
        elif mode == 1:
            f = [rng.randint(0, 3500) for _ in range(n)]
        elif mode == 2:
            f = [1] * n
        elif mode == 3:
            f = [int(2 ** (rng.random() * 11.5)) for _ in range(n)]
        else:
            f = [rng.choice([0, 1, 5, 5, 5, 900]) for _ in range(n)]
        tables.append(f)
    fired = 0
    for f in tables:
        got, lm = eng.debug_code_lengths(f)
        exp, elm = oracle.bzip2_code_lengths(f, 17)
        assert (got, lm) == (exp, elm), f[:8]
        fired += lm
    assert fired >= 5  # the length-limited path (cano_huff_table.rs:58-151) is exercised
    with pytest.raises(Exception):  # out of the domain: refused, not answered
        eng.debug_code_lengths([1 << 19, 1 << 19, 5])


SMALL = [b"", b"a", b"a\n", b"ab" * 500, b"a" * 1000, b"aabbaabbaabbaabb\n", b"a" * 255, b"a" * 256, b"a" * 259,
         b"abc" * 7, bytes(range(256)), bytes(range(256)) * 3, b"\x00" * 70000, b"xy" * 40000]


@pytest.mark.parametrize("i", range(len(SMALL)))
def test_small_streams(pkg, oracle, i):
    d = SMALL[i]
    out = pkg.compress(d, 9)
    assert out == oracle.encode(d, 9)
    assert bz2.decompress(out) == d


@pytest.mark.parametrize("i", [1, 2, 3, 4])
def test_samples_level9(pkg, oracle, i):
    d = sample(i)
    out = pkg.compress(d, 9)
    assert out == oracle.encode(d, 9)
    assert bz2.decompress(out) == d


@pytest.mark.parametrize("i,level", [(1, 1), (2, 2), (3, 3), (2, 1), (4, 1)])
def test_samples_reference_levels(pkg, oracle, i, level):
    """src/bzip2/mod.rs:84-139 uses levels 1/2/3 (several blocks per sample)."""
    d = sample(i)
    out = pkg.compress(d, level)
    assert out == oracle.encode(d, level)
    assert bz2.decompress(out) == d


def _text(n, seed):
    rng = random.Random(seed)
    words = ["".join(rng.choice("etaoinshrdlucmfw") for _ in range(rng.randint(2, 9))) for _ in range(600)]
    out = bytearray()
    while len(out) < n:
        out += rng.choice(words).encode() + (b" " if rng.random() > 0.1 else b".\n")
    return bytes(out[:n])


def test_multiblock_text_level9(pkg, oracle):
    d = _text(2_100_000, 1)
    out = pkg.compress(d, 9)
    assert out == oracle.encode(d, 9)


def test_multiblock_runs_level1(pkg, oracle):
    rng = random.Random(5)
    d = b"".join(bytes([rng.randrange(4)]) * rng.randint(1, 700) for _ in range(3000))
    out = pkg.compress(d, 1)
    assert out == oracle.encode(d, 1)
    assert bz2.decompress(out) == d


def test_block_stats_match_oracle(pkg, oracle):
    import torch
    d = sample(2)
    eng = pkg.GpuEngine(0, 8)
    eng.profile(2)  # (bit 1: the per-pass figures of block_sections)
    t = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
    o = torch.empty(pkg.encode_bound(len(d)) + 16, dtype=torch.uint8, device="cuda")
    n = eng.encode_device(1, t.data_ptr(), len(d), o.data_ptr(), o.numel())
    got = eng.block_stats()
    exp_stream, exp = oracle.encode(d, 1, with_stats=True)
    assert bytes(o[:n].cpu().numpy()) == exp_stream
    assert len(got) == len(exp)
    for g, e in zip(got, exp):
        for k in ("nblock", "block_crc", "orig_ptr", "mtf_count", "in_use_count", "group_num", "n_selectors", "max_len"):
            assert g[k] == e[k], k
    # the figures of the reference's other two debug lines (src/bzip2/encoder.rs:483-498 "pass k: size is .., grp uses are ..",
    # :556-636 "bits: mapping .., selectors .., code lengths .., codes .."): VERDICT r5 missing #4
    sec = eng.block_sections()
    assert len(sec) == len(exp)
    for g, e in zip(sec, exp):
        for k in ("pass_size", "fave", "bits_mapping", "bits_selectors", "bits_lengths", "bits_codes"):
            assert g[k] == e[k], (k, g[k], e[k])
    eng.close()
    # ... and on text at level 9 (two tables per block take the length-limited path), several blocks in a batch
    import corpus
    d9 = corpus.chapter(5, 2_000_000)
    eng = pkg.GpuEngine(0, 8)
    eng.profile(2)
    t = torch.frombuffer(bytearray(d9), dtype=torch.uint8).cuda()
    o = torch.empty(pkg.encode_bound(len(d9)) + 16, dtype=torch.uint8, device="cuda")
    n = eng.encode_device(9, t.data_ptr(), len(d9), o.data_ptr(), o.numel())
    exp_stream, exp = oracle.encode(d9, 9, with_stats=True)
    assert bytes(o[:n].cpu().numpy()) == exp_stream
    sec = eng.block_sections()
    assert [{k: g[k] for k in ("pass_size", "fave", "bits_mapping", "bits_selectors", "bits_lengths", "bits_codes")} for g in sec] == \
           [{k: e[k] for k in ("pass_size", "fave", "bits_mapping", "bits_selectors", "bits_lengths", "bits_codes")} for e in exp]
    eng.close()


# ---- Action semantics through the streaming context ---------------------------------------

def test_streaming_run_then_finish(pkg, oracle):
    d = sample(1)
    enc = pkg.BZip2Encoder(1)
    ora = oracle.Encoder(1)
    a = enc.encode_all(d[:40000], pkg.Action.RUN)
    b = enc.encode_all(d[40000:], pkg.Action.FINISH)
    assert a == ora.encode_iter(d[:40000], oracle.ACTION_RUN)
    assert b == ora.encode_iter(d[40000:], oracle.ACTION_FINISH)
    assert a + b == oracle.encode(d, 1)


def test_streaming_flush_sequences(pkg, oracle):
    rng = random.Random(8)
    d = sample(3)[:60000] + _text(150000, 3)
    for trial in range(4):
        enc = pkg.BZip2Encoder(1)
        ora = oracle.Encoder(1)
        pos = 0
        while pos < len(d):
            step = rng.randint(1, 90000)
            act = rng.choice([pkg.Action.RUN, pkg.Action.FLUSH, pkg.Action.RUN])
            piece = d[pos:pos + step]
            pos += step
            assert enc.encode_all(piece, act) == ora.encode_iter(piece, int(act)), (trial, pos, act)
        assert enc.encode_all(b"", pkg.Action.FINISH) == ora.encode_iter(b"", oracle.ACTION_FINISH)


def test_streaming_write_after_finish(pkg, oracle):
    """Input written BEHIND Action::Finish (encoder.rs:80-85 + :671-697: EncoderInner::next has no `finished` test): the
    reference goes on collecting it and writes further blocks behind the trailer whenever a block has come together, while
    flush() and finish() do nothing any more (:718-739).  Byte for byte against the oracle's BZip2Encoder mirror, call by
    call: Finish -> too little for a block -> more (a block comes out, byte-aligned, without a header) -> Flush (the pad
    byte of the bit writer only) -> Run -> Finish twice; and a stream whose FIRST call is an empty Finish (the header is
    written again in front of the first block: block_no is still 1)."""
    d = (sample(1) + sample(2)) * 3
    plan = [(d[:150000], 2), (d[150000:200000], 2), (d[200000:420000], 2), (b"", 1), (d[:300000], 0), (b"", 2), (b"", 2)]
    enc, ora = pkg.BZip2Encoder(1), oracle.Encoder(1)
    sizes = []
    for piece, act in plan:
        got, want = enc.encode_all(piece, pkg.Action(act)), ora.encode_iter(piece, act)
        assert got == want, (len(piece), act, len(got), len(want))
        sizes.append(len(got))
    assert sizes[1] == 0 and sizes[2] > 1000 and sizes[3] == 1 and sizes[4] > 1000 and sizes[6] == 0, sizes
    enc, ora = pkg.BZip2Encoder(1), oracle.Encoder(1)
    assert enc.encode_all(b"", pkg.Action.FINISH) == ora.encode_iter(b"", oracle.ACTION_FINISH)
    got, want = enc.encode_all(d[:250000], pkg.Action.FINISH), ora.encode_iter(d[:250000], oracle.ACTION_FINISH)
    assert got == want and got[:4] == b"BZh1" and len(got) > 1000
    # pieces that cross the pipeline's chunking: 2.5 MB behind a Finish at level 9, in three writes and one Run
    rng = random.Random(12)
    enc, ora = pkg.BZip2Encoder(9), oracle.Encoder(9)
    big = _text(2_500_000, 5)
    assert enc.encode_all(big[:1000], pkg.Action.FINISH) == ora.encode_iter(big[:1000], oracle.ACTION_FINISH)
    pos = 1000
    while pos < len(big):
        k = rng.choice([1, 70000, 900000, 1200000])
        act = rng.choice([0, 1, 2])
        piece = big[pos:pos + k]
        pos += len(piece)
        assert enc.encode_all(piece, pkg.Action(act)) == ora.encode_iter(piece, act), (pos, act)


def test_flush_on_fresh_encoder(pkg, oracle):
    enc = pkg.BZip2Encoder(9)
    ora = oracle.Encoder(9)
    assert enc.encode_all(b"", pkg.Action.FLUSH) == ora.encode_iter(b"", oracle.ACTION_FLUSH)
    assert enc.encode_all(b"hello world", pkg.Action.FLUSH) == ora.encode_iter(b"hello world", oracle.ACTION_FLUSH)
    assert enc.encode_all(b"", pkg.Action.FINISH) == ora.encode_iter(b"", oracle.ACTION_FINISH)


def test_encoder_iterator_api(pkg):
    enc = pkg.BZip2Encoder(9)
    out = bytes(pkg.encode(b"a\n", enc, pkg.Action.FINISH))
    assert out.hex() == V["stream_a_nl_level9"]["output_hex"]


def test_invalid_level(pkg):
    for lv in (0, 10):
        with pytest.raises(ValueError):
            pkg.BZip2Encoder(lv)
        with pytest.raises(ValueError):
            pkg.compress(b"x", lv)


def test_determinism(pkg):
    d = _text(1_200_000, 9)
    h = {hashlib.sha256(pkg.compress(d, 9)).hexdigest() for _ in range(3)}
    assert len(h) == 1


def test_cpp_host_mirror(pkg):
    """host/compression.hpp: the reference's test_unit / test_long written against the C++ mirror."""
    import subprocess
    from conftest import ROOT
    host = os.path.join(ROOT, "rust-compression_amd", "host")
    exe = os.path.join(host, "example_test_unit")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(host, "example_test_unit.cpp"),
                           "-L" + os.path.join(ROOT, "rust-compression_amd"), "-lbz2_mi355x",
                           "-Wl,-rpath," + os.path.join(ROOT, "rust-compression_amd"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "test_unit ok" in r.stdout


def test_host_buffer_roundtrip_large(pkg):
    """bz_encode_buffer on 40 MB (PCIe-inclusive path): decodes to the input."""
    import corpus
    d = corpus.chapter(3, 16 << 20) + corpus.stress_t2(8 << 20) + corpus.chapter(4, 16 << 20)
    out = pkg.compress(d, 9)
    assert bz2.decompress(out) == d


@pytest.mark.parametrize("world", [2, 3, 5])
def test_slab_sharded_partition_equals_serial(pkg, oracle, world):
    """The N > 1 split (rust-compression_amd/sharded.py) replayed on one GPU: `world` engines each
    take a slab of tiles, the cut chain is handed from engine to engine, and the concatenated
    blocks must give the oracle's stream (runs and blocks straddle the slab edges)."""
    import torch
    sharded = importlib.import_module("rust-compression_amd.sharded")
    rng = random.Random(40 + world)
    runs = b"".join(bytes([rng.randrange(3)]) * rng.randint(1, 900) for _ in range(1500))
    d = _text(300_000, 7) + runs + _text(260_000, 8) + b"z" * 70_000 + _text(100_001, 9)
    level = 1
    n = len(d)
    t_in = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
    engs = [pkg.GpuEngine(0, 16) for _ in range(world)]
    lasts = [e.slab_begin(level, t_in.data_ptr(), n, *sharded.slab_tiles(n, r, world)) for r, e in enumerate(engs)]
    for r, e in enumerate(engs):
        e.slab_count(max(lasts[:r], default=-1))
    start, counts = 0, []
    for r, e in enumerate(engs):
        nb, start, _ = e.slab_finish(start, r == world - 1)
        counts.append(nb)
    assert start == n and sum(counts) >= 5
    cap_words = pkg.encode_bound(n) // 4 + 64
    bufs, woff, blen, crcs = [], [], [], []
    base = 0
    for r, e in enumerate(engs):
        p = torch.zeros(cap_words, dtype=torch.int32, device="cuda")
        w, b, c, used = e.encode_blocks(0, 1, counts[r], p.data_ptr(), cap_words)
        bufs.append(p)
        woff += [x + base for x in w]
        blen += b
        crcs += c
        base += cap_words
    allp = torch.cat(bufs)
    out = torch.empty(pkg.encode_bound(n) + 16, dtype=torch.uint8, device="cuda")
    k, _, _, _ = engs[0].assemble(level, allp.data_ptr(), woff, blen, crcs, out.data_ptr(), out.numel())
    got = bytes(out[:k].cpu().numpy())
    assert got == oracle.encode(d, level)
    for e in engs:
        e.close()


def _rand(n, seed, k=256):
    rng = random.Random(seed)
    return bytes(rng.randrange(k) for _ in range(n))


WIDE = {
    "random256_3blocks": lambda: _rand(2_500_000, 21),
    "alphabet129": lambda: _rand(400_000, 22, 129),
    "alphabet128": lambda: _rand(400_000, 23, 128),
    "alphabet2": lambda: _rand(300_000, 24, 2),
    "alphabet3_periodic_mix": lambda: (_rand(977, 25, 3) * 700)[:600_000] + _rand(100_000, 26, 3),
    "zeros_40MB": lambda: b"\x00" * 40_000_000,
    "long_runs_mixed": lambda: b"".join(bytes([i % 7]) * (3000 + 517 * (i % 11)) for i in range(4000)),
    "t2_deep_lcp": lambda: __import__("corpus").stress_t2(2_400_000),
    "text_zeros_random": lambda: _text(700_000, 31) + b"\x00" * 3_000_000 + _rand(500_000, 32) + _text(300_000, 33),
    "run_255_boundaries": lambda: b"".join(b"a" * r + b"b" for r in (254, 255, 256, 257, 509, 510, 511, 1020, 4, 3, 5) * 4000),
}


@pytest.mark.parametrize("name", sorted(WIDE))
def test_wide_inputs_level9(pkg, oracle, name):
    """Alphabet sizes around the key-packing switches, long runs (cut windows, 255-cuts), deep LCPs."""
    d = WIDE[name]()
    out = pkg.compress(d, 9)
    assert out == oracle.encode(d, 9), name
    assert bz2.decompress(out) == d


def test_wide_inputs_small_blocks(pkg, oracle):
    d = WIDE["long_runs_mixed"]()[:6_000_000] + WIDE["text_zeros_random"]()[:2_000_000]
    for level in (1, 4):
        assert pkg.compress(d, level) == oracle.encode(d, level), level


def test_radix_pass_flavours_agree(oracle):
    """the fused radix passes (default), the three-kernel passes (BZ_ONESWEEP=0) and the fallback from
    one to the other (BZ_ONESWEEP_FAILTEST at the first pass, BZ_ONESWEEP_LATEFAILTEST at the end of a sort)
    give the oracle's stream; so do the round-3 alternatives: flags + apply as two kernels everywhere
    (BZ_FUSED_REFINE=0), the Huffman stage in one workgroup per block (BZ_HUFF_SPLIT=0), the ZLE stage as three
    kernels (BZ_FUSED_ZLE=0) and its redo after a failed ticket check (BZ_FUSED_ZLE_FAILTEST), no period round
    (BZ_PERIOD_ROUND=0).  Each runs in its own process because the switches are read once."""
    import subprocess
    import sys
    code = (
        "import importlib,sys,hashlib;sys.path.insert(0,%r);pkg=importlib.import_module('rust-compression_amd');"
        "sys.path.insert(0,%r+'/tests');from conftest import sample;"
        "d=sample(1)*3+bytes(range(256))*700+sample(2);"
        "print(hashlib.sha256(pkg.compress(d,9)).hexdigest(), hashlib.sha256(pkg.compress(d,1)).hexdigest())" % (ROOT, ROOT))
    d = sample(1) * 3 + bytes(range(256)) * 700 + sample(2)
    want = "%s %s" % (hashlib.sha256(oracle.encode(d, 9)).hexdigest(), hashlib.sha256(oracle.encode(d, 1)).hexdigest())
    for env in ({}, {"BZ_ONESWEEP": "0"}, {"BZ_ONESWEEP_FAILTEST": "1"}, {"BZ_ONESWEEP_LATEFAILTEST": "1"},
                {"BZ_FUSED_REFINE": "0"}, {"BZ_HUFF_SPLIT": "0"}, {"BZ_FUSED_ZLE": "0"}, {"BZ_FUSED_ZLE_FAILTEST": "1"},
                {"BZ_PERIOD_ROUND": "0"}):
        e = dict(os.environ)
        e.update(env)
        out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        assert out.stdout.strip().splitlines()[-1] == want, (env, out.stdout, out.stderr[-500:])
        if "BZ_ONESWEEP_FAILTEST" in env:
            assert "fused radix passes disabled" in out.stderr
        if "BZ_ONESWEEP_LATEFAILTEST" in env:  # a pass after the first misbehaves: the batch is sorted again
            assert "batch sorted again" in out.stderr
        if "BZ_FUSED_ZLE_FAILTEST" in env:
            assert "stage redone with three kernels" in out.stderr


def test_pass_counter_wraps(pkg, oracle):
    """one engine, enough sorts for the fused passes' epoch tag (1023 values) to wrap several times"""
    import torch
    d = sample(1)[:120000] + bytes(range(256)) * 40
    want = oracle.encode(d, 1)
    e = pkg.GpuEngine(0, 8)
    tin = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
    cap = (pkg.encode_bound(len(d)) + 15) & ~15
    tout = torch.empty(cap, dtype=torch.uint8, device="cuda")
    try:
        for i in range(420):
            n = e.encode_device(1, tin.data_ptr(), len(d), tout.data_ptr(), cap)
            if i % 60 == 0 or i > 410:
                assert bytes(tout[:n].cpu().numpy()) == want, i
    finally:
        e.close()


def test_both_forms_of_the_table_kernel(pkg, oracle):
    """k_huff_tables has two forms (k_huff.hip): a wave per table with the heap procedure pipelined for batches of up to
    kTabPipeBlocks = 320 blocks, a lane per table for larger ones.  The same level-1 input -- text, 256-symbol blocks
    (sample2 tiled), random bytes: 369 blocks -- through an engine that holds all of them in one batch and through one
    that holds 64: the same stream, the oracle's (cano_huff_table.rs:153-196 for both).  (Tables that take the
    length-limited path: text at level 9 -- the 2 MB of test_block_stats_match_oracle through the six-wave form, the
    1 GiB golden of test_gpu_configs_full through the one-wave form.)"""
    import torch
    import corpus
    rng = random.Random(77)
    d = corpus.chapter(3, 18_000_000) + (sample(2) * 80)[:16_000_000] + bytes(rng.randrange(256) for _ in range(3_000_000))
    want = oracle.encode(d, 1)
    tin = torch.frombuffer(bytearray(d), dtype=torch.uint8).cuda()
    cap = (pkg.encode_bound(len(d)) + 15) & ~15
    tout = torch.empty(cap, dtype=torch.uint8, device="cuda")
    for blocks in (1024, 64):
        e = pkg.GpuEngine(0, blocks)
        try:
            n = e.encode_device(1, tin.data_ptr(), len(d), tout.data_ptr(), cap)
            got = bytes(tout[:n].cpu().numpy())
            assert len(e.block_stats()) > 320
        finally:
            e.close()
        assert got == want, blocks


def test_streaming_across_chunk_boundaries(oracle):
    """The host pipeline with 1 MiB chunks (BZ_ENC_CHUNK_MIB=1, read once per process): many chunks per
    stream, block tails carried from chunk to chunk on the device, runs that cover whole chunks (the
    pending-run scan then reads back device memory), Run / Flush / Finish sequences in between --
    against the oracle's BZip2Encoder mirror, and the one-shot call against the oracle's stream."""
    import subprocess
    import sys
    code = r'''
import importlib, random, sys
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
pkg = importlib.import_module("rust-compression_amd")
from oracle import oracle
from conftest import sample
rng = random.Random(5)
text = (sample(1) + sample(2)) * 6
inputs = [
    (text[:5_300_000], 1),
    (text[:3_000_000] + b"\0" * 3_400_000 + text[:700_000] + b"q" * 1_048_576 + b"r" * 1_048_577 + text[:100], 1),
    (b"\0" * 4_194_304, 9),
    (text[:2_500_000], 9),
]
for data, level in inputs:
    assert pkg.compress(data, level) == oracle.encode(data, level), ("one-shot", len(data), level)
    enc, ref = pkg.BZip2Encoder(level), oracle.Encoder(level)
    got, want, pos = bytearray(), bytearray(), 0
    while pos < len(data):
        k = rng.choice([1, 4096, 300_000, 1 << 20, (1 << 20) + 1, 2_500_000])
        piece = data[pos:pos + k]
        pos += len(piece)
        act = rng.choice([0, 0, 0, 1]) if pos < len(data) else 2
        enc.write(piece)
        enc.end(act)
        got += enc.read_all()
        want += ref.encode_iter(piece, act)
    assert bytes(got) == bytes(want), ("stream", len(data), level)
print("ok")
''' % (ROOT, ROOT)
    e = dict(os.environ, BZ_ENC_CHUNK_MIB="1")
    out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-500:] + out.stderr[-3000:]


def test_multi_device_context_equals_oracle(oracle):
    """bz_enc_create_multi / bz_encode_buffer_multi (== BZip2Encoder::with_devices): several lanes go round the
    device list -- [0, 0, 0] = six lanes on the one GPU of the test box -- with 1 MiB chunks, so the tail of
    every chunk's input crosses from lane to lane (hipMemcpyPeerAsync between devices), action-only jobs shift
    the lane a chunk lands on, and runs cover whole chunks.  One-shot streams against the oracle's, Run / Flush /
    Finish sequences against the oracle's BZip2Encoder mirror; the single-device entries give the same bytes."""
    import subprocess
    import sys
    code = r'''
import importlib, random, sys
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
pkg = importlib.import_module("rust-compression_amd")
from oracle import oracle
from conftest import sample
rng = random.Random(11)
text = (sample(1) + sample(2)) * 8
inputs = [
    (text[:9_700_000], 1),
    (text[:3_000_000] + b"\0" * 3_400_000 + text[:700_000] + b"q" * 1_048_576 + b"r" * 1_048_577 + text[:100], 1),
    (b"\0" * 6_291_456, 9),
    (text[:7_500_000], 9),
    (b"", 9),
    (b"ab" * 40, 5),
]
for devices in ([0, 0, 0], [0, 0], [0]):
    for data, level in inputs:
        want = oracle.encode(data, level)
        assert pkg.compress(data, level, devices=devices) == want, ("one-shot", devices, len(data), level)
        enc, ref = pkg.BZip2Encoder.with_devices(level, devices), oracle.Encoder(level)
        got, exp, pos = bytearray(), bytearray(), 0
        while True:
            k = rng.choice([1, 4096, 300_000, 1 << 20, (1 << 20) + 1, 2_500_000])
            piece = data[pos:pos + k]
            pos += len(piece)
            act = rng.choice([0, 0, 0, 1]) if pos < len(data) else 2
            enc.write(piece)
            enc.end(act)
            if rng.random() < 0.2 and act != 2:
                enc.end(act)                      # an Action with no input at all: a job of its own
                exp += ref.encode_iter(piece, act) + ref.encode_iter(b"", act)
            else:
                exp += ref.encode_iter(piece, act)
            got += enc.read_all()
            if act == 2:
                break
        assert bytes(got) == bytes(exp), ("stream", devices, len(data), level)
        del enc
    pkg.release_cached_resources()
assert pkg.compress(inputs[0][0], 1) == oracle.encode(inputs[0][0], 1)
print("ok")
''' % (ROOT, ROOT)
    e = dict(os.environ, BZ_ENC_CHUNK_MIB="1")
    out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-500:] + out.stderr[-3000:]


def test_period_round_on_deep_repeats(pkg, oracle, eng):
    """Blocks with a LINEAR period (a paragraph repeated, the block length no multiple of it): the sort finishes
    their groups with one sort by start position instead of log2(n) doubling rounds (k_block_period, the period
    round of run_bwt_once).  Rotation order against the oracle for both directions of the wrap comparison, for
    tiny and large periods, for paragraphs whose first 16 bytes recur inside them (the period search has to reject
    candidates), for a cyclic period (n a multiple of p: k_periodic_place's case, the round must keep its hands
    off) and for a defect in the middle (no linear period: the doubling rounds do it)."""
    rng = random.Random(77)

    def para(k, alpha=b"abcdefgh "):
        return bytes(rng.choice(alpha) for _ in range(k))

    cases = []
    for p_len in (2, 3, 7, 64, 1000, 4096, 30011):
        for n in (70_001, 99_000):
            q = para(p_len) if p_len > 3 else (b"ab", b"abc")[p_len - 2]
            cases.append((q * (n // p_len + 1))[:n])
    # both directions: the byte after the last whole period decides
    base = para(500)
    cases.append((base * 200)[:90_123])
    cases.append((bytes(255 - b for b in base) * 200)[:90_123])
    # the first 16 bytes recur inside the paragraph
    tricky = b"0123456789abcdef" + para(100) + b"0123456789abcdef" + para(333) + b"0123456789abcdeX" + para(50)
    cases.append((tricky * 300)[:95_000])
    # cyclic period, and a defect in the middle
    cases.append(para(1000) * 90)
    d = bytearray((para(777) * 130)[:96_000])
    d[40_000] ^= 1
    cases.append(bytes(d))
    for i, blk in enumerate(cases):
        got = eng.debug_bwt(blk)
        assert got == oracle.bwt(blk), (i, len(blk))
    # the round did run: a deep-repeat block takes a handful of rounds, not seventeen
    eng.debug_bwt((para(4096) * 25)[:99_981])
    assert eng.bwt_stats()["rounds"] <= 6
    # and whole streams (several periodic blocks per stream) against the oracle
    for data, level in (((para(4096) * 400)[:1_500_000], 5), ((b"ab" * 300_000)[:555_555], 1)):
        assert pkg.compress(data, level) == oracle.encode(data, level)


def test_period_round_on_drifting_copies(pkg, oracle, eng):
    """Copies that DRIFT (round 5): a stretch and edited copies of it inside one block -- a few bytes
    inserted or dropped between the copies, so that the block agrees with itself at SEVERAL distances and at none of them
    over half of its length (a tar of similar files; bench.py's corpus "binary").  The period round lists up to eight
    distances per block (k_period_find), keys every survivor by a pair it belongs to at one of them and checks every pair
    of neighbours of the sorted list against the first difference behind it (k_period_mark) -- the comparison that
    /root/reference/src/suffix_array/sais.rs:266-272 defines, so the ORDER is the oracle's: rotation order of whole
    blocks, the rounds it took, and whole streams."""
    rng = random.Random(505)
    alpha = b"etaoinshrdlu ,.\n"

    def stretch(k):
        return bytes(rng.choice(alpha) for _ in range(k))

    def edited(b, every):
        out, pos = bytearray(), 0
        while pos < len(b):
            k = min(len(b) - pos, every + rng.randrange(-every // 4, every // 4))
            out += b[pos:pos + k]
            pos += k
            r = rng.randrange(4)
            if r == 0:
                out += bytes([rng.choice(alpha)])            # a byte inserted: the distance to the copy grows
            elif r == 1:
                pos += 1                                     # a byte dropped: it shrinks
            elif r == 2 and out:
                out[-1] = rng.choice(b"XYZ")                 # a byte changed: the pair is decided here
        return bytes(out)

    base = stretch(30_000)
    blocks = [
        (base + edited(base, 3000) + edited(base, 3000))[:99_000],        # three copies, two to four distances
        (stretch(5000) + base + stretch(77) + edited(base, 1500))[:70_000],  # foreign bytes in front and between
        (base[:20_000] + edited(base[:20_000], 500) + edited(base[:20_000], 700) + edited(base[:20_000], 900))[:82_000],
        edited(base * 3, 4096)[:95_000],                                  # the bench corpus' shape: a byte per 4 KiB
    ]
    # two copies a few distances apart: the second with a byte inserted, one dropped and one changed -- 30 000, 30 001 and
    # 30 000 again; every group is a pair, and every pair's order is read off the first difference behind it.  (Three copies
    # make groups of three whose members stand in MIXED order half of the time -- rot(a) < rot(b) > rot(c) --, which a sort
    # by start or mirrored start cannot lay out: those go on doubling, blocks[0] and blocks[2] above.)
    second = bytearray(base[:14_000] + b"#" + base[14_000:22_000] + base[22_001:])
    second[5_000] ^= 1
    few = (stretch(77) + base + bytes(second) + stretch(999))[:99_000]
    blocks.append(few)
    for i, blk in enumerate(blocks):
        assert eng.debug_bwt(blk) == oracle.bwt(blk), (i, len(blk))
    eng.debug_bwt(few)
    with_round = eng.bwt_stats()["rounds"]
    assert with_round <= 6, with_round  # (doubling alone: log2(14 000 / 12) + 2 = 12 rounds; blocks[3], whose copies drift
    #                                      every 4 KiB -- a dozen distances and more -- takes 11: only the listed ones help)
    big = edited(stretch(400_000) * 5, 4096)
    for data, level in ((big[:1_900_000], 9), (b"".join(blocks), 1)):
        assert pkg.compress(data, level) == oracle.encode(data, level)


def test_link_rounds_on_copies_of_copies(pkg, oracle, eng):
    """Round 6: SMALL groups of survivors -- two to eight members -- are ranked member by member from direct
    comparisons of their rotations (k_link_scan: one scan per stretch of copies, whatever the distances; k_link_keys /
    k_link_permute), in the period round and in link rounds behind it.  The shape it was built for: a stretch that occurs
    four times in a block -- it repeats inside itself and the whole occurs twice --, every copy with a byte changed every few
    KiB (bench.py's corpus "binary"): groups of four whose members stand in mixed order, a changed byte in a middle copy
    between two that agree beyond it.  The comparison is what /root/reference/src/suffix_array/sais.rs:266-272 defines, so
    the ORDER is the oracle's: rotation order of whole blocks, the rounds it took, whole streams, and the same streams with
    the link rounds off."""
    rng = random.Random(606)
    alpha = bytes(range(33, 97))

    def stretch(k):
        return bytes(rng.choice(alpha) for _ in range(k))

    def touched(b, every):
        out = bytearray(b)
        for pos in range(rng.randrange(every), len(out), every):
            out[pos] = out[pos] ^ (1 + rng.randrange(7))
        return bytes(out)

    inner = stretch(9_000)
    unit = inner + stretch(3_000) + touched(inner, 2048) + stretch(1_500)   # repeats inside itself
    blocks = [
        (touched(unit, 4096) + touched(unit, 4096) + touched(unit, 4096) + touched(unit, 4096))[:99_000],  # groups of 2, 4, 6, 8
        (stretch(333) + touched(unit * 2, 1024) + stretch(7) + touched(unit * 2, 3000))[:97_000],
        (touched(inner * 9, 4096) + stretch(100))[:82_000],                    # nine copies: groups beyond kLinkMax go on doubling
        (touched(inner[:5000] * 3, 700) + b"q" * 3000 + touched(inner[:5000] * 3, 900) + (b"abcd" * 5000))[:70_000],  # runs and a short period beside copies
        touched(stretch(20_000) * 2, 4096) * 2,                                # cyclic period of the WHOLE block: equal rotations stay undecided
    ]
    # stretches with a SHORT period cut by changed bytes ("ugh\n" x 30 000 in libbzip2's sample3): groups of thousands of
    # rotations from many stretches, split by where their stretches end (per_ekey) -- deviations upward and downward, stretches
    # of equal length, a stretch across the block's wrap, period 1, two different periods in one block
    ugh = bytearray(b"ugh\n" * 12_000)
    for pos in range(700, len(ugh), 1900):
        ugh[pos] = (ugh[pos] + (1 if (pos // 1900) % 3 else 251)) & 255
    for pos in (10_000, 20_000, 30_000):           # three stretches of the same length behind the same byte
        ugh[pos] = ord("#")
        ugh[pos + 1_204] = ord("#")
    blocks.append(bytes(ugh[:40_000]) + stretch(5_000) + bytes(ugh[40_000:48_000]))
    blocks.append(b"gh\n" + bytes(ugh[:30_000]) + stretch(2_000) + b"ab" * 9_000 + b"!" + b"ab" * 700 + stretch(50) + b"u")
    blocks.append((b"z" * 3_000 + stretch(10) + b"z" * 2_999 + stretch(10) + b"z" * 3_000 + b"y" + touched(inner, 4096))[:40_000])
    rounds = []
    for i, blk in enumerate(blocks):
        assert eng.debug_bwt(blk) == oracle.bwt(blk), (i, len(blk))
        rounds.append(eng.bwt_stats()["rounds"])
    assert rounds[0] <= 7 and rounds[1] <= 7, rounds  # (doubling alone: log2(4096 / 12) + 2 = 10 and more)
    data = b"".join(blocks)
    big = touched((unit * 80)[:1_850_000], 4096)
    for d, level in ((big, 9), (data, 1), (data, 3)):
        assert pkg.compress(d, level) == oracle.encode(d, level)
    # the same streams with the link rounds off (the period round keeps its tables and chains): identical bytes
    code = """
import sys, importlib, hashlib
sys.path.insert(0, %r)
pkg = importlib.import_module("rust-compression_amd")
d = open(sys.argv[1], "rb").read()
print(hashlib.sha256(pkg.compress(d, 1)).hexdigest())
""" % ROOT
    import subprocess
    import sys
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".bin") as f:
        f.write(data)
        f.flush()
        out = subprocess.run([sys.executable, "-c", code, f.name], env=dict(os.environ, BZ_LINK_ROUND="0"), capture_output=True,
                             text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == hashlib.sha256(oracle.encode(data, 1)).hexdigest()


@pytest.mark.parametrize("name", ["fuzz_r6_links_880.bin", "fuzz_r6_small_1522.bin", "fuzz_r6_links_2414.bin"])
def test_fuzz_cases_of_round_6(pkg, oracle, name):
    """Two inputs tools/fuzz_parity.py found against the first build of the link rounds: a level-1 block of copies and,
    behind the cut, a block of one or two bytes in the same batch -- its group of two equal rotations was taken for a small
    group, k_link_scan leaves such blocks alone, and the stale link byte of an earlier batch ranked it; and a PERIODIC block
    of 72 bytes behind the cut, whose equal rotations the scan of a 512-byte step compared beyond the block's end."""
    with open(os.path.join(GOLDEN, name), "rb") as f:
        d = f.read()
    for level in (1, 9):
        assert pkg.compress(d, level) == oracle.encode(d, level), (name, level)
    assert pkg.compress(d + d[:3], 1) == oracle.encode(d + d[:3], 1)


def test_trace_switches_change_no_byte(oracle):
    """BZ_BWT_TRACE / BZ_ENC_TRACE / BZ_DEC_TRACE / BZ_DF_TRACE print timelines on stderr (sort rounds, jobs, chunks, parts) and
    must leave every stream as it is: one process with all four on, a deep-repeat input (so that the period and link
    rounds print), the host pipeline, the decoder and the Deflate path against the oracle."""
    import subprocess
    import sys
    code = """
import sys, importlib, random
sys.path.insert(0, %r)
pkg = importlib.import_module("rust-compression_amd")
from oracle import oracle
rng = random.Random(9)
unit = bytes(rng.randrange(40) for _ in range(30000))
d = (unit * 3 + bytes(rng.randrange(200) for _ in range(5000)) + unit)[:120000]
z = pkg.compress(d, 1)
assert z == oracle.encode(d, 1)
assert pkg.decompress(z) == (d, 0)
assert pkg.deflate_compress(d, pkg.ZLIB) == oracle.deflate_encode(d, pkg.ZLIB)
print("ok")
""" % ROOT
    env = dict(os.environ, BZ_BWT_TRACE="1", BZ_ENC_TRACE="1", BZ_DEC_TRACE="1", BZ_DF_TRACE="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout[-500:] + out.stderr[-3000:]
    assert "sort round" in out.stderr and "bz_enc job" in out.stderr


def test_mailbox_off_changes_no_byte(oracle):
    """BZ_MAILBOX=0: the host's small reads between launches (block counts, a sort round's survivors, bit counts; mail_fetch,
    k_emit.hip) and its small writes (mail_poke) go through hipMemcpyAsync + hipStreamSynchronize as in rounds 1-5 instead of
    the per-stream mailbox in host-mapped memory: same streams either way -- one block, several batches, a deep-repeat block
    with its period and link rounds, levels 1 and 9."""
    import subprocess
    import sys
    code = """
import sys, importlib, random
sys.path.insert(0, %r)
pkg = importlib.import_module("rust-compression_amd")
from oracle import oracle
rng = random.Random(19)
unit = bytes(rng.randrange(40) for _ in range(30000))
cases = [(unit * 3 + bytes(rng.randrange(200) for _ in range(5000)) + unit)[:120000], b"", b"x", bytes(rng.randrange(256) for _ in range(350000)),
         open(%r, "rb").read()]
for d in cases:
    for level in (1, 9):
        assert pkg.compress(d, level) == oracle.encode(d, level), (len(d), level)
print("ok")
""" % (ROOT, os.path.join(ROOT, "tests", "golden", "sample1.ref"))
    for val in ("0", "1"):
        env = dict(os.environ, BZ_MAILBOX=val)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and out.stdout.strip().endswith("ok"), val + out.stdout[-500:] + out.stderr[-3000:]


def test_period_round_in_mixed_batches(pkg, oracle):
    """A batch in which every third block is a deep repeat (text, text, a 4 KiB paragraph repeated, ...): the period
    round is triggered by the BLOCKS that need it (round 4; rounds 1-3 looked at the batch as a whole, which such a
    batch never satisfied: seventeen full-width rounds for everybody), and a block need not be periodic from its first
    byte to its last: the period is found at anchors inside the block and the order of rotations i and i + p is read off
    the first difference behind them (k_period_find / k_period_bits).  Streams against the oracle's -- the text blocks
    go through a period round that must leave them alone -- and the round count says the round did its work.
      aligned     every unit is exactly one level-1 block (no runs of four, so RLE1 leaves the bytes alone; the last
                  byte of a unit is a separator, which is a defect at the end of every periodic block)
      straddling  RLE1 moves the cuts: blocks hold the tail of one paragraph's repeats, the head of another's and text
      defect      a deep block with one changed byte in the middle: two periodic stretches"""
    import numpy as np
    import torch
    import corpus
    rng = random.Random(5)

    def para(k):
        return bytes(rng.choice(b"abcdefgh \n") for _ in range(k))

    def clean(b):
        return bytes(corpus._no_long_runs(np.frombuffer(b, dtype=np.uint8)))

    blk = 99_981  # a level-1 block
    text = corpus.chapter(2, 1_500_000)
    ctext = clean(text)

    def unit(b):
        return b[:blk - 1] + b"\x01"

    cdeep = [unit(clean(para(4200))[:4096] * 30), unit(clean(para(800))[:777] * 140), unit(clean(para(31_000))[:30_011] * 5)]
    deep = [(para(4096) * 30)[:blk], (para(777) * 140)[:blk], (para(30_011) * 5)[:blk]]
    broken = bytearray(cdeep[0])
    broken[50_000] ^= 1
    mixes = {
        "aligned 2:1": (b"".join(unit(ctext[(2 * i) * blk:]) + unit(ctext[(2 * i + 1) * blk:]) + cdeep[i % 3] for i in range(5)), 5),
        "aligned, one deep in twelve": (b"".join(unit(ctext[i * blk:]) for i in range(6)) + cdeep[0]
                                        + b"".join(unit(ctext[i * blk:]) for i in range(6, 11)), 5),
        # (a changed byte in the middle: the chains of equal residues run through it and their halves disagree about the
        # direction -- such groups stay impure and go on doubling, as in rounds 2-3; the block with the other paragraph
        # in the same batch is finished by the round all the same)
        "aligned, defect in the middle": (unit(ctext) + bytes(broken) + cdeep[1] + unit(ctext[blk:]), 14),
        "straddling 2:1": (b"".join(text[i * 2 * blk:(i + 1) * 2 * blk] + deep[i % 3] for i in range(6)), 11),
        "straddling, one deep in twelve": (text[:6 * blk] + deep[0] + text[6 * blk:11 * blk], 11),
    }
    eng = pkg.GpuEngine(0, 32)
    try:
        for name, (data, most) in mixes.items():
            t = torch.frombuffer(bytearray(data), dtype=torch.uint8).cuda()
            cap = (pkg.encode_bound(len(data)) + 15) & ~15
            o = torch.empty(cap, dtype=torch.uint8, device="cuda")
            k = eng.encode_device(1, t.data_ptr(), len(data), o.data_ptr(), cap)
            assert bytes(o[:k].cpu().numpy()) == oracle.encode(data, 1), name
            if name.startswith("aligned"):
                assert all(b["nblock"] == blk for b in eng.block_stats()[:-1]), name
            rounds = eng.bwt_stats()["rounds"]
            assert rounds <= most, (name, rounds)  # (without the round: 14, the doubling runs to the blocks' length)
    finally:
        eng.close()
