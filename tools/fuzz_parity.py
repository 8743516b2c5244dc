#!/usr/bin/env python3
"""Randomised parity fuzzing on the GPU box: encoder vs oracle (bit-exact streams) and decoder vs
oracle (bytes + verdict), valid and corrupted inputs.  usage: fuzz_parity.py [seconds] [seed]"""
import bz2
import importlib
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle
if os.environ.get("FUZZ_REPLAY"):
    # FUZZ_REPLAY=<case>: no GPU -- the oracle stands in for the library, the input of that case is written to
    # /tmp/fuzz_case.bin (the generators draw from one seeded stream: a failing case of a GPU run is found again here)
    class _Stub:
        compress = staticmethod(lambda d, lvl: oracle.encode(d, lvl))
        decompress = staticmethod(lambda z: oracle.decode(z, max(1 << 20, 300 * len(z) + 1024)))
    pkg = _Stub()
else:
    pkg = importlib.import_module("rust-compression_amd")

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
flavour = sys.argv[3] if len(sys.argv) > 3 else "small"   # small | big | stream | dstream | deflate
rng = random.Random(seed)


def gen_copies():
    """round 4: what the period round is about -- a stretch repeated at some distance, with foreign bytes in front and
    behind, a few changed bytes, two different stretches in one block, distances beyond half a block"""
    k = rng.choice([2, 4, 16, 64, 256])
    n = rng.choice([30000, 99981, 120000, 250000])
    out = bytearray(bytes(rng.randrange(k) for _ in range(rng.choice([0, 1, 3, 50, 5000]))))
    for _ in range(rng.choice([1, 1, 2])):
        L = rng.choice([2, 7, 100, 3000, 4096, 40000, 60000, 90000])
        base = bytes(rng.randrange(k) for _ in range(L))
        reps = max(2, min(n // L, rng.choice([2, 3, 30, 1000])))
        piece = bytearray(base * reps + base[:rng.randrange(L)])
        for _ in range(rng.choice([0, 0, 1, 2, 5])):
            piece[rng.randrange(len(piece))] ^= 1 + rng.randrange(3)
        out += piece
        out += bytes(rng.randrange(k) for _ in range(rng.choice([0, 0, 2, 700])))
    return bytes(out[:n + rng.randrange(0, 5)])


def gen_links():
    """round 6: what the link rounds and the split of short-period stretches are about -- a unit that repeats inside itself,
    copied two to ten times with a byte changed every few hundred to few thousand bytes (groups of 2 .. 8 and beyond, in mixed
    order, a changed byte in a middle copy), and stretches that repeat 1 .. 9 bytes, cut by changed bytes (up- and downward)"""
    k = rng.choice([4, 16, 64, 200])
    n = rng.choice([30000, 99981, 120000, 250000])

    def rnd(m):
        return bytes(rng.randrange(k) for _ in range(m))

    def touched(b, every):
        b = bytearray(b)
        for pos in range(rng.randrange(every), len(b), every):
            b[pos] = (b[pos] + rng.choice([1, 2, 255, 128])) & 255
        return bytes(b)

    out = bytearray(rnd(rng.choice([0, 1, 3, 500])))
    while len(out) < n:
        what = rng.randrange(3)
        if what == 0:
            inner = rnd(rng.choice([100, 1000, 5000]))
            unit = inner + rnd(rng.choice([0, 50, 2000])) + touched(inner, rng.choice([64, 700, 4096]))
            for _ in range(rng.choice([2, 3, 4, 10])):
                out += touched(unit, rng.choice([300, 2048, 4096]))
        elif what == 1:
            p = rnd(rng.choice([1, 2, 4, 5, 9]))
            out += touched(p * rng.choice([200, 3000, 20000]), rng.choice([97, 1900, 4096]))
        else:
            out += rnd(rng.choice([10, 3000]))
    return bytes(out[:n + rng.randrange(0, 5)])


def gen():
    if os.environ.get("FUZZ_ONLY") == "links":
        return gen_links()
    kind = rng.randrange(12)
    if kind >= 10:
        return gen_links()
    if kind >= 8:
        return gen_copies()
    n = rng.choice([0, 1, 2, 3, 5, 17, 255, 256, 257, 1000, 4096, 50000, 99980, 99981, 99982, 100010, 150000, 250000])
    n = max(0, n + rng.randrange(-3, 4)) if n > 10 else n
    k = rng.choice([1, 2, 3, 4, 8, 16, 64, 200, 256])
    if kind == 0:
        return bytes(rng.randrange(k) for _ in range(n))
    if kind == 1:  # runs
        out = bytearray()
        while len(out) < n:
            out += bytes([rng.randrange(k)]) * rng.choice([1, 2, 3, 4, 5, 6, 254, 255, 256, 257, 300, 1000])
        return bytes(out[:n])
    if kind == 2:  # periodic
        p = bytes(rng.randrange(k) for _ in range(rng.choice([1, 2, 3, 7, 64, 255, 1000])))
        return (p * (n // max(len(p), 1) + 1))[:n]
    if kind == 3:  # periodic with a defect
        p = bytes(rng.randrange(k) for _ in range(rng.choice([2, 5, 100])))
        b = bytearray((p * (n // len(p) + 1))[:n])
        if b:
            b[rng.randrange(len(b))] ^= 1
        return bytes(b)
    if kind == 4:  # text-like
        words = [bytes(rng.randrange(97, 97 + 20) for _ in range(rng.randrange(1, 9))) for _ in range(50)]
        out = bytearray()
        while len(out) < n:
            out += rng.choice(words) + b" "
        return bytes(out[:n])
    if kind == 5:  # sorted / reverse sorted
        b = sorted(bytes(rng.randrange(k) for _ in range(n)))
        return bytes(b if rng.random() < 0.5 else b[::-1])
    if kind == 6:  # two-level repetition
        base = bytes(rng.randrange(k) for _ in range(rng.choice([10, 100, 1000])))
        return (base * 3 + bytes(rng.randrange(k) for _ in range(5))) * max(1, n // (3 * len(base) + 5))
    return bytes(rng.randrange(256) for _ in range(min(n, 20000)))


def big():
    """multi-block inputs (level 9 blocks are 900 kB): a few MB assembled from the generators"""
    parts = []
    want = rng.choice([900000, 1800010, 2500000, 3600000])
    while sum(map(len, parts)) < want:
        p = gen()
        parts.append(p * rng.choice([1, 1, 3, 10]) if len(p) < 300000 else p)
    return b"".join(parts)[:want + rng.randrange(-20, 20)]


def stream_case():
    """the streaming context: random write / Action sequences against the oracle's BZip2Encoder mirror"""
    lvl = rng.choice([1, 1, 2, 9])
    enc, ora = pkg.BZip2Encoder(lvl), oracle.Encoder(lvl)
    d = gen() + gen()
    pos = 0
    while pos < len(d):
        step = rng.choice([1, 7, 1000, 50000, 99981, 150000])
        act = rng.choice([pkg.Action.RUN, pkg.Action.RUN, pkg.Action.FLUSH])
        piece = d[pos:pos + step]
        pos += step
        if enc.encode_all(piece, act) != ora.encode_iter(piece, int(act)):
            return False
    last = rng.choice([pkg.Action.FINISH, pkg.Action.FINISH, pkg.Action.FLUSH])
    return enc.encode_all(b"", last) == ora.encode_iter(b"", int(last))


def dstream_case():
    """the streaming DEcoder: random write sizes and decode thresholds against the oracle's one-shot verdict"""
    parts = [gen() for _ in range(rng.choice([1, 1, 2, 3]))]
    z = b"".join(bz2.compress(p, rng.choice([1, 1, 9])) for p in parts)
    mode = rng.randrange(4)
    if mode == 1 and len(z) > 4:
        z = z[:rng.randrange(len(z))]
    elif mode == 2 and z:
        zb = bytearray(z)
        zb[rng.randrange(len(zb))] ^= 1 << rng.randrange(8)
        z = bytes(zb)
    elif mode == 3:
        z += b"tail"
    cap = max(1 << 20, 300 * len(z) + 1024)
    want, st = oracle.decode(z, cap)
    if st == -100:
        return True
    os.environ["BZ_DEC_CHUNK"] = str(rng.choice([1, 1000, 20000, 100000, 1 << 30]))
    dec = pkg.BZip2Decoder()
    got, code, pos = bytearray(), 0, 0
    try:
        while pos < len(z):
            step = rng.choice([1, 3, 100, 5000, 70000, 1 << 20])
            dec.write(z[pos:pos + step])
            pos += step
            got += dec.read_available()
        got += dec.decode_all(b"")
    except pkg.BZip2Error as e:
        got += e.partial
        code = e.code
    return (bytes(got), code) == (want, st)


def deflate_flush_case():
    """an Inflater stream fed iterator by iterator with Run / Flush / Finish in between, against the oracle's"""
    d = big()[:rng.choice([70000, 200000, 400000])] if rng.random() < 0.2 else gen()
    dict_ = gen()[:rng.choice([0, 0, 100, 40000])] if rng.random() < 0.3 else b""
    enc, ref = pkg.Inflater(dict_=dict_), oracle.DeflateEncoder(dict_)
    got, pos, trace = bytearray(), 0, []
    while True:
        k = rng.choice([0, 1, 3, 100, 5000, 65535, 65536, 70000, len(d)])
        piece = d[pos:pos + k]
        pos += len(piece)
        act = 2 if (pos >= len(d) and rng.random() < 0.5) else rng.choice([0, 1, 1])
        trace.append((len(piece), act))
        enc.write(piece)
        enc.end(act)
        got += enc.read_all()
        ref.feed(piece, act)
        if act == 2:
            break
    if bytes(got) != ref.output():
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        open(os.path.join(ROOT, "gpurun_out", "fuzz_deflate_flush_fail.bin"), "wb").write(d)
        print("deflate flush mismatch: n", len(d), "dict", len(dict_), "pieces", trace[:40])
        return False
    return True


def deflate_wrapper_case():
    """ZlibEncoder / GZipEncoder driven with Run or Flush: the container ends at the inner encoder's first None
    (zlib/encoder.rs:138-150) -- against the oracle's iterator-level restatement, piece by piece"""
    d = big()[:rng.choice([70000, 140000, 200000, 400000])] if rng.random() < 0.5 else gen()
    kind = rng.choice([1, 2])
    dict_ = gen()[:rng.choice([100, 40000])] if (kind == 1 and rng.random() < 0.2) else b""
    enc = (pkg.ZlibEncoder if kind == 1 else pkg.GZipEncoder)(dict_=dict_)
    ref = oracle.WrapperEncoder(kind, dict_)
    cut = rng.randrange(len(d) + 1)
    acts = [rng.randrange(3), rng.randrange(3)]
    for piece, act in ((d[:cut], acts[0]), (d[cut:], acts[1])):
        enc.write(piece)
        enc.end(act)
        if enc.read_all() != ref.encode_iter(piece, act):
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            open(os.path.join(ROOT, "gpurun_out", "fuzz_deflate_wrapper_fail.bin"), "wb").write(d)
            print("deflate wrapper mismatch: n", len(d), "kind", kind, "cut", cut, "actions", acts, "dict", len(dict_))
            return False
    return True


def deflate_case():
    """Deflate / zlib / gzip streams against the oracle; every so often a multi-block input"""
    import zlib
    if rng.random() < 0.35:
        return deflate_flush_case()
    if rng.random() < 0.2:
        return deflate_wrapper_case()
    d = big()[:rng.choice([70000, 200000, 700000])] if rng.random() < 0.15 else gen()
    kind = rng.randrange(3)
    got = pkg.deflate_compress(d, kind)
    want = oracle.deflate_encode(d, kind)
    if got != want:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        open(os.path.join(ROOT, "gpurun_out", "fuzz_deflate_fail.bin"), "wb").write(d)
        first = next((i for i in range(min(len(got), len(want))) if got[i] != want[i]), -1)
        print("deflate mismatch: n", len(d), "kind", kind, "len got/want", len(got), len(want), "first diff", first)
        return False
    # the reference writes HDIST = 0 and NO distance code length for a dynamic block without matches
    # (deflate/encoder.rs:431-436, 449-451): such a stream is the reference's, RFC 1951 decoders reject or
    # misread it.  Every other stream must inflate back with zlib.
    e = oracle.DeflateEncoder()
    e.feed(d, oracle.ACTION_FINISH)
    if any(b[2] == 2 and b[0] == b[1] for b in e.blocks()):
        global quirks
        quirks += 1
        return True
    return zlib.decompress(got, [-15, 15, 31][kind]) == d


t0 = time.time()
cases = enc_ok = dec_ok = quirks = 0
while flavour == "deflate" and time.time() - t0 < budget:
    cases += 1
    if not deflate_case():
        print("DEFLATE MISMATCH seed", seed, "case", cases)
        sys.exit(1)
if flavour == "deflate":
    print("fuzz ok: %d deflate streams in %.0f s (seed %d); %d of them match-free dynamic blocks zlib rejects (reference quirk)"
          % (cases, time.time() - t0, seed, quirks))
    sys.exit(0)
while flavour == "dstream" and time.time() - t0 < budget:
    cases += 1
    if not dstream_case():
        print("DECODE-STREAM MISMATCH seed", seed, "case", cases)
        sys.exit(1)
if flavour == "dstream":
    print("fuzz ok: %d streaming decodes in %.0f s (seed %d)" % (cases, time.time() - t0, seed))
    sys.exit(0)
while flavour == "stream" and time.time() - t0 < budget:
    cases += 1
    if not stream_case():
        print("STREAM MISMATCH seed", seed, "case", cases)
        sys.exit(1)
if flavour == "stream":
    print("fuzz ok: %d streaming sequences in %.0f s (seed %d)" % (cases, time.time() - t0, seed))
    sys.exit(0)
if flavour == "big":
    _small = gen
    gen_case = big
else:
    gen_case = gen
while time.time() - t0 < budget:
    d = gen_case()
    lvl = rng.choice([1, 1, 1, 2, 9]) if flavour == "small" else rng.choice([9, 9, 5])
    cases += 1
    if os.environ.get("FUZZ_REPLAY") and cases == int(os.environ["FUZZ_REPLAY"]):
        open("/tmp/fuzz_case.bin", "wb").write(d)
        print("case", cases, "len", len(d), "level", lvl, "written to /tmp/fuzz_case.bin")
        sys.exit(0)
    if os.environ.get("FUZZ_VERBOSE"):
        print("case", cases, "len", len(d), "level", lvl, "t %.1f" % (time.time() - t0), flush=True)
    want = oracle.encode(d, lvl)
    if os.environ.get("FUZZ_VERBOSE"):
        print("  oracle done t %.1f" % (time.time() - t0), flush=True)
    got = pkg.compress(d, lvl)
    if got != want:
        open("/tmp/fuzz_fail_enc.bin", "wb").write(d)
        print("ENCODE MISMATCH seed", seed, "case", cases, "len", len(d), "level", lvl)
        sys.exit(1)
    enc_ok += 1
    z = want if rng.random() < 0.5 else bz2.compress(d, lvl)
    mode = rng.randrange(5)
    if mode == 1 and len(z) > 4:
        z = z[:rng.randrange(len(z))]
    elif mode == 2 and z:
        zb = bytearray(z)
        for _ in range(rng.choice([1, 1, 2, 5])):
            zb[rng.randrange(len(zb))] ^= 1 << rng.randrange(8)
        z = bytes(zb)
    elif mode == 3:
        z = z + bz2.compress(gen()[:5000], rng.choice([1, 9])) + (b"" if rng.random() < 0.7 else b"junk")
    cap = max(1 << 20, 300 * len(z) + 1024)
    w = oracle.decode(z, cap)
    if w[1] == -100:  # the oracle's own buffer limit (a block that expands forever in the reference): skip
        continue
    g = pkg.decompress(z)
    if g != w:
        open("/tmp/fuzz_fail_dec.bin", "wb").write(z)
        print("DECODE MISMATCH seed", seed, "case", cases, "len", len(z), "mode", mode, "got", (len(g[0]), g[1]), "want", (len(w[0]), w[1]))
        sys.exit(1)
    dec_ok += 1
print("fuzz ok: %d cases, %d encodes, %d decodes in %.0f s (seed %d)" % (cases, enc_ok, dec_ok, time.time() - t0, seed))
