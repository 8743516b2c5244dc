"""Deterministic synthetic "repeating text" corpus of BASELINE.json configs[1]/[2].

A chapter is CHAPTER_BYTES of pseudo-English: 4096 lowercase pseudo-words (2-12 letters, English
letter frequencies) drawn Zipf(s=1.1), single spaces, '.' / ',' with p = 0.08 / 0.06, newline at
>= 72 columns, a blank line + 4-space indent every ~12 lines, a 40 x '-' rule every ~200 lines and an
occasional 300-character rule (exercises RLE1 runs >= 4 and the 255 cut).  The corpus is the chapter
repeated (period 16 MiB > block, so no intra-block mega-repeats); GiB k uses chapter seed + k.

Randomness: a counter-based splitmix64 stream (vectorised with numpy), seed 0x9E3779B97F4A7C15.
Nothing external is read; generation is excluded from every timed region.
"""
import numpy as np

SEED = 0x9E3779B97F4A7C15
CHAPTER_BYTES = 16 << 20
_LETTERS = "etaoinshrdlcumwfgypbvkjxqz"
_FREQ = np.array([12.7, 9.1, 8.2, 7.5, 7.0, 6.7, 6.3, 6.1, 6.0, 4.3, 4.0, 2.8, 2.8, 2.4, 2.4, 2.2, 2.0, 2.0,
                  1.9, 1.5, 1.0, 0.8, 0.15, 0.15, 0.1, 0.07])


def _splitmix64(idx, seed):
    with np.errstate(over="ignore"):
        z = (idx.astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _uniform(n, seed, stream):
    r = _splitmix64(np.arange(n, dtype=np.uint64) + np.uint64(stream << 40), seed)
    return (r >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def _vocab(seed):
    u = _uniform(4096 * 13, seed, 1).reshape(4096, 13)
    lens = 2 + (u[:, 0] * 11).astype(int)
    cdf = np.cumsum(_FREQ / _FREQ.sum())
    letters = np.searchsorted(cdf, u[:, 1:], side="right").clip(0, 25)
    return [bytes(ord(_LETTERS[c]) for c in letters[i, :lens[i]]) for i in range(4096)]


def chapter(k=0, nbytes=CHAPTER_BYTES):
    """Chapter number k as bytes (deterministic)."""
    seed = (SEED + k) & 0xFFFFFFFFFFFFFFFF
    vocab = _vocab(SEED)  # one vocabulary for all chapters
    nwords = nbytes // 4 + 1024
    ranks = np.arange(1, 4097, dtype=np.float64)
    p = ranks ** -1.1
    cdf = np.cumsum(p / p.sum())
    widx = np.searchsorted(cdf, _uniform(nwords, seed, 2), side="right").clip(0, 4095)
    punct = _uniform(nwords, seed, 3)
    out = bytearray()
    col = 0
    line = 0
    i = 0
    while len(out) < nbytes:
        w = vocab[widx[i]]
        out += w
        col += len(w)
        pu = punct[i]
        if pu < 0.08:
            out += b"."
            col += 1
        elif pu < 0.14:
            out += b","
            col += 1
        i += 1
        if col >= 72:
            out += b"\n"
            col = 0
            line += 1
            if line % 200 == 0:
                out += b"-" * (300 if (line // 200) % 7 == 0 else 40) + b"\n"
            if line % 12 == 0:
                out += b"\n    "
                col = 4
        else:
            out += b" "
            col += 1
    return bytes(out[:nbytes])


def corpus_bytes(total_bytes, first_chapter=0):
    """Host bytes of a corpus: chapter(first_chapter + g) repeated within GiB g."""
    out = bytearray()
    g = 0
    while len(out) < total_bytes:
        ch = chapter(first_chapter + g)
        want = min(1 << 30, total_bytes - len(out))
        reps = (want + len(ch) - 1) // len(ch)
        out += (ch * reps)[:want]
        g += 1
    return bytes(out)


def corpus_on_device(total_bytes, device, first_chapter=0):
    """The same corpus as a torch uint8 tensor on `device` (chapters are uploaded once and tiled there)."""
    import torch
    parts = []
    done = 0
    g = 0
    while done < total_bytes:
        ch = torch.frombuffer(bytearray(chapter(first_chapter + g)), dtype=torch.uint8).to(device)
        want = min(1 << 30, total_bytes - done)
        reps = (want + ch.numel() - 1) // ch.numel()
        parts.append(ch.repeat(reps)[:want])
        done += want
        g += 1
    out = parts[0] if len(parts) == 1 else torch.cat(parts)
    # The engines run on their own streams: the corpus must be complete -- and the pieces it was made of back in
    # torch's allocator for good -- before a pointer to it (or to a buffer allocated after it) is handed to one.
    torch.cuda.synchronize(device)
    return out


def chapters(ks, workers=8):
    """{k: chapter(k)} for the chapter numbers `ks`, generated side by side in fresh processes (a chapter is ~3 s of
    Python; spawned, not forked: the caller may have initialised a GPU)."""
    ks = sorted(set(ks))
    if len(ks) <= 1 or workers <= 1:
        return {k: chapter(k) for k in ks}
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    with ProcessPoolExecutor(min(workers, len(ks)), mp_context=mp.get_context("spawn")) as ex:
        return dict(zip(ks, ex.map(chapter, ks)))


def corpus_numpy(total_bytes, first_chapter=0, workers=8):
    """The corpus as one numpy uint8 array (no second copy: 8 GiB stay 8 GiB)."""
    out = np.empty(total_bytes, dtype=np.uint8)
    ngib = (total_bytes + (1 << 30) - 1) >> 30
    chs = chapters([first_chapter + g for g in range(ngib)], workers)
    for g in range(ngib):
        ch = np.frombuffer(chs[first_chapter + g], dtype=np.uint8)
        lo, hi = g << 30, min(total_bytes, (g + 1) << 30)
        for p in range(lo, hi, len(ch)):
            k = min(len(ch), hi - p)
            out[p:p + k] = ch[:k]
    return out


def slice_chapters(off, nbytes, first_chapter=0):
    """chapter numbers the corpus bytes [off, off + nbytes) are made of"""
    if nbytes <= 0:
        return []
    return [first_chapter + g for g in range(off >> 30, ((off + nbytes - 1) >> 30) + 1)]


def slice_on_device(off, nbytes, device, first_chapter=0, chs=None):
    """Bytes [off, off + nbytes) of the corpus as a torch uint8 tensor on `device`, without the rest of it."""
    import torch
    parts = []
    end = off + nbytes
    for g in range(off >> 30, ((end - 1) >> 30) + 1 if nbytes else 0):
        raw = chs[first_chapter + g] if chs is not None else chapter(first_chapter + g)
        ch = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        lo, hi = max(off, g << 30), min(end, (g + 1) << 30)
        first = (lo - (g << 30)) % ch.numel()
        reps = (first + (hi - lo) + ch.numel() - 1) // ch.numel()
        parts.append(ch.repeat(reps)[first:first + (hi - lo)])
    out = (parts[0].clone() if len(parts) == 1 else torch.cat(parts)) if parts else torch.empty(0, dtype=torch.uint8, device=device)
    torch.cuda.synchronize(device)
    return out


def stress_t2(total_bytes):
    """Stress variant T2: a 4 KiB paragraph repeated -- deep LCPs."""
    para = chapter(0, 1 << 16)[:4096]
    reps = (total_bytes + 4095) // 4096
    return (para * reps)[:total_bytes]


# ---- the corpus matrix (bench.py extra.corpora): inputs that are NOT Zipf text ------------------------
# All deterministic (the same counter-based splitmix64 streams), all numpy-vectorised: 256 MiB in a few seconds.
MATRIX = ("random", "dna", "binary", "mix", "logs")


def _rand_u64(n, stream):
    return _splitmix64(np.arange(n, dtype=np.uint64) + np.uint64(stream << 40), SEED)


def _no_long_runs(a):
    """`a` without the bytes that would make a run of four or more equal bytes (RLE1 then leaves the data as it is:
    n input bytes are n block bytes, so a unit of 899 981 bytes is exactly one level-9 block)."""
    eq = a[1:] == a[:-1]
    # byte i is dropped when it equals the three bytes in front of it
    drop = np.zeros(a.size, dtype=bool)
    drop[3:] = eq[2:] & eq[1:-1] & eq[:-2]
    return a[~drop]


def matrix_corpus(name, nbytes, golden_dir=None):
    """One of MATRIX as a numpy uint8 array of `nbytes` bytes.
      random  uniform random bytes: 256 symbols, wide keys (11/11/10-bit digits), nothing to compress
      dna     four symbols {A,C,G,T}, skewed i.i.d. base with 20 % of the positions covered by copies of earlier
              segments (100 .. 5000 bytes, 1 % point mutations): 2-bit symbols, 8 of them per key, deep groups
      binary  the reference's binary fixtures data/sample2.ref + sample3.ref + sample4.ref (libbzip2's test files,
              tests/golden/) tiled, one byte per 4 KiB of every copy changed: 200+ symbols, runs, RLE1 at work
      mix     level-9 blocks 2:1: two blocks of text, one block that is a 4 KiB paragraph repeated (stress T2), each unit
              exactly one block (899 981 bytes without runs of four: the cut falls on the unit's end)
      logs    fixed-width log lines (timestamp, host, service, level, template, numbers): long shared line prefixes"""
    n = int(nbytes)
    if name == "random":
        return _rand_u64((n + 7) // 8, 11).view(np.uint8)[:n].copy()
    if name == "dna":
        u = _rand_u64((n + 3) // 4, 12).view(np.uint16)[:n]  # 16 bits per symbol, four symbols per draw
        idx = (u >= np.uint16(0.3 * 65536)).astype(np.uint8)
        idx += u >= np.uint16(0.5 * 65536)
        idx += u >= np.uint16(0.7 * 65536)
        out = np.frombuffer(b"ACGT", dtype=np.uint8)[idx]
        del u, idx
        nseg = max(1, n // 12_000)  # segments of mean ~2.5 KB over 20 % of the positions
        r = _rand_u64(nseg * 3, 13)
        lens = (100 + r[0::3] % np.uint64(4900)).astype(np.int64)
        dst = (r[1::3] % np.uint64(max(n - 5000, 1))).astype(np.int64)
        src = (r[2::3] % np.uint64(max(n - 5000, 1))).astype(np.int64)
        order = np.argsort(dst, kind="stable")
        for i in order:  # (copies from EARLIER text, in position order)
            d, s_, k = int(dst[i]), int(src[i]), int(lens[i])
            if s_ + k <= d:
                out[d:d + k] = out[s_:s_ + k]
        mut = (_rand_u64(n // 100 + 1, 14) % np.uint64(n)).astype(np.int64)
        out[mut] = np.frombuffer(b"ACGT", dtype=np.uint8)[(mut * 7 + 3) & 3]
        return out
    if name == "binary":
        import os
        gd = golden_dir or os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden")
        base = np.concatenate([np.fromfile(os.path.join(gd, "sample%d.ref" % i), dtype=np.uint8) for i in (2, 3, 4)])
        out = np.resize(base, n)
        pos = np.arange(0, n, 4096, dtype=np.int64)
        r = _rand_u64(pos.size, 15)
        pos = np.minimum(pos + (r % np.uint64(4096)).astype(np.int64), n - 1)
        out[pos] ^= ((r >> np.uint64(20)) % np.uint64(255) + np.uint64(1)).astype(np.uint8)
        return out
    if name == "mix":
        unit = 899_981
        nunits = (n + unit - 1) // unit
        ntext = nunits - nunits // 3
        text = _no_long_runs(np.frombuffer(chapter(40, min(ntext * unit + (1 << 20), 40 << 20)), dtype=np.uint8))
        text = np.resize(text, ntext * unit)  # (the text repeats after 40 MB: farther than a block)
        para = _no_long_runs(np.frombuffer(chapter(41, 1 << 16), dtype=np.uint8))
        out = np.empty(nunits * unit, dtype=np.uint8)
        t = 0
        for k in range(nunits):
            if k % 3 == 2:
                p0 = (k // 3 * 4096) % (para.size - 4096)  # another paragraph for every deep block
                out[k * unit:(k + 1) * unit] = np.resize(para[p0:p0 + 4096], unit)
            else:
                out[k * unit:(k + 1) * unit] = text[t * unit:(t + 1) * unit]
                t += 1
        return out[:n].copy()
    if name == "logs":
        width = 128
        nl = (n + width - 1) // width
        r = _rand_u64(nl, 16)
        lines = np.full((nl, width), ord(" "), dtype=np.uint8)
        lines[:, width - 1] = ord("\n")

        def put_digits(col, value, digits):
            v = value.astype(np.uint64)
            for d in range(digits - 1, -1, -1):
                lines[:, col + d] = (v % np.uint64(10)).astype(np.uint8) + ord("0")
                v //= np.uint64(10)

        def put_choice(col, idx, table, w):
            tab = np.frombuffer(b"".join(t.ljust(w)[:w] for t in table), dtype=np.uint8).reshape(len(table), w)
            lines[:, col:col + w] = tab[idx % len(table)]

        i = np.arange(nl, dtype=np.uint64)
        ms = i * np.uint64(7) + (r % np.uint64(5))  # a clock that moves on by a few ms per line
        lines[:, 0:11] = np.frombuffer(b"2026-10-03T", dtype=np.uint8)
        put_digits(11, (ms // np.uint64(3_600_000)) % np.uint64(24), 2)
        lines[:, 13] = ord(":")
        put_digits(14, (ms // np.uint64(60_000)) % np.uint64(60), 2)
        lines[:, 16] = ord(":")
        put_digits(17, (ms // np.uint64(1000)) % np.uint64(60), 2)
        lines[:, 19] = ord(".")
        put_digits(20, ms % np.uint64(1000), 3)
        lines[:, 23] = ord("Z")
        put_choice(25, ((r >> np.uint64(8)) % np.uint64(64)).astype(np.int64),
                   [b"host-%04d" % h for h in range(64)], 10)
        put_choice(36, ((r >> np.uint64(16)) % np.uint64(8)).astype(np.int64),
                   [b"frontend[1187]:", b"frontend[1188]:", b"authd[402]:", b"store[77]:", b"store[78]:", b"sched[9]:",
                    b"gateway[2210]:", b"gateway[2211]:"], 16)
        lvl = ((r >> np.uint64(24)) % np.uint64(16)).astype(np.int64)
        put_choice(53, np.where(lvl < 12, 0, np.where(lvl < 15, 1, 2)), [b"INFO", b"WARN", b"ERROR"], 6)
        put_choice(60, ((r >> np.uint64(28)) % np.uint64(12)).astype(np.int64),
                   [b"request served path=/api/v1/items", b"request served path=/api/v1/users", b"cache miss key=session",
                    b"cache hit key=session", b"token refreshed for user", b"connection reset by peer",
                    b"request served path=/static/app", b"slow query table=orders", b"replica lag above limit",
                    b"request served path=/api/v1/cart", b"job finished queue=default", b"job started queue=default"], 36)
        lines[:, 97:100] = np.frombuffer(b"id=", dtype=np.uint8)
        put_digits(100, (r >> np.uint64(32)) % np.uint64(100_000_000), 8)
        lines[:, 109:113] = np.frombuffer(b"dur=", dtype=np.uint8)
        put_digits(113, (r >> np.uint64(44)) % np.uint64(10_000), 4)
        lines[:, 117:119] = np.frombuffer(b"ms", dtype=np.uint8)
        return lines.reshape(-1)[:n].copy()
    raise KeyError(name)


def t2_slice(off, nbytes):
    """bytes [off, off + nbytes) of the stress corpus T2 (stress_t2 of any length that holds them)"""
    para = chapter(0, 1 << 16)[:4096]
    first = off % 4096
    reps = (first + nbytes + 4095) // 4096
    return (para * reps)[first:first + nbytes]
