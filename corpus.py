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
