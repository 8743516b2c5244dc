"""The block cuts without a serial chain (csrc/k_rle1.hip "kernels H"), restated in numpy and checked against the serial
rule of the reference (/root/reference/src/bzip2/encoder.rs:671-697: a block is closed by the first chunk that brings it
to 100000 * level - 19 bytes or more; a chunk is at most 255 equal bytes and leaves at most 5 bytes in the block).

What the kernels rely on:
  * block j of the whole input starts at image offset S_j with j L <= S_j <= j (L + 4);
  * S_{j+1} = the end of the first chunk that ends at or behind the target S_j + L, so one table per step with 4 j + 1
    entries, filled position by position ("a chunk that ends at c_e and holds em bytes answers the targets in
    (c_e - em, c_e]"), reproduces the serial chain exactly;
  * sixteen steps compose into one look-up.
CPU only: the GPU side is compared with the oracle's streams in tests/test_gpu_parity.py and test_gpu_sharded.py."""
import numpy as np
import pytest


def chunk_ends(data):
    """image offset behind every RLE1 chunk of `data`, and the bytes each chunk holds"""
    a = np.frombuffer(data, dtype=np.uint8)
    if a.size == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    starts = np.flatnonzero(np.concatenate(([True], a[1:] != a[:-1])))
    lens = np.diff(np.concatenate((starts, [a.size])))
    em = []
    for n in lens:  # a run is cut every 255 bytes from its start (encoder.rs:682)
        full, rest = divmod(int(n), 255)
        em += [5] * full
        if rest:
            em.append(rest if rest < 4 else 5)
    em = np.array(em, dtype=np.int64)
    return np.cumsum(em), em


def serial_cuts(ends, L):
    cuts, s = [0], 0
    for e in ends:
        if e - s >= L:
            cuts.append(int(e))
            s = int(e)
    return cuts


def table_cuts(ends, em, L, compose=16):
    total = int(ends[-1]) if ends.size else 0
    nsteps = total // L
    tabs = []
    for j in range(nsteps):
        lb = (j + 1) * L
        row = np.full(4 * j + 1, -1, dtype=np.int64)
        # every chunk end answers its targets (the kernel's loop over T)
        lo = np.searchsorted(ends, lb, side="left")
        hi = np.searchsorted(ends, lb + 4 * j + 4, side="right")
        for c_e, m in zip(ends[lo:hi], em[lo:hi]):
            for T in range(max(int(c_e - m) + 1, lb), min(int(c_e), lb + 4 * j, total) + 1):
                row[T - lb] = c_e - T
        tabs.append(row)
    comp = {}
    for g in range(0, nsteps // compose):
        j0 = g * compose
        out = np.full(4 * j0 + 1, -1, dtype=np.int64)
        for i in range(4 * j0 + 1):
            idx, ok = i, True
            for q in range(compose):
                d = tabs[j0 + q][idx]
                if d < 0:
                    ok = False
                    break
                idx += int(d)
            if ok:
                out[i] = idx - i
        comp[g] = out
    cuts, j, idx = [0], 0, 0
    while (j + 1) * L + idx <= total:
        if j % compose == 0 and j // compose in comp and comp[j // compose][idx] >= 0:
            # the blocks in between are walked again by the kernel's threads; here only the chain's end is compared
            k, i2 = j, idx
            for q in range(compose):
                i2 += int(tabs[k][i2])
                k += 1
                cuts.append(k * L + i2)
            assert i2 - idx == comp[j // compose][idx]
            j, idx = k, i2
            continue
        d = tabs[j][idx]
        assert d >= 0, "an owned target without an answer"
        idx += int(d)
        j += 1
        cuts.append(j * L + idx)
    return cuts


def _inputs():
    rng = np.random.default_rng(3)
    yield b""
    yield b"a" * 1000
    yield bytes(rng.integers(0, 256, 20000, dtype=np.uint8))
    yield bytes(rng.integers(0, 2, 30000, dtype=np.uint8))
    runs = b"".join(bytes([int(rng.integers(0, 3))]) * int(rng.choice([1, 2, 3, 4, 5, 6, 254, 255, 256, 257, 600, 3000])) for _ in range(4000))
    yield runs
    yield b"ab" * 5000 + b"c" * 100000 + bytes(rng.integers(0, 4, 5000, dtype=np.uint8))


@pytest.mark.parametrize("L", [31, 100, 1000 - 19])
def test_tables_reproduce_the_serial_chain(L):
    for data in _inputs():
        ends, em = chunk_ends(data)
        want = serial_cuts(ends, L)
        for j, s in enumerate(want):
            assert j * L <= s <= j * (L + 4)
        assert table_cuts(ends, em, L) == want


def test_chunk_model_matches_the_oracle_image(oracle):
    """the numpy chunk model above == the RLE1 image the oracle's encoder cuts into blocks (where every block ends)"""
    rng = np.random.default_rng(11)
    data = b"".join(bytes([int(rng.integers(0, 5))]) * int(rng.choice([1, 1, 2, 3, 4, 5, 9, 255, 256, 1000])) for _ in range(60000))
    ends, em = chunk_ends(data)
    L = 100000 - 19
    cuts = serial_cuts(ends, L)
    if int(ends[-1]) > cuts[-1]:
        cuts.append(int(ends[-1]))  # the unfinished last block
    image, block_ends, _in_ends, _crcs = oracle.rle1_blocks(data, 1)
    assert len(image) == int(ends[-1])
    assert cuts[1:] == block_ends
    assert table_cuts(ends, em, L) == cuts[:len(table_cuts(ends, em, L))]
