"""The engines the one-shot calls over host buffers keep between calls (round 4: bz_decode_buffer, bz_dec_*,
df_encode_buffer, df_enc_* park ONE engine per device; bz_release_cached_resources frees it).  An engine made by one
kind of call serves the next call of ANOTHER kind -- decode, Deflate, streaming contexts, in any order --, results
are the oracle's every time, and the calling thread's current device is left as it was.
(/root/reference/src/bzip2/decoder.rs:583-612 and src/deflate/encoder.rs are what the calls stand for; the cache is the
library's own.)"""
import pytest

pytestmark = pytest.mark.gpu


def _text(n, seed):
    import corpus
    return corpus.chapter(seed, max(n, 1 << 16))[:n]


def test_one_shot_calls_share_the_parked_engine(pkg, oracle):
    data = _text(2_500_000, 5) + b"\x00" * 70_000 + _text(300_000, 6)
    z = oracle.encode(data, 9)
    z1 = oracle.encode(data[:400_000], 1)
    df = oracle.deflate_encode(data[:1_200_000], 0)
    for round_ in range(2):
        # decode -> Deflate -> decode (another stream) -> Deflate again: the engine goes from call to call
        back, verdict = pkg.decompress(z)
        assert verdict == 0 and back == data
        assert pkg.deflate_compress(data[:1_200_000], pkg.DEFLATE) == df
        back, verdict = pkg.decompress(z1)
        assert verdict == 0 and back == data[:400_000]
        assert pkg.deflate_compress(data[:1_200_000], pkg.DEFLATE) == df
        # a truncated stream: the bytes in front of the error, then the verdict -- and the engine is still good
        back, verdict = pkg.decompress(z[:len(z) // 2])
        assert verdict != 0 and data.startswith(back)
        back, verdict = pkg.decompress(z)
        assert verdict == 0 and back == data
        # the streaming contexts take the same engine
        dec = pkg.BZip2Decoder()
        assert dec.decode_all(z) == data
        del dec
        enc = pkg.Inflater()
        assert enc.encode_all(data[:1_200_000]) == df
        del enc
        back, verdict = pkg.decompress(z)
        assert verdict == 0 and back == data
        pkg.release_cached_resources()  # ... and the next round starts without one


def test_decode_buffer_hands_out_memory_free_takes(pkg, oracle):
    """bz_decode_buffer's result lives in posix_memalign'ed memory (2 MiB-aligned, huge pages asked for) from 4 MiB on and
    in malloc'ed memory below: bz_free (= free) takes both; an empty result is a 1-byte allocation."""
    import ctypes as C
    L = pkg.lib()
    for n in (0, 1, 5_000, 6_000_000):
        data = _text(n, 9) if n else b""
        z = oracle.encode(data, 9)
        for _ in range(2):
            outp, outn = C.POINTER(C.c_uint8)(), C.c_size_t(0)
            assert L.bz_decode_buffer(0, z, len(z), C.byref(outp), C.byref(outn)) == 0
            assert outn.value == n and C.string_at(outp, outn.value) == data
            if n >= (4 << 20):
                assert C.addressof(outp.contents) % (2 << 20) == 0
            L.bz_free(outp)
    pkg.release_cached_resources()


def test_peer_copy_selftest_on_the_boxes_devices(pkg):
    """bz_peer_copy_selftest (round 5; bench.py's preflight at N > 1): a list that names the one device twice copies
    nothing (peer_access -1), bad arguments are refused, and the calling thread's current device is left as it was.  (Two
    different devices need a second GPU: the driver's node; the pairs' arithmetic runs under the sanitizers against the HIP
    shim, tests/host_stub/host_pipeline_stress.cpp.)"""
    import ctypes as C
    import torch
    L = pkg.lib()
    torch.cuda.set_device(0)
    devs = (C.c_int * 2)(0, 0)
    peer = (C.c_int * 2)(9, 9)
    ms = (C.c_double * 2)(-2.0, -2.0)
    assert L.bz_peer_copy_selftest(devs, 2, 1 << 20, peer, ms) == 0
    assert list(peer) == [-1, -1] and list(ms) == [0.0, 0.0]
    assert L.bz_peer_copy_selftest(devs, 0, 1 << 20, None, None) == pkg.BZ_E_PARAM
    assert L.bz_peer_copy_selftest((C.c_int * 1)(77), 1, 1 << 20, None, None) == pkg.BZ_E_PARAM
    assert L.bz_peer_copy_selftest(devs, 2, 0, None, None) == pkg.BZ_E_PARAM
    n = torch.cuda.device_count()
    if n >= 2:  # (a multi-GPU box: the real thing, a round trip per neighbour pair)
        devs = (C.c_int * n)(*range(n))
        peer = (C.c_int * n)()
        ms = (C.c_double * n)()
        assert L.bz_peer_copy_selftest(devs, n, 1 << 20, peer, ms) == 0
        assert all(p in (0, 1) for p in peer) and all(t > 0 for t in ms)
    assert torch.cuda.current_device() == 0


def test_streaming_decoder_reads_behind_the_end_wait_for_the_worker(pkg, oracle):
    """Round 5: bz_dec_end returns at once and bz_dec_read waits behind it.  A stream large enough for the worker thread
    (chunks of 4 MiB and more are decoded beside the caller): written in one piece, ended, then read in small pieces until
    the verdict -- every byte, then 0; a corrupted copy: the bytes in front of the error, then the error
    (/root/reference/src/bzip2/decoder.rs:583-612: the iterator yields exactly that sequence)."""
    import bz2
    import ctypes as C
    data = _text(30_000_000, 11)
    z = bz2.compress(data, 9)
    assert len(z) > (5 << 20)
    L = pkg.lib()
    for corrupt in (False, True):
        zz = bytearray(z)
        if corrupt:
            zz[len(zz) * 3 // 4] ^= 0x55
        want, verdict = oracle.decode(bytes(zz))
        h = C.c_void_p()
        assert L.bz_dec_create(C.byref(h), 0) == 0
        assert L.bz_dec_write(h, bytes(zz), len(zz)) == 0
        rc_end = L.bz_dec_end(h)
        assert rc_end in (0, verdict)
        buf = (C.c_uint8 * (3 << 20))()
        got = bytearray()
        while True:
            k = L.bz_dec_read(h, buf, len(buf))
            if k <= 0:
                break
            got += bytes(buf[:k])
        L.bz_dec_destroy(h)
        assert k == verdict and bytes(got) == want, (corrupt, k, verdict, len(got), len(want))
        assert (verdict == 0) == (not corrupt)


def _stream_through(L, zz, piece=1 << 20, read_cap=3 << 20, read_between=True):
    """the loop a host runs: write in pieces, read what has come between the writes, end, read to the verdict"""
    import ctypes as C
    h = C.c_void_p()
    assert L.bz_dec_create(C.byref(h), 0) == 0
    buf = (C.c_uint8 * read_cap)()
    got = bytearray()
    zb = bytes(zz)
    for i in range(0, len(zb), piece):
        assert L.bz_dec_write(h, zb[i:i + piece], len(zb[i:i + piece])) == 0
        while read_between:
            k = L.bz_dec_read(h, buf, len(buf))
            if k <= 0:
                break
            got += bytes(buf[:k])
    L.bz_dec_end(h)
    while True:
        k = L.bz_dec_read(h, buf, len(buf))
        if k <= 0:
            break
        got += bytes(buf[:k])
    L.bz_dec_destroy(h)
    return bytes(got), k


def test_streaming_decoder_two_lanes(pkg, oracle, monkeypatch):
    """Round 5: chunks of 4 MiB and more go to the context's two lanes -- the next chunk's scan and Huffman stage run
    beside the rebuilding of this chunk's blocks, a lane whose chunk in front ends with an error gives up before it hands
    out a byte.  Several chunks per stream here (BZ_DEC_CHUNK = 4 MiB): a clean stream, two streams in one file with the
    seam inside a chunk, a block corrupted in the second chunk (the third is in flight on the other lane by then), a
    stream cut off in mid-block, a context destroyed with chunks still in flight: the bytes and the verdict of the oracle's
    decoder every time (/root/reference/src/bzip2/decoder.rs:583-612: bytes in front of an error, then the error)."""
    import ctypes as C
    monkeypatch.setenv("BZ_DEC_CHUNK", str(4 << 20))
    monkeypatch.setenv("BZ_DEC_FIRST_CHUNK", str(4 << 20))
    data = _text(80_000_000, 21)
    z = pkg.compress(data, 9)
    assert len(z) > (14 << 20)  # four chunks at least
    L = pkg.lib()
    # clean, with and without reads between the writes; one lane gives the same
    for between in (True, False):
        got, verdict = _stream_through(L, z, read_between=between)
        assert verdict == 0 and got == data
    monkeypatch.setenv("BZ_DEC_LANES", "1")
    got, verdict = _stream_through(L, z)
    assert verdict == 0 and got == data
    monkeypatch.delenv("BZ_DEC_LANES")
    # two streams, the seam somewhere inside the third chunk
    cut = 40_500_123
    zz = pkg.compress(data[:cut], 9) + pkg.compress(data[cut:], 5)
    got, verdict = _stream_through(L, zz)
    assert verdict == 0 and got == data
    # a block of the second chunk corrupted / the stream cut off inside the third chunk / junk behind the stream
    for name, bad in (("flip", bytes(z[:6 << 20]) + bytes([z[6 << 20] ^ 0x41]) + bytes(z[(6 << 20) + 1:])),
                      ("cut", bytes(z[:(9 << 20) + 12345])),
                      ("junk", bytes(z) + b"\x00" * 5)):
        want, st = oracle.decode(bad)
        assert st != 0, name
        for between in (True, False):
            got, verdict = _stream_through(L, bad, read_between=between)
            assert verdict == st and got == want, (name, between, verdict, st, len(got), len(want))
    # a context that is dropped while its chunks are in flight (nothing read): no hang, the next context is fine
    h = C.c_void_p()
    assert L.bz_dec_create(C.byref(h), 0) == 0
    assert L.bz_dec_write(h, bytes(z), len(z)) == 0
    L.bz_dec_destroy(h)
    got, verdict = _stream_through(L, z[:5 << 20] + z[5 << 20:])
    assert verdict == 0 and got == data
