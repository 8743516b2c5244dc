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
