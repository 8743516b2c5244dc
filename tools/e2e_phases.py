#!/usr/bin/env python3
"""The host pipeline's phases for one bz_encode_buffer_multi call over a device list (run on the GPU box):
tools/e2e_phases.py <gib> "0,0,0,0,0,0,0,0" ...  -- wall time, MB/s, and where the call's time went
(bz_encode_buffer_last_phases: the caller's copies, the serial SPLIT and ASSEMBLE sections, ENCODE summed over the jobs)."""
import ctypes, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import corpus
pkg = importlib.import_module("rust-compression_amd")


def main():
    gib = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    lists = [[int(x) for x in a.split(",")] for a in sys.argv[2:]] or [[0]]
    h = corpus.corpus_numpy(gib << 30)
    n = h.size
    L = pkg.lib()
    for devices in lists:
        devs = (ctypes.c_int * len(devices))(*devices)
        for it in range(3):
            outp, outn = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_size_t(0)
            t0 = time.perf_counter()
            rc = L.bz_encode_buffer_multi(int(os.environ.get("BZ_LEVEL", "9")), devs, len(devices), ctypes.cast(h.ctypes.data, ctypes.c_char_p), n, ctypes.byref(outp), ctypes.byref(outn))
            dt = time.perf_counter() - t0
            assert rc == 0, rc
            L.bz_free(outp)
            ph = pkg.last_call_phases()
            print("devices x%d, %d GiB, call %d: %.1f ms, %.0f MB/s, phases %s" % (len(devices), gib, it, dt * 1e3, n / dt / 1e6, json.dumps(ph)), flush=True)
        pkg.release_cached_resources()


if __name__ == "__main__":  # (corpus_numpy starts worker processes)
    main()
