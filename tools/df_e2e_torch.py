#!/usr/bin/env python3
"""df_encode_buffer from the memory bench.py hands it (a torch CPU tensor's) against a numpy array's: tools/df_e2e_torch.py [MiB]"""
import ctypes, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import corpus
pkg = importlib.import_module("rust-compression_amd")


def run(tag, h, n):
    L = pkg.lib()
    for rep in range(4):
        dp, dn = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_size_t(0)
        t0 = time.perf_counter()
        rc = L.df_encode_buffer(0, 0, ctypes.cast(h.ctypes.data, ctypes.c_char_p), n, ctypes.byref(dp), ctypes.byref(dn))
        dt = time.perf_counter() - t0
        L.bz_free(dp)
        print("%s: df_encode_buffer rc %d: %.1f ms = %.0f MB/s" % (tag, rc, dt * 1e3, n / dt / 1e6), flush=True)


def main():
    mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    dev = torch.device("cuda", 0)
    d_in = corpus.corpus_on_device(mib << 20, dev)
    n = d_in.numel()
    h_t = d_in[:n].cpu().numpy()
    print("torch cpu tensor at %#x" % h_t.ctypes.data)
    run("torch .cpu().numpy()", h_t, n)
    h_n = np.empty(n, dtype=np.uint8)
    h_n[:] = h_t
    print("numpy array at %#x" % h_n.ctypes.data)
    run("numpy copy", h_n, n)
    eng = pkg.GpuEngine(0, 1400)
    cap = (pkg.encode_bound(n) + 15) & ~15
    o = torch.empty(cap, dtype=torch.uint8, device=dev)
    eng.encode_device(9, d_in.data_ptr(), n, o.data_ptr(), cap)
    torch.cuda.synchronize()
    run("numpy copy, an encoder engine with its workspace alive", h_n, n)


if __name__ == "__main__":
    main()
