import sys, importlib, time, zlib
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("rust-compression_amd")
from oracle import oracle
for d in (b"a", b"hello hello hello hello"):
    for kind in (1, 2):
        g = pkg.deflate_compress(d, kind); w = oracle.deflate_encode(d, kind)
        print(kind, len(d), g == w, g.hex(), w.hex())
