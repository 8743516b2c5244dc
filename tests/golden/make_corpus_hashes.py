#!/usr/bin/env python3
"""Writes tests/golden/corpus_hashes.json: SHA-256 and length of the ORACLE's streams for the synthetic
corpora bench.py / bench_deflate.py encode (BASELINE.json configs[1], [2], [4]; stress T2), so that the
full-size GPU results can be compared with the reference algorithm's without running the oracle for
minutes on the GPU box.  The oracle is the C restatement of the reference (oracle/bz2_oracle.c,
oracle/deflate_oracle.c; single thread, one serial stream each).

    python3 tests/golden/make_corpus_hashes.py [--only KEY[,KEY...]] [--jobs N]

Entries (key -> corpus):  bzip2_l9_text_<N>gib (N = 1, 2, 4, 8: corpus.corpus_bytes(N GiB), the stream of
bench.py --gpus N), bzip2_l9_text_64mib, bzip2_l9_t2_1gib, deflate_text_1gib, deflate_text_64mib, deflate_text_2gib
(raw Deflate; zlib / gzip wrap the same bits; 2 GiB: more than one call of the GPU path handles in one part);
bzip2_l9_<random|dna|binary|mix|logs>_32mib: the first 32 MiB of the 256 MiB corpora of corpus.matrix_corpus;
bzip2_l1_text_256mib, bzip2_l5_text_256mib: levels 1 and 5 on the first 256 MiB of the text corpus."""
import argparse
import hashlib
import json
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "corpus_hashes.json")

KEYS = ["bzip2_l9_text_64mib", "deflate_text_64mib", "bzip2_l9_text_1gib", "deflate_text_1gib", "bzip2_l9_t2_1gib",
        "bzip2_l9_text_2gib", "bzip2_l9_text_4gib", "bzip2_l9_text_8gib", "deflate_text_2gib",
        # levels other than 9 at size (round 5): 256 MiB of the text corpus at levels 1 and 5 (2 685 and 537 blocks)
        "bzip2_l1_text_256mib", "bzip2_l5_text_256mib",
        # the corpus matrix of bench.py extra.corpora (corpus.MATRIX): the first 32 MiB of each 256 MiB corpus
        "bzip2_l9_random_32mib", "bzip2_l9_dna_32mib", "bzip2_l9_binary_32mib", "bzip2_l9_mix_32mib", "bzip2_l9_logs_32mib"]
MATRIX_BYTES = 256 << 20  # the corpora are generated at this size (their content depends on it) and cut


def make(key):
    import corpus
    from oracle import oracle
    oracle.lib()
    t0 = time.time()
    codec, rest = key.split("_", 1)
    size = rest.rsplit("_", 1)[1]
    n = int(size[:-3]) << (30 if size.endswith("gib") else 20)
    kind = rest.rsplit("_", 1)[0].split("_", 1)[1]
    if kind in corpus.MATRIX:
        data = bytes(corpus.matrix_corpus(kind, MATRIX_BYTES)[:n])
    else:
        data = corpus.stress_t2(n) if "_t2_" in key else corpus.corpus_bytes(n)
    if codec == "bzip2":
        out = oracle.encode(data, int(rest.split("_", 1)[0][1:]))  # (bzip2_l<level>_...)
    else:
        out = oracle.deflate_encode(data, 0)
    return key, {"sha256": hashlib.sha256(out).hexdigest(), "bytes": len(out), "input_bytes": n,
                 "input_sha256": hashlib.sha256(data).hexdigest(), "oracle_seconds": round(time.time() - t0, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--jobs", type=int, default=2)
    args = ap.parse_args()
    keys = [k for k in KEYS if not args.only or k in args.only.split(",")]
    have = json.load(open(OUT)) if os.path.exists(OUT) else {}
    with ProcessPoolExecutor(args.jobs) as ex:
        for key, rec in ex.map(make, keys):
            have[key] = rec
            print(key, rec, flush=True)
            json.dump(have, open(OUT, "w"), indent=1, sort_keys=True)
            open(OUT, "a").write("\n")


if __name__ == "__main__":
    main()
