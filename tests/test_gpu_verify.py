"""The encoder's self-check (bz_enc_set_verify / bz_gpu_engine_set_verify / BZ_VERIFY=1) and the structural checks in
front of it.  The reference's sequential encoder (/root/reference/src/bzip2/encoder.rs:224-291) cannot write a stream
that does not decode to its input; the GPU pipeline has look-back words, ticket counters and stream-ordered clears, and
round 3 saw one wrong stream in six from a clear issued on the wrong stream.  Here faults are INJECTED (switches the
library reads from the environment, so every case runs in a process of its own):

  BZ_TEST_CORRUPT=1        a wrong origPtr for the first block of a batch: a well-formed stream of other bytes
  BZ_TEST_STALE_TICKETS=1  ticket counters of a fused pass that were not cleared: no tile of that pass is sorted, every
                           counter ends at twice its share (the old ">=" check accepted that; "==" does not)
  BZ_TEST_LATE_CLEAR=1     the outcome of round 3's fault (look-back words cleared late): a sorted order that is not the
                           block's -- two entries of the last column swapped (replaying the fault itself ends in GPU
                           memory faults: misdirected tiles break the invariants the later passes index with)
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

_RUNNER = r"""
import hashlib, importlib, json, os, sys
sys.path.insert(0, %(root)r)
import corpus
pkg = importlib.import_module("rust-compression_amd")
cfg = json.loads(%(cfg)r)
data = corpus.chapter(3, cfg["bytes"])
out = {"streams": []}
if cfg["mode"] == "engine":
    import torch
    dev = torch.device("cuda", 0)
    d_in = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
    cap = (pkg.encode_bound(len(data)) + 15) & ~15
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    eng = pkg.GpuEngine(0, 64)
    eng.set_verify(cfg["verify"])
    for _ in range(cfg["repeats"]):
        k = eng.encode_device(cfg["level"], d_in.data_ptr(), len(data), d_out.data_ptr(), cap)
        out["streams"].append(hashlib.sha256(bytes(d_out[:k].cpu().numpy())).hexdigest())
    out["verify"] = eng.verify_stats()
    out["fallbacks"] = eng.bwt_stats()["fused_fallbacks"]
    out["blocks"] = len(eng.block_stats())
else:
    for _ in range(cfg["repeats"]):
        enc = pkg.BZip2Encoder(cfg["level"], devices=cfg["devices"], verify=cfg["verify"])
        z = enc.encode_all(data)
        out["streams"].append(hashlib.sha256(z).hexdigest())
        st = enc.verify_stats()
        out.setdefault("verify", []).append(st)
        del enc
        if cfg.get("release"):
            pkg.release_cached_resources()
print("RESULT " + json.dumps(out))
"""


def _run(cfg, env_extra, timeout=900):
    env = dict(os.environ, **env_extra)
    env.pop("BZ_VERIFY", None)
    p = subprocess.run([sys.executable, "-c", _RUNNER % {"root": ROOT, "cfg": json.dumps(cfg)}], env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [x for x in p.stdout.splitlines() if x.startswith("RESULT ")]
    assert line, p.stdout[-1000:] + p.stderr[-2000:]
    return json.loads(line[0][7:]), p.stderr


def _want(oracle, nbytes, level):
    import hashlib
    import corpus
    return hashlib.sha256(oracle.encode(corpus.chapter(3, nbytes), level)).hexdigest()


def test_verify_passes_clean_streams_and_counts_blocks(oracle):
    cfg = {"mode": "engine", "bytes": 3_000_000, "level": 1, "verify": True, "repeats": 2}
    res, _ = _run(cfg, {})
    assert res["streams"] == [_want(oracle, 3_000_000, 1)] * 2
    assert res["blocks"] >= 30
    assert res["verify"]["blocks_checked"] == 2 * res["blocks"]
    assert res["verify"]["jobs_redone"] == 0 and res["verify"]["jobs_failed_again"] == 0 and res["fallbacks"] == 0
    assert res["verify"]["nanoseconds"] > 0


def test_silent_corruption_is_caught_only_by_the_self_check(oracle):
    want = _want(oracle, 2_000_000, 9)
    cfg = {"mode": "engine", "bytes": 2_000_000, "level": 9, "verify": False, "repeats": 1}
    off, _ = _run(cfg, {"BZ_TEST_CORRUPT": "1"})
    assert off["streams"] != [want]                      # the fault is real, and nothing structural notices it
    assert off["fallbacks"] == 0
    on, err = _run(dict(cfg, verify=True), {"BZ_TEST_CORRUPT": "1"})
    assert on["streams"] == [want]                       # checked, encoded again without look-back passes, checked again
    assert on["verify"]["jobs_redone"] == 1 and on["verify"]["jobs_failed_again"] == 0
    assert on["fallbacks"] == 1
    assert "self-check" in err
    # ... and through the host pipeline (BZ_VERIFY=1 in the environment: the one-shot call has no handle)
    host, _ = _run({"mode": "context", "bytes": 2_000_000, "level": 9, "verify": True, "devices": [0], "repeats": 1},
                   {"BZ_TEST_CORRUPT": "1"})
    assert host["streams"] == [want] and host["verify"][0]["jobs_redone"] == 1


def test_stale_ticket_counters_fail_the_exact_check(oracle):
    """Counters at twice their share satisfy ">= tiles" (rounds 1-3) although no tile of the pass was sorted; the exact
    check sends the batch to the three-kernel passes.  No self-check involved."""
    want = _want(oracle, 2_000_000, 9)
    res, err = _run({"mode": "engine", "bytes": 2_000_000, "level": 9, "verify": False, "repeats": 1},
                    {"BZ_TEST_STALE_TICKETS": "1"})
    assert res["streams"] == [want]
    assert res["fallbacks"] == 1
    assert "fused radix pass" in err


def test_late_clear_outcome_never_leaves_the_library_with_the_self_check_on(oracle):
    """The outcome of round 3's fault (a sorted order that is not the block's: BZ_TEST_LATE_CLEAR=1 swaps two entries of
    the last column): without the self-check the stream is well formed and wrong, with it the job is encoded again
    and the stream is the oracle's."""
    nbytes = 24_000_000
    want = _want(oracle, nbytes, 9)
    cfg = {"mode": "engine", "bytes": nbytes, "level": 9, "verify": False, "repeats": 2}
    off, _ = _run(cfg, {"BZ_TEST_LATE_CLEAR": "1"})
    assert all(s != want for s in off["streams"]) and off["fallbacks"] == 0
    on, err = _run(dict(cfg, verify=True), {"BZ_TEST_LATE_CLEAR": "1"})
    assert on["streams"] == [want] * 2
    assert on["verify"]["jobs_redone"] == 1 and on["verify"]["jobs_failed_again"] == 0 and on["fallbacks"] == 1
    assert "self-check" in err


def test_lane_creation_stress_with_verify(oracle):
    """Thirty contexts over the device list [0, 0, 0] (six lanes whose engines are created while the other lanes are
    sorting: chunks of 4 MiB, eleven jobs per stream), each destroyed and its parked resources released, self-check
    on: every stream is the oracle's and no job ever failed its check."""
    nbytes = 40 << 20
    want = _want(oracle, nbytes, 9)
    cfg = {"mode": "context", "bytes": nbytes, "level": 9, "verify": True, "devices": [0, 0, 0], "repeats": 30, "release": True}
    res, _ = _run(cfg, {"BZ_ENC_CHUNK_MIB": "4"}, timeout=1400)
    assert res["streams"] == [want] * 30
    assert all(v["jobs_redone"] == 0 and v["jobs_failed_again"] == 0 for v in res["verify"])
    assert all(v["blocks_checked"] >= 40 for v in res["verify"])
