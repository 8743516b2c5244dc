"""The host pipeline of the drop-in surface (csrc/capi.hip: lanes, one worker thread per lane, the drainer, pinned
staging buffers, the resource cache -- a thousand lines of threads and queues) under ThreadSanitizer and under
AddressSanitizer + UBSan + leak detection, on the CPU box ("ASan build of host code"; sanitizers
belong on the CPU build -- the GPU pool refuses them).

capi.hip is compiled AS IT IS by g++ with -DBZ_HOST_PIPELINE_TEST, which swaps two includes: the HIP runtime calls are
served by tests/host_stub/hip_shim.h (streams are real threads with queues, so a forgotten wait is a data race the
sanitizer sees) and the device engine by tests/host_stub/stub_engine.cpp (blocks cut at chunk starts like the
reference, src/bzip2/encoder.rs:671-697; the "bit string" of a block is its bytes plus an odd number of framing bits).
tests/host_stub/host_pipeline_stress.cpp drives random write sizes, Actions (Run / Flush / Finish, repeated, with and
without input), device lists ([0] ... [0, 1, 2, 3], repeats) and chunk sizes (4 KiB ... 300 KB) and compares every stream
with the same sequence through one lane pair and one chunk; streams side by side on four threads; contexts destroyed
with jobs in flight while another thread releases the resource cache."""
import os
import subprocess

import pytest

from conftest import ROOT

STUB = os.path.join(ROOT, "tests", "host_stub")
SRC = [os.path.join(ROOT, "rust-compression_amd", "csrc", "capi.hip"), os.path.join(STUB, "stub_engine.cpp"),
       os.path.join(STUB, "host_pipeline_stress.cpp")]


def _build(tmp_path, name, flags):
    exe = str(tmp_path / name)
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-DBZ_HOST_PIPELINE_TEST", "-I", STUB] + flags + \
          ["-x", "c++"] + SRC + ["-o", exe, "-lpthread"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    return exe


def _can_run(exe):
    """(a sandbox may forbid the address-space tricks a sanitizer runtime needs)"""
    p = subprocess.run([exe, "0"], capture_output=True, text=True, timeout=120)
    return "ok" in p.stdout, p.stderr[-500:]


def test_host_pipeline_under_thread_sanitizer(tmp_path):
    exe = _build(tmp_path, "hps_tsan", ["-fsanitize=thread"])
    ok, why = _can_run(exe)
    if not ok:
        pytest.skip("ThreadSanitizer cannot run here: " + why)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    p = subprocess.run([exe, "8", "1"], env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), p.stdout[-500:] + p.stderr[-4000:]
    # the harness does see a missing wait: with the pinned buffer's event wait dropped (a test-build switch) it reports
    bad = subprocess.run([exe, "3", "1"], env=dict(env, BZ_TEST_HOST_RACE="1"), capture_output=True, text=True, timeout=1200)
    assert bad.returncode != 0 and "data race" in bad.stderr, bad.stdout[-300:] + bad.stderr[-1500:]


def test_host_pipeline_under_address_sanitizer(tmp_path):
    exe = _build(tmp_path, "hps_asan", ["-fsanitize=address,undefined"])
    ok, why = _can_run(exe)
    if not ok:
        pytest.skip("AddressSanitizer cannot run here: " + why)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1")
    p = subprocess.run([exe, "16", "100"], env=env, capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), p.stdout[-500:] + p.stderr[-4000:]
