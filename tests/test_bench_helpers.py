"""CPU-side checks of bench.py's plumbing that needs no GPU: the per-step statistics and the watchdog that ends
a rank which makes no progress (a peer died inside a collective) with status 5."""
import importlib.util
import os
import subprocess
import sys

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_step_stats():
    b = _bench()
    assert b.step_stats([]) is None
    s = b.step_stats([0.003, 0.001, 0.002])
    assert s == {"median": 2.0, "min": 1.0, "max": 3.0, "n": 3}
    s = b.step_stats([0.004, 0.001, 0.002, 0.003])
    assert s["median"] == 2.5 and s["n"] == 4


def test_watchdog_ends_a_stalled_process_with_status_5():
    code = ("import sys, time; sys.path.insert(0, %r); import importlib.util as u;"
            "s = u.spec_from_file_location('b', %r); m = u.module_from_spec(s); s.loader.exec_module(m);"
            "d = m.Watchdog(1.0, 3); d.beat('x'); time.sleep(0.5); d.beat('inside a collective'); time.sleep(30)"
            % (ROOT, os.path.join(ROOT, "bench.py")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=25)
    assert r.returncode == 5
    assert "rank 3 made no progress" in r.stderr and "inside a collective" in r.stderr


def test_watchdog_leaves_a_live_process_alone():
    code = ("import sys, time; sys.path.insert(0, %r); import importlib.util as u;"
            "s = u.spec_from_file_location('b', %r); m = u.module_from_spec(s); s.loader.exec_module(m);"
            "d = m.Watchdog(1.0, 0)\nfor i in range(8):\n    time.sleep(0.3); d.beat('step %%d' %% i)\nprint('done')"
            % (ROOT, os.path.join(ROOT, "bench.py")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=25)
    assert r.returncode == 0 and r.stdout.strip() == "done"
