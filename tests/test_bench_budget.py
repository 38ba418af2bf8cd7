"""CPU: bench.py's wall-clock budget (bench_support.Budget) -- a child process that never returns, a leg without time
left and a leg that hangs inside the process must all leave the run its one JSON line (driver_run_s grew
to 247 s of 600 with 1200 s of child time-outs and no guard)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_a_hung_child_is_killed_at_what_the_budget_has_left():
    sys.path.insert(0, ROOT)
    from bench_support import Budget
    b = Budget(4.0)
    t0 = time.time()
    r = b.child("cli_files.plain_fasta", [sys.executable, "-c", "import time; time.sleep(1000)"], 600, 1)
    assert r is None and time.time() - t0 < 8.0
    assert b.dropped and b.dropped[0]["leg"] == "cli_files.plain_fasta" and "killed_after_s" in b.dropped[0]
    # nothing is left now: the next leg does not start at all, and says so
    assert not b.want("dump_load_cli", 25) and b.dropped[1]["leg"] == "dump_load_cli"
    assert b.child("gzip_inflate", [sys.executable, "-c", "print(1)"], 300, 25) is None
    rec = b.record()
    assert rec["budget_s"] == 4.0 and len(rec["dropped"]) == 3


def test_a_child_that_finishes_hands_over_its_line():
    sys.path.insert(0, ROOT)
    from bench_support import Budget
    b = Budget(60.0)
    rc, out = b.child("x", [sys.executable, "-c", "print('{\"a\": 1}')"], 30, 1)
    assert rc == 0 and json.loads(out.decode().strip().splitlines()[-1]) == {"a": 1} and not b.dropped


def test_the_watchdog_prints_the_line_when_a_leg_hangs_in_process(tmp_path):
    prog = tmp_path / "hang.py"
    prog.write_text(
        "import sys, time, json, os\n"
        "sys.path.insert(0, %r)\n"
        "from bench_support import Budget\n"
        "out = {'metric': 'm', 'value': 1.0}\n"
        "done = []\n"
        "def emit(hard=False):\n"
        "    if done: return\n"
        "    done.append(1); out['budget'] = b.record(); out['budget']['hard_stop'] = hard\n"
        "    os.write(1, (json.dumps(out) + '\\n').encode())\n"
        "b = Budget(1.0)\n"
        "b.arm(1.0, emit)\n"
        "time.sleep(1000)      # the leg that never comes back\n" % ROOT)
    t0 = time.time()
    r = subprocess.run([sys.executable, str(prog)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 0 and time.time() - t0 < 20
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["value"] == 1.0 and j["budget"]["hard_stop"] is True
