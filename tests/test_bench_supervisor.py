"""bench.py's supervising parent (CPU): the measuring child runs under a time limit; a child that stops making progress is
killed with its process group and started again, and the relayed JSON line says so."""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r'''
import json, os, sys, time
marker = sys.argv[1]
n = int(open(marker).read()) if os.path.exists(marker) else 0
open(marker, "w").write(str(n + 1))
if n < int(sys.argv[2]):
    time.sleep(3600)            # "hangs"
print("noise on stdout")
print(json.dumps({"metric": "frames_per_s", "value": 1.0, "argv": sys.argv[3:]}))
'''


def _run(tmp_path, monkeypatch, capsys, hangs, schedule="pipelined"):
    import bench
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    marker = tmp_path / "count"
    monkeypatch.setenv("CCVS_BENCH_TIME_LIMIT", "1.5")
    args = types.SimpleNamespace(steps=1, warmup=0, schedule=schedule, no_cpu_baseline=False)
    rc = bench.supervise(args, [sys.executable, str(script), str(marker), str(hangs)], dict(os.environ))
    out = [ln for ln in capsys.readouterr().out.splitlines() if ln.strip()]
    return rc, out, int(marker.read_text())


def test_first_attempt_is_relayed_untouched(tmp_path, monkeypatch, capsys):
    rc, out, runs = _run(tmp_path, monkeypatch, capsys, hangs=0)
    assert rc == 0 and runs == 1 and len(out) == 1
    rec = json.loads(out[0])
    assert rec["value"] == 1.0 and "supervisor" not in rec


def test_hung_child_is_killed_and_restarted(tmp_path, monkeypatch, capsys):
    rc, out, runs = _run(tmp_path, monkeypatch, capsys, hangs=1)
    assert rc == 0 and runs == 2 and len(out) == 1
    rec = json.loads(out[0])
    assert rec["supervisor"]["attempts"] == 2 and "killed" in rec["supervisor"]["note"]
    assert "--schedule" not in rec["argv"] and rec["argv"] == ["--no-cpu-baseline"]


def test_third_attempt_takes_the_serial_schedule(tmp_path, monkeypatch, capsys):
    rc, out, runs = _run(tmp_path, monkeypatch, capsys, hangs=2)
    rec = json.loads(out[0])
    assert rc == 0 and runs == 3 and rec["supervisor"]["attempts"] == 3
    assert rec["argv"] == ["--no-cpu-baseline", "--schedule", "serial"] and "--schedule serial" in rec["supervisor"]["note"]


def test_gives_up_after_three(tmp_path, monkeypatch, capsys):
    rc, out, runs = _run(tmp_path, monkeypatch, capsys, hangs=9)
    assert rc == 3 and runs == 3 and out == []
