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
    assert rec["supervisor"]["attempts"] == 2 and "stopped" in rec["supervisor"]["note"]
    assert rec["argv"] == ["--no-cpu-baseline"]          # the same measurement again: only the host-side leg is dropped


def test_gives_up_after_two_and_never_changes_the_schedule(tmp_path, monkeypatch, capsys):
    rc, out, runs = _run(tmp_path, monkeypatch, capsys, hangs=9)
    assert rc == 3 and runs == 2 and out == []


GRANDCHILD = r'''
import os, subprocess, sys, time
marker = sys.argv[1]
n = int(open(marker).read()) if os.path.exists(marker) else 0
open(marker, "w").write(str(n + 1))
if n == 0:
    # like a torch.distributed.run agent: the worker lives in a session of its own and inherits our stdout
    w = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(3600)"], start_new_session=True)
    open(marker + ".pid", "w").write(str(w.pid))
    time.sleep(3600)
print('{"metric": "frames_per_s", "value": 2.0}')
'''


def test_descendants_in_their_own_session_are_killed(tmp_path, monkeypatch, capsys):
    """The N-rank case: ranks started with start_new_session=True are not in the agent's process group; the supervisor must
    still end them (they hold the GPUs and the stdout pipe) and must not block on the pipe."""
    import time
    import bench
    import psutil
    script = tmp_path / "agent.py"
    script.write_text(GRANDCHILD)
    marker = tmp_path / "count"
    monkeypatch.setenv("CCVS_BENCH_TIME_LIMIT", "1.5")
    monkeypatch.setenv("CCVS_BENCH_KILL_GRACE", "1")
    args = types.SimpleNamespace(steps=1, warmup=0, schedule="pipelined", no_cpu_baseline=True)
    t0 = time.time()
    rc = bench.supervise(args, [sys.executable, str(script), str(marker)], dict(os.environ))
    assert rc == 0 and time.time() - t0 < 30
    out = [ln for ln in capsys.readouterr().out.splitlines() if ln.strip()]
    assert json.loads(out[0])["value"] == 2.0 and json.loads(out[0])["supervisor"]["attempts"] == 2
    pid = int(open(str(marker) + ".pid").read())
    time.sleep(0.2)
    assert not psutil.pid_exists(pid) or psutil.Process(pid).status() == psutil.STATUS_ZOMBIE, "the orphaned worker survived"
