"""N>1 path on CPU: two processes over gloo exercise spring_amd.lanes (rank-specific lane seeds,
barrier-bracketed timing, max/sum over ranks, rank-0-only reporting) exactly as bench.py drives it on
the GPUs with the nccl (RCCL) backend."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, os, sys, time
sys.path.insert(0, %r)
from spring_amd.lanes import Lanes
ln = Lanes(backend="gloo")
calls = []
def step():
    time.sleep(0.05 * (ln.rank + 1))      # rank 1 is the slow lane
    calls.append(1)
    return {"reads": 1000 * (ln.rank + 1), "seed": ln.lane_seed(11)}
el, res = ln.timed_steps(step, steps=3, warmup=1)
tot = ln.sum_over_ranks(res["reads"] * 3)
out = {"rank": ln.rank, "world": ln.world, "el": el, "calls": len(calls), "seed": res["seed"], "tot": tot}
print("RESULT " + json.dumps(out), flush=True)
ln.close()
""" % ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_lanes_over_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True, cwd=ROOT))
    outs = []
    for p in procs:
        o, _ = p.communicate(timeout=180)
        assert p.returncode == 0, o
        line = [x for x in o.splitlines() if x.startswith("RESULT ")][0]
        outs.append(json.loads(line[7:]))
    outs.sort(key=lambda d: d["rank"])
    assert [d["world"] for d in outs] == [2, 2]
    assert [d["calls"] for d in outs] == [4, 4]               # 1 warmup + exactly 3 timed steps
    assert outs[0]["seed"] == 11 and outs[1]["seed"] == 1011  # independent lanes
    assert outs[0]["el"] == outs[1]["el"] >= 0.29              # max over ranks: the slow lane (3 x 0.1 s)
    assert outs[0]["tot"] == outs[1]["tot"] == 3 * (1000 + 2000)


def test_single_process_needs_no_process_group():
    sys.path.insert(0, ROOT)
    from spring_amd.lanes import Lanes
    env_backup = {k: os.environ.pop(k, None) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    try:
        ln = Lanes()
        assert ln.world == 1 and ln.dist is None
        el, res = ln.timed_steps(lambda: 7, steps=2, warmup=0)
        assert res == 7 and el >= 0 and ln.max_over_ranks(3.5) == 3.5 and ln.lane_seed(11) == 11
    finally:
        for k, v in env_backup.items():
            if v is not None:
                os.environ[k] = v
