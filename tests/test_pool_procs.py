"""The single-pool multi-GPU data path with REAL processes.

* CPU (gloo, no GPU): the host all-gather callback the library calls once per round (spring_amd.pool.host_allgather)
  fills every rank's slice of the staging buffer -- two processes, raw buffers, exactly the C callback signature.
* GPU box (one MI355X): two processes, BOTH on device 0, run one shared read pool through
  spring_reorder_mg_run with the host-staged exchange over gloo (RCCL cannot put two ranks on one device); the
  merged per-rank streams must equal the single-GPU run with the same total number of chains, byte for byte,
  and the rounds oracle.  With --gpus N the only difference is the transport (ncclAllGather on the library's
  stream instead of the staging copies)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(script, world, extra_env=None, timeout=600):
    last = ""
    for attempt in range(3):  # a rendezvous can fail when the probed port is grabbed in between: new port, once more
        port = _free_port()
        procs = []
        for r in range(world):
            env = dict(os.environ, WORLD_SIZE=str(world), RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), **(extra_env or {}))
            procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.STDOUT, text=True, cwd=ROOT))
        outs, ok = [], True
        for p in procs:
            try:
                o, _ = p.communicate(timeout=timeout)
            except subprocess.TimeoutExpired:
                p.kill()
                o, _ = p.communicate()
                ok = False
            ok = ok and p.returncode == 0
            outs.append(o)
        if ok:
            return outs
        last = "\n".join(outs)
        if "RESULT" in last and "AssertionError" in last:
            break  # a real failure, not a rendezvous problem
    raise AssertionError(last)


CB_WORKER = r"""
import ctypes as C, json, os, sys
sys.path.insert(0, %r)
import numpy as np
import torch.distributed as dist
from spring_amd.pool import GroupView, host_allgather
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
K = 1000                                   # words per rank
# the default group, then a second group behind GroupView (bench.py's fallback next to an nccl default group)
view = GroupView(dist, dist.new_group(backend="gloo"), "gloo")
ok = view.get_world_size() == world and view.get_rank() == rank and view.get_backend() == "gloo"
for rnd in range(6):                        # the library calls it once per round
    cb = host_allgather(dist if rnd < 3 else view)
    buf = np.zeros(world * K, np.uint64)
    buf[rank * K:(rank + 1) * K] = np.arange(K, dtype=np.uint64) + (rank + 1) * 1_000_000 + rnd
    rc = cb(buf.ctypes.data, rank * K * 8, K * 8, world * K * 8, None)
    want = np.concatenate([np.arange(K, dtype=np.uint64) + (r + 1) * 1_000_000 + rnd for r in range(world)])
    ok = ok and rc == 0 and np.array_equal(buf, want)
print("RESULT " + json.dumps({"rank": rank, "ok": bool(ok)}), flush=True)
dist.destroy_process_group()
""" % ROOT


def test_host_allgather_callback_two_processes(tmp_path):
    script = tmp_path / "cb_worker.py"
    script.write_text(CB_WORKER)
    outs = _spawn(script, 2, timeout=180)
    res = [json.loads([x for x in o.splitlines() if x.startswith("RESULT ")][0][7:]) for o in outs]
    assert sorted(r["rank"] for r in res) == [0, 1] and all(r["ok"] for r in res)


POOL_WORKER = r"""
import json, os, sys
sys.path.insert(0, %r)
import numpy as np
import torch
import torch.distributed as dist
from spring_amd.pool import DistPool, PoolComm
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
n, L, K, T, out = [int(os.environ[k]) for k in ("P_N", "P_L", "P_K", "P_T")] + [os.environ["P_OUT"]]
comm = PoolComm(dist, torch.device("cuda", 0), transport="host")
dp = DistPool(comm, K, num_thr=T, collect_stats=True, phases=int(os.environ.get("P_PH", "1")))
st = dp.run(lambda s: s.load_synth(n, L, n * L // 25, 3, 10000))
s = dp.streams()
np.savez(os.path.join(out, "rank%%d.npz" %% rank), **{k: s[k] for k in ("order", "rc", "flag", "pos", "rlen", "order_s", "tid_off", "tid_off_s", "tid_mid", "tid_mid_s")})
print("RESULT " + json.dumps({"rank": rank, "rounds": int(st["rounds"]), "probes": int(st["probes"]), "hits": int(st["hits"])}), flush=True)
dp.close()
comm.close()
dist.destroy_process_group()
""" % ROOT


@pytest.mark.gpu
@pytest.mark.parametrize("n,L,K,T,PH", [(200_000, 100, 64, 3, 1), (300_000, 150, 500, 1, 1), (200_000, 100, 8192, 3, 2)])
def test_one_pool_two_processes_one_gpu(tmp_path, n, L, K, T, PH):
    import spring_amd
    from oracle import pyoracle as po
    from spring_amd.pool import merge_rank_streams
    script = tmp_path / "pool_worker.py"
    script.write_text(POOL_WORKER)
    # (PH = 2: the chains in two groups -- every process owns a slice of each group, two exchanges per round through the host)
    outs = _spawn(script, 2, extra_env=dict(P_N=str(n), P_L=str(L), P_K=str(K), P_T=str(T), P_PH=str(PH), P_OUT=str(tmp_path)))
    res = sorted((json.loads([x for x in o.splitlines() if x.startswith("RESULT ")][0][7:]) for o in outs),
                 key=lambda d: d["rank"])
    assert res[0]["rounds"] == res[1]["rounds"] > 0
    per_rank = [dict(np.load(tmp_path / ("rank%d.npz" % r))) for r in range(2)]
    got = merge_rank_streams(per_rank, T)
    with spring_amd.ReorderStage(spring_amd.ReorderOpts(device=0, num_chains=K, num_thr=T, collect_stats=True, phases=PH)) as s:
        s.load_synth(n, L, n * L // 25, 3, 10000)
        want = s.run().streams()
        dna = s.download_dna()
    for k in ("order", "rc", "flag", "pos", "rlen", "order_s", "tid_off", "tid_off_s"):
        assert np.array_equal(got[k], want[k]), k
    assert res[0]["probes"] + res[1]["probes"] == want["stats"]["probes"]
    assert res[0]["hits"] + res[1]["hits"] == want["stats"]["hits"]
    read, ln = po.load_dna(dna, n, L)
    orc = po.reorder_rounds(read, ln, L, K, T) if PH == 1 else po.reorder_rounds_ph(read, ln, L, K, T)
    for k in ("order", "rc", "flag", "pos", "rlen", "order_s"):
        assert np.array_equal(got[k], orc[k]), ("oracle", k)


RCCL_WORKER = r"""
import json, os, sys
sys.path.insert(0, %r)
import numpy as np
import torch
import torch.distributed as dist
import spring_amd
from spring_amd.pool import DistPool, PoolComm
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
n, L, T = 300_000, 150, 4
comm = PoolComm(dist, torch.device("cuda", 0), transport="rccl")
ok = True
for rep in range(3):                      # the communicator is reused by several runs; the last one runs two chain groups
    K = 777 if rep < 2 else 16384         # (the library's choice at that chain count: the exchanges go through their own stream)
    dp = DistPool(comm, K, num_thr=T)
    dp.run(lambda s: s.load_synth(n, L, n * L // 25, 5 + rep, 10000))
    got = dp.streams()
    dp.close()
    with spring_amd.ReorderStage(spring_amd.ReorderOpts(device=0, num_chains=K, num_thr=T)) as s:
        s.load_synth(n, L, n * L // 25, 5 + rep, 10000)
        want = s.run().streams()
    ok = ok and all(np.array_equal(got[k], want[k]) for k in ("order", "rc", "flag", "pos", "rlen", "order_s", "tid_off", "tid_off_s"))
    ok = ok and got["stats"]["phases"] == want["stats"]["phases"] == (2 if rep == 2 else 1)
comm.close()
print("RESULT " + json.dumps({"ok": bool(ok)}), flush=True)
dist.destroy_process_group()
""" % ROOT


@pytest.mark.gpu
def test_in_library_rccl_exchange_one_rank(tmp_path):
    """The production transport of the pool -- ncclAllGather issued by the library on its own stream -- with a 1-rank
    communicator (all a single-GPU box allows): same streams as run_chains; the communicator serves two runs."""
    script = tmp_path / "rccl_worker.py"
    script.write_text(RCCL_WORKER)
    outs = _spawn(script, 1, timeout=600)
    res = json.loads([x for x in outs[0].splitlines() if x.startswith("RESULT ")][0][7:])
    assert res["ok"]
