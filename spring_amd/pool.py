"""Single-pool multi-GPU reorder: every rank holds the whole read pool and
both dictionaries, chains are sharded by contiguous id range, and one all-gather of the per-chain
proposal words per round keeps taken[] / resv[] / cursor identical on all ranks.  The output is
bit-identical to one GPU running the same total number of chains, whatever the rank count.

VirtualPool : G "ranks" as G contexts on one device, in-process exchange (how the G-independence is
              tested on a single GPU).
DistPool    : one process per GPU, the round loop and the exchange inside the library
              (spring_reorder_mg_run: ncclAllGather on the library's stream, or a host-staged all-gather
              for ranks that share a GPU).
"""
import ctypes as C

import numpy as np

from . import _lib
from .reorder import ReorderOpts, ReorderStage, _chk

STREAM_KEYS = ("order", "rc", "flag", "pos", "rlen")


def merge_rank_streams(per_rank, num_thr):
    """tid t of the whole job = chain ids ascending (chain c belongs to tid c % num_thr): every rank's tid-t records of the
    chains of the first chain group, ranks ascending, then every rank's of the second group -- a rank owns one contiguous
    slice of the chains, or (a pool that ran two groups, stats.phases = 2) one slice of each group; tid_mid[t] is where a
    rank's second part begins (= tid_off[t + 1] with one group)."""
    out = {k: [] for k in STREAM_KEYS}
    out_s = []
    tid_off, tid_off_s = [0], [0]
    for t in range(num_thr):
        for part in (0, 1):
            for r in per_rank:
                a, b = (int(r["tid_off"][t]), int(r["tid_mid"][t])) if part == 0 else (int(r["tid_mid"][t]), int(r["tid_off"][t + 1]))
                for k in STREAM_KEYS:
                    out[k].append(r[k][a:b])
                a, b = (int(r["tid_off_s"][t]), int(r["tid_mid_s"][t])) if part == 0 else (int(r["tid_mid_s"][t]), int(r["tid_off_s"][t + 1]))
                out_s.append(r["order_s"][a:b])
        tid_off.append(tid_off[-1] + sum(int(r["tid_off"][t + 1] - r["tid_off"][t]) for r in per_rank))
        tid_off_s.append(tid_off_s[-1] + sum(int(r["tid_off_s"][t + 1] - r["tid_off_s"][t]) for r in per_rank))
    res = {k: np.concatenate(v) if v else np.zeros(0) for k, v in out.items()}
    res["order_s"] = np.concatenate(out_s) if out_s else np.zeros(0, np.uint32)
    res["tid_off"] = np.array(tid_off, np.uint64)
    res["tid_off_s"] = np.array(tid_off_s, np.uint64)
    return res


class _MgStage(ReorderStage):
    def mg_begin(self, rank, world, total_chains, d_prop=None):
        _chk(self._L.spring_reorder_mg_begin(self._h, rank, world, total_chains, C.c_void_p(d_prop or 0)))

    def mg_search(self):
        _chk(self._L.spring_reorder_mg_search(self._h))

    def mg_apply(self, check_alive=True):
        a = C.c_uint32(0xFFFFFFFF)
        _chk(self._L.spring_reorder_mg_apply(self._h, int(check_alive), C.byref(a)))
        return a.value

    def mg_end(self):
        _chk(self._L.spring_reorder_mg_end(self._h))

    def check_seed_state(self):
        """Between two rounds: (bitmap words with an untaken read above the cursor, blocks below the cursor's block
        whose untaken-read count is off) -- both must be zero for the seed pick to be exact."""
        v = (C.c_uint64 * 2)()
        _chk(self._L.spring_reorder_debug_check_seed_state(self._h, v))
        return int(v[0]), int(v[1])

    def mg_slice(self):
        p, off, nb, tot = C.c_void_p(), C.c_size_t(), C.c_size_t(), C.c_size_t()
        _chk(self._L.spring_reorder_mg_slice(self._h, C.byref(p), C.byref(off), C.byref(nb), C.byref(tot)))
        return p.value, off.value, nb.value, tot.value


class VirtualPool:
    """G virtual ranks on one device.  load(stage) is called once per rank to load the same reads."""

    def __init__(self, world, total_chains, num_thr=1, **opt_kw):
        self.world, self.K, self.T = world, total_chains, num_thr
        self.check_every = 7  # rounds between checks of the seed-pick invariants (ublk[] / cursor vs the bitmap)
        self.stages = [_MgStage(ReorderOpts(num_chains=total_chains, num_thr=num_thr, **opt_kw)) for _ in range(world)]

    def run(self, load):
        L_ = _lib.lib()
        for s in self.stages:
            load(s)
            s.build_dict()
        for r, s in enumerate(self.stages):
            s.mg_begin(r, self.world, self.K)
        arr = (C.c_void_p * self.world)(*[s._h for s in self.stages])
        # (two chain groups: every group has its own view of the pool and its own seed range; the debug check below knows one)
        one_group = int(self.stages[0].stats()["phases"]) != 2
        rounds = 0
        while True:
            for s in self.stages:
                s.mg_search()
            _chk(L_.spring_reorder_mg_exchange_virtual(arr, self.world))
            alive = [s.mg_apply(True) for s in self.stages]
            rounds += 1
            if one_group and rounds % self.check_every == 0:  # what find_seed relies on, on every rank's replica
                for s in self.stages:
                    assert s.check_seed_state() == (0, 0), "seed-pick invariants broken after round %d" % rounds
            assert len(set(alive)) == 1, "ranks disagree on the number of running chains: %r" % (alive,)
            if alive[0] == 0:
                break
        per_rank = []
        for s in self.stages:
            s.mg_end()
            s.finalize()
            per_rank.append(s.streams())
        res = merge_rank_streams(per_rank, self.T)
        res["rounds"] = rounds
        res["per_rank_stats"] = [r["stats"] for r in per_rank]
        return res

    def close(self):
        for s in self.stages:
            s.close()


def host_allgather(dist):
    """spring_mg_allgather_fn over a torch.distributed group working on HOST memory (gloo): the library hands
    over its staging buffer with this rank's slice filled in; the other slices are filled here.  Used where RCCL
    cannot be (tests: two processes sharing one GPU) -- and by any caller with its own transport."""
    import torch

    def fn(buf, off, nbytes, total, user):
        try:
            arr = np.ctypeslib.as_array((C.c_uint8 * total).from_address(buf))
            t = torch.from_numpy(arr)
            mine = t[off:off + nbytes].clone()
            dist.all_gather(list(t.split(nbytes)), mine)  # the chunks are views of the staging buffer
            return 0
        except Exception:  # never unwind through the C frame
            import traceback
            traceback.print_exc()
            return 1
    return _lib.MG_ALLGATHER_FN(fn)


class GroupView:
    """The slice of the torch.distributed module PoolComm / host_allgather use, bound to one process group (e.g. a
    gloo group next to a default nccl group: the host transport moves HOST buffers)."""

    def __init__(self, dist, group, backend):
        self._d, self._g, self._b = dist, group, backend

    def get_world_size(self):
        return self._d.get_world_size(self._g)

    def get_rank(self):
        return self._d.get_rank(self._g)

    def get_backend(self):
        return self._b

    def broadcast(self, t, src=0):
        return self._d.broadcast(t, src=src, group=self._g)

    def all_gather(self, out, t):
        return self._d.all_gather(out, t, group=self._g)


class PoolComm:
    """spring_mg_comm: the exchange transport of one rank, made once per process and reused by every run.
    transport "rccl": ncclAllGather on the library's stream; the 128-byte id is made by rank 0 and broadcast through
    `dist` (an initialised torch.distributed module).  transport "host": all-gather on a host staging buffer
    through `dist` (gloo) -- for ranks that share a GPU."""

    def __init__(self, dist, device, transport="rccl"):
        import torch
        self.dist, self.device, self.transport = dist, device, transport
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self._L = _lib.lib()
        self._h = C.c_void_p()
        self._cb = None
        if transport == "rccl":
            idbuf = np.zeros(128, np.uint8)
            if self.rank == 0:
                _chk(self._L.spring_mg_rccl_unique_id(idbuf.ctypes.data))
            t = torch.from_numpy(idbuf).to(device) if dist.get_backend() == "nccl" else torch.from_numpy(idbuf)
            dist.broadcast(t, src=0)
            idbuf = np.ascontiguousarray(t.cpu().numpy())
            _chk(self._L.spring_mg_comm_create_rccl(C.byref(self._h), device.index, idbuf.ctypes.data, self.rank,
                                                    self.world))
        elif transport == "host":
            self._cb = host_allgather(dist)  # the trampoline must live as long as the communicator
            _chk(self._L.spring_mg_comm_create_host(C.byref(self._h), self._cb, None, self.rank, self.world))
        else:
            raise ValueError("transport must be 'rccl' or 'host'")

    def close(self):
        if self._h:
            self._L.spring_mg_comm_destroy(self._h)
            self._h = C.c_void_p()


class OneRankComm:
    """A 1-rank RCCL communicator without torch: the in-library exchange path (spring_reorder_mg_run) on one GPU -- tests,
    pools that fit one device.  (No torch import: a process that loads this library first and torch later would hold two
    copies of the ROCm runtime, torch's bundled one and the system's, and RCCL's second HSA handle is not initialised.)"""
    transport, world, rank = "rccl", 1, 0

    def __init__(self, device_index=0):
        import types
        self._L = _lib.lib()
        self._h = C.c_void_p()
        self._cb = None
        idbuf = np.zeros(128, np.uint8)
        _chk(self._L.spring_mg_rccl_unique_id(idbuf.ctypes.data))
        _chk(self._L.spring_mg_comm_create_rccl(C.byref(self._h), device_index, idbuf.ctypes.data, 0, 1))
        self.device = types.SimpleNamespace(index=device_index, type="cuda")

    def close(self):
        if self._h:
            self._L.spring_mg_comm_destroy(self._h)
            self._h = C.c_void_p()


class DistPool:
    """One process per GPU, ONE read pool: the whole round loop runs inside the library (spring_reorder_mg_run)."""

    def __init__(self, comm: PoolComm, total_chains, num_thr=1, **opt_kw):
        self.comm, self.K, self.T = comm, total_chains, num_thr
        dev = comm.device
        self.stage = _MgStage(ReorderOpts(device=dev.index if dev.type == "cuda" else -1, num_chains=total_chains,
                                          num_thr=num_thr, **opt_kw))

    def run(self, load):
        s = self.stage
        load(s)
        s.build_dict()
        _chk(s._L.spring_reorder_mg_run(s._h, self.comm._h, self.K))
        s.finalize()
        st = s.stats()
        self.rounds = st["rounds"]
        return st

    def streams(self):
        return self.stage.streams()

    def close(self):
        self.stage.close()
