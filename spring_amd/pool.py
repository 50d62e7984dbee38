"""Single-pool multi-GPU reorder: every rank holds the whole read pool and
both dictionaries, chains are sharded by contiguous id range, and one all-gather of the per-chain
proposal words per round keeps taken[] / resv[] / cursor identical on all ranks.  The output is
bit-identical to one GPU running the same total number of chains, whatever the rank count.

VirtualPool : G "ranks" as G contexts on one device, in-process exchange (how the G-independence is
              tested on a single GPU).
DistPool    : one process per GPU, exchange = torch.distributed.all_gather_into_tensor (backend
              "nccl" = RCCL over xGMI on the GPU box).
"""
import ctypes as C

import numpy as np

from . import _lib
from .reorder import ReorderOpts, ReorderStage, _chk

STREAM_KEYS = ("order", "rc", "flag", "pos", "rlen")


def merge_rank_streams(per_rank, num_thr):
    """tid t of the whole job = concatenation over ranks of every rank's tid-t segment (ranks own
    ascending contiguous chain ranges and chain c belongs to tid c % num_thr)."""
    out = {k: [] for k in STREAM_KEYS}
    out_s = []
    tid_off, tid_off_s = [0], [0]
    for t in range(num_thr):
        for r in per_rank:
            a, b = int(r["tid_off"][t]), int(r["tid_off"][t + 1])
            for k in STREAM_KEYS:
                out[k].append(r[k][a:b])
            a, b = int(r["tid_off_s"][t]), int(r["tid_off_s"][t + 1])
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

    def mg_slice(self):
        p, off, nb, tot = C.c_void_p(), C.c_size_t(), C.c_size_t(), C.c_size_t()
        _chk(self._L.spring_reorder_mg_slice(self._h, C.byref(p), C.byref(off), C.byref(nb), C.byref(tot)))
        return p.value, off.value, nb.value, tot.value


class VirtualPool:
    """G virtual ranks on one device.  load(stage) is called once per rank to load the same reads."""

    def __init__(self, world, total_chains, num_thr=1, **opt_kw):
        self.world, self.K, self.T = world, total_chains, num_thr
        self.stages = [_MgStage(ReorderOpts(num_chains=total_chains, num_thr=num_thr, **opt_kw)) for _ in range(world)]

    def run(self, load):
        L_ = _lib.lib()
        for s in self.stages:
            load(s)
            s.build_dict()
        for r, s in enumerate(self.stages):
            s.mg_begin(r, self.world, self.K)
        arr = (C.c_void_p * self.world)(*[s._h for s in self.stages])
        rounds = 0
        while True:
            for s in self.stages:
                s.mg_search()
            _chk(L_.spring_reorder_mg_exchange_virtual(arr, self.world))
            alive = [s.mg_apply(True) for s in self.stages]
            rounds += 1
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


class DistPool:
    """One process per GPU.  `dist` is an initialised torch.distributed module; the proposal buffer
    is a torch tensor so the collective can run on it directly."""

    def __init__(self, dist, device, total_chains, num_thr=1, check_every=16, **opt_kw):
        import torch
        self.torch, self.dist, self.device = torch, dist, device
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.K, self.T, self.check_every = total_chains, num_thr, check_every
        self.stage = _MgStage(ReorderOpts(device=device.index if device.type == "cuda" else -1,
                                          num_chains=total_chains, num_thr=num_thr, **opt_kw))
        self.prop = torch.zeros(total_chains, dtype=torch.int64, device=device)

    def run(self, load):
        torch, dist, s = self.torch, self.dist, self.stage
        load(s)
        s.build_dict()
        s.mg_begin(self.rank, self.world, self.K, self.prop.data_ptr())
        per = self.K // self.world
        mine = self.prop[self.rank * per:(self.rank + 1) * per]
        send = torch.empty_like(mine)
        rounds = 0
        while True:
            s.mg_search()                       # library stream, synchronised on return
            send.copy_(mine)
            dist.all_gather_into_tensor(self.prop, send)
            torch.cuda.synchronize(self.device)  # the library's stream reads prop next
            rounds += 1
            check = rounds % self.check_every == 0
            alive = s.mg_apply(check)
            if check and alive == 0:
                break
        s.mg_end()
        s.finalize()
        self.rounds = rounds
        return s.stats()

    def close(self):
        self.stage.close()
