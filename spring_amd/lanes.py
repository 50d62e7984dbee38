"""Multi-GPU driver logic for independent lanes (one process per GPU, no data-path collective).

SPRING compresses one FASTQ (pair) per invocation and a sequencing run is many lanes/samples, so the
natural multi-GPU unit today is "one read set per GPU".  torch.distributed is used only for the
start/stop barriers and the max-over-ranks timing (backend "nccl" = RCCL on the GPU box, "gloo" in
the CPU tests).  The other multi-GPU mode, one read pool shared by all GPUs, is spring_amd/pool.py
(DESIGN.md section 7)."""
import os
import time


class Lanes:
    def __init__(self, backend=None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.backend = backend
        self._device = None
        if self.world > 1:
            import torch
            import torch.distributed as dist
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            backend = backend or "nccl"
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
                self._device = torch.device("cuda", self.local_rank)
            else:
                dist.init_process_group(backend)
                self._device = torch.device("cpu")
            self.dist = dist
            self.backend = backend

    def lane_seed(self, base_seed):
        """Every lane gets its own genome + reads (different seed)."""
        return base_seed + 1000 * self.rank

    def _sync_device(self):
        if self._device is not None and self._device.type == "cuda":
            import torch
            torch.cuda.synchronize()

    def barrier(self):
        self._sync_device()
        if self.dist is not None:
            self.dist.barrier()
            self._sync_device()

    def max_over_ranks(self, x):
        if self.dist is None:
            return float(x)
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64, device=self._device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x):
        if self.dist is None:
            return float(x)
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64, device=self._device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def timed_steps(self, step_fn, steps, warmup):
        """warmup untimed steps, then exactly `steps` steps bracketed by barrier + device sync on both
        sides; returns (max-over-ranks seconds, last step result)."""
        res = None
        for _ in range(warmup):
            res = step_fn()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            res = step_fn()
        self.barrier()
        el = time.perf_counter() - t0
        return self.max_over_ranks(el), res

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None
