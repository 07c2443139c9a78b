"""One process per GPU: the only cross-rank traffic of the HEVM path is the benchmark's barrier and the reduction
of (elapsed, work) -- independent ciphertext streams never exchange data inside an op.
backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests."""
from __future__ import annotations

import os


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def streams_of_rank(n_streams: int, rank: int, world: int):
    """stream s runs on rank s mod world"""
    return [s for s in range(n_streams) if s % world == rank]


class Group:
    def __init__(self, backend: str | None = None):
        self.rank, self.local_rank, self.world = env_rank()
        self.dist = None
        self.device = "cpu"
        # DACAPO_FORCE_DIST=1: build the process group at world size 1 too (exercises the RCCL init / barrier / reduction path on a 1-GPU box)
        if self.world > 1 or os.environ.get("DACAPO_FORCE_DIST") == "1":
            import torch
            import torch.distributed as dist

            backend = backend or "nccl"
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                self.device = torch.device("cuda", self.local_rank)
                dist.init_process_group("nccl", device_id=self.device)
            else:
                dist.init_process_group(backend)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
            if str(self.device) != "cpu":
                import torch

                torch.cuda.synchronize()

    def job_totals(self, elapsed_s: float, work_units: float):
        """(max elapsed over ranks, total work over ranks): whole-job throughput = total work / max elapsed"""
        if self.dist is None:
            return elapsed_s, work_units
        import torch

        t = torch.tensor([elapsed_s], dtype=torch.float64, device=self.device)
        w = torch.tensor([work_units], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        self.dist.all_reduce(w, op=self.dist.ReduceOp.SUM)
        return float(t.item()), float(w.item())

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
