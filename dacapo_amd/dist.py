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

    # ---- key replication ------------------------------------------------------------------------------------------
    # Replicas serve ONE client: every rank must hold the same secret / public / evaluation keys.  Two ways:
    #   "seed"      every rank expands the same key set from the same seed on its own GPU (nothing crosses xGMI);
    #   "broadcast" rank 0's buffers are shipped: one flat broadcast per key buffer (0.09-0.4 GB each), which RCCL runs from GPU 0 over
    #               its seven point-to-point links at once -- not a ring, there is nothing to reduce.
    # Either way the ranks then compare a digest of their key material and the run aborts on a mismatch.
    def share_keys(self, digest_fn, buffers_fn=None, copy_in=None, copy_out=None, mode="seed"):
        """digest_fn() -> int; broadcast mode also needs buffers_fn() -> [(handle, words)], copy_out(handle, words) -> 1-D int64 tensor on
        self.device holding the buffer, copy_in(handle, tensor) writing it back.  Returns {"keys": "shared", "mode", "digest", "bytes"}"""
        moved = 0
        if self.dist is not None and mode == "broadcast":
            for handle, words in buffers_fn():
                t = copy_out(handle, words)
                self.dist.broadcast(t, src=0)
                if self.rank != 0:
                    copy_in(handle, t)
                moved += 8 * int(words)
                del t
        d = int(digest_fn()) & 0xFFFFFFFFFFFFFFFF
        if self.dist is not None:
            import torch

            mine = torch.tensor([d >> 32, d & 0xFFFFFFFF], dtype=torch.int64, device=self.device)
            every = [torch.zeros_like(mine) for _ in range(self.world)]
            self.dist.all_gather(every, mine)
            seen = {(int(e[0].item()) << 32) | int(e[1].item()) for e in every}
            if len(seen) != 1:
                raise RuntimeError(f"rank {self.rank}: the replicas hold different key sets ({sorted(hex(x) for x in seen)})")
        return {"keys": "shared", "mode": mode, "digest": f"{d:016x}", "broadcast_bytes": moved}

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()
