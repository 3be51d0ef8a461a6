"""Multi-GPU plan: a batch of independent .drc streams shards across the GPUs of one node with no
data-path collective.  One process per GPU; each rank builds a dsa Batch of its
own streams.  The only communication is the timing reduction of the benchmark (barrier + MAX/SUM)."""
import os


def shard_bounds(n_units, rank, world):
    """Contiguous split of n_units over `world` ranks, sizes differing by at most one."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(n_units, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def balanced_assignment(lengths, world):
    """Longest-first greedy assignment of streams to ranks by compressed length (a proxy for decode
    work).  Returns a list of index lists, one per rank; every index appears exactly once."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    loads = [0] * world
    out = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += int(lengths[i])
    for lst in out:
        lst.sort()
    return out


class Comm:
    """Thin wrapper over torch.distributed used by bench.py (backend "nccl" = RCCL on the GPUs, "gloo" in
    the CPU tests).  With world == 1 nothing is initialised."""

    def __init__(self, backend=None, device=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.device = device
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29513")
            kw = {}
            if backend == "nccl" and device is not None:
                kw["device_id"] = device
            dist.init_process_group(backend or "gloo", rank=self.rank, world_size=self.world, **kw)
            self.dist = dist
            # a second group on gloo for waits that must not touch the GPUs: an RCCL barrier is a kernel that spins on every
            # waiting rank's device, which is in the way when one rank is measuring on those devices (bench.py's pool leg)
            self.host_group = dist.new_group(backend="gloo") if (backend or "gloo") != "gloo" else None

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def host_barrier(self):
        """Barrier over gloo: the waiting ranks block on the host and leave their GPUs idle."""
        if self.dist is not None:
            self.dist.barrier(group=getattr(self, "host_group", None))

    def _reduce(self, value, op):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=op)
        return float(t.item())

    def max(self, value):
        return self._reduce(value, None if self.dist is None else self.dist.ReduceOp.MAX)

    def sum(self, value):
        return self._reduce(value, None if self.dist is None else self.dist.ReduceOp.SUM)

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None


def aggregate_throughput(units_per_rank, elapsed_s, steps, comm):
    """Benchmark contract: whole-job units per second = units all ranks processed / MAX over ranks of the
    time the timed region took."""
    worst = comm.max(elapsed_s)
    total_units = comm.sum(units_per_rank)
    return total_units * steps / worst, worst
