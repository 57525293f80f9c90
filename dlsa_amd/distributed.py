"""Row-shard data parallelism: one process per GPU, one collective.

The reference's only communication is the sum over partitions of the p x (p+2) blocks
(dlsa.py:30-34: Spark groupby-sum + toPandas).  Here every rank first sums its own partitions
on the device and the ranks exchange ONE contiguous fp64 message
    [ Sig_inv (p*p) | Sig_invMcoef (p) | coef (p) | n_partitions (1) ]
with a single all-reduce (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).
Nothing here touches the compute kernels, so it is testable without a GPU.
"""
import os

import torch


def is_distributed():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun)."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or dist.is_initialized():
        return int(os.environ.get("RANK", "0")), world
    rank = int(os.environ["RANK"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world


def owned_partitions(num_partitions, world, rank):
    """GPU g owns the logical partitions {k : k % G == g} (the analogue of
    repartition(K, "partition_id"), logistic_dlsa.py:295)."""
    return [k for k in range(num_partitions) if k % world == rank]


def shard_rows(n, world, rank):
    """Contiguous row range [lo, hi) of a rank when the n rows are dealt in equal blocks."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def pack_message(Sig_inv_sum, Sig_invMcoef_sum, coef_sum, n_partitions):
    p = Sig_invMcoef_sum.numel()
    msg = torch.empty(p * p + 2 * p + 1, dtype=torch.float64, device=Sig_inv_sum.device)
    msg[: p * p] = Sig_inv_sum.reshape(-1)
    msg[p * p: p * p + p] = Sig_invMcoef_sum
    msg[p * p + p: p * p + 2 * p] = coef_sum
    msg[-1] = float(n_partitions)
    return msg


def unpack_message(msg, p):
    return (msg[: p * p].view(p, p), msg[p * p: p * p + p], msg[p * p + p: p * p + 2 * p], float(msg[-1].item()))


def allreduce_message(msg, comm=None):
    """The algorithm's one round of communication (in place; identity when not distributed).

    ONE reduce path with two carriers of the same RCCL collective: `comm` = an `engine.RcclComm` (the C ABI's
    dlsa_allreduce_f64, for hosts that do not run torch.distributed), else the initialised torch.distributed group
    (backend "nccl" IS RCCL; gloo in the CPU tests).  Both are a sum all-reduce of the same contiguous fp64 buffer;
    tests/test_gpu_distributed.py checks that they return identical bits on the same message."""
    if comm is not None:
        if not hasattr(comm, "allreduce"):
            raise TypeError("comm must be an engine.RcclComm (or offer .allreduce(msg)), got %r" % (type(comm),))
        comm.allreduce(msg)
    elif is_distributed():
        import torch.distributed as dist
        dist.all_reduce(msg, op=dist.ReduceOp.SUM)
    return msg
