"""The N>1 path on CPU: world_size-2 gloo run of the one-round reduce (row-shards per rank, one
all-reduce of the [Sig_inv | Sig_invMcoef | coef | count] message), checked against the oracle's
global sums.  The per-partition blocks come from the oracle here (no GPU in this tier)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, K, n, p, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from dlsa_amd import distributed
    from oracle import dlsa_oracle as orc
    r, w = distributed.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and distributed.is_distributed()
    X, y = orc.synth_logistic(99, 0, n, p)
    parts = orc.partition_rows(n, K)
    mine = distributed.owned_partitions(K, world, rank)
    S = np.zeros((p, p)); v = np.zeros(p); c = np.zeros(p)
    for k in mine:
        ck, vk, Sk = orc.logistic_model_block(X[parts[k]], y[parts[k]])
        S += Sk; v += vk; c += ck
    msg = distributed.pack_message(torch.from_numpy(S), torch.from_numpy(v), torch.from_numpy(c), len(mine))
    msg = distributed.allreduce_message(msg)
    Sg, vg, cg, cnt = distributed.unpack_message(msg, p)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), S=Sg.numpy(), v=vg.numpy(), c=cg.numpy(), cnt=cnt)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_one_round_reduce(tmp_path):
    from oracle import dlsa_oracle as orc
    K, n, p, world = 5, 3000, 6, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, K, n, p, str(tmp_path)), nprocs=world, join=True)
    X, y = orc.synth_logistic(99, 0, n, p)
    parts = orc.partition_rows(n, K)
    blocks = [orc.logistic_model_block(X[q], y[q]) for q in parts]
    ols, oneshot, S = orc.dlsa_mapred_blocks([b[0] for b in blocks], [b[1] for b in blocks], [b[2] for b in blocks])
    for rank in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        assert float(z["cnt"]) == K
        assert np.allclose(z["S"], S, rtol=1e-13, atol=0)
        theta = np.linalg.solve(z["S"], z["v"])
        assert np.max(np.abs(theta - ols)) / np.max(np.abs(ols)) < 1e-11
        assert np.allclose(z["c"] / z["cnt"], oneshot, rtol=1e-13, atol=0)


def test_shard_helpers():
    from dlsa_amd import distributed
    assert distributed.owned_partitions(7, 3, 1) == [1, 4]
    cover = []
    for r in range(4):
        lo, hi = distributed.shard_rows(10, 4, r)
        cover += list(range(lo, hi))
    assert cover == list(range(10))
    assert distributed.shard_rows(2, 4, 3) == (2, 2)
    msg = distributed.pack_message(torch.arange(4.0, dtype=torch.float64).view(2, 2),
                                   torch.tensor([5.0, 6.0], dtype=torch.float64),
                                   torch.tensor([7.0, 8.0], dtype=torch.float64), 3)
    S, v, c, k = distributed.unpack_message(distributed.allreduce_message(msg), 2)
    assert S.tolist() == [[0.0, 1.0], [2.0, 3.0]] and v.tolist() == [5.0, 6.0] and c.tolist() == [7.0, 8.0] and k == 3


def test_allreduce_message_takes_a_communicator_object():
    """One reduce path, two carriers: distributed.allreduce_message(msg, comm=...) hands the message to the communicator's
    allreduce (engine.RcclComm on a GPU host) instead of torch.distributed; anything without .allreduce is refused."""
    import torch
    from dlsa_amd import distributed

    class FakeComm:
        def __init__(self):
            self.calls = 0

        def allreduce(self, msg):
            self.calls += 1
            msg.mul_(2.0)              # "two ranks with the same message"
            return msg

    c = FakeComm()
    msg = torch.arange(7, dtype=torch.float64)
    out = distributed.allreduce_message(msg, comm=c)
    assert out is msg and c.calls == 1 and torch.equal(msg, torch.arange(7, dtype=torch.float64) * 2)
    with pytest.raises(TypeError):
        distributed.allreduce_message(msg, comm=object())
    # no communicator, no process group: the identity
    assert torch.equal(distributed.allreduce_message(msg.clone()), msg)
