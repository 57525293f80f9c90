"""Two ranks sharing the one GPU of the test box (gloo transport, CUDA tensors): exercises the whole
row-shard path -- per-rank map step on the HIP engine, ONE all-reduce of the block message, WLS
combine + LARS on every rank -- against the single-process oracle."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, K, n, p, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import torch
    import dlsa_amd
    from dlsa_amd import distributed
    from oracle import dlsa_oracle as orc
    torch.cuda.set_device(0)
    distributed.init_from_env(backend="gloo")
    X, y = orc.synth_logistic(314, 0, n, p)
    parts = orc.partition_rows(n, K)
    mine = distributed.owned_partitions(K, world, rank)
    rows = np.concatenate([parts[k] for k in mine])
    offs = np.concatenate([[0], np.cumsum([len(parts[k]) for k in mine])])
    mb = dlsa_amd.fit_logistic_partitions(torch.from_numpy(X[rows]).cuda(), torch.from_numpy(y[rows]).cuda(),
                                          part_offsets=offs, fit_intercept=True)
    out = dlsa_amd.dlsa_mapred(mb)                  # local sum + all-reduce + WLS
    sel = dlsa_amd.dlsa(out.iloc[:, 2:], out["beta_byOLS"], n, fit_intercept=True)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), ols=out["beta_byOLS"].to_numpy(),
             oneshot=out["beta_byONESHOT"].to_numpy(), S=out.iloc[:, 2:].to_numpy(), bic=sel["beta_byBIC"].to_numpy())
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_one_gpu_full_path(tmp_path):
    import torch.multiprocessing as mp
    from oracle import dlsa_oracle as orc
    K, n, p, world = 6, 24000, 12, 2
    mp.spawn(_worker, args=(world, _free_port(), K, n, p, str(tmp_path)), nprocs=world, join=True)
    X, y = orc.synth_logistic(314, 0, n, p)
    parts = orc.partition_rows(n, K)
    blocks = [orc.logistic_model_block(X[q], y[q], True) for q in parts]
    ols, oneshot, S = orc.dlsa_mapred_blocks([b[0] for b in blocks], [b[1] for b in blocks], [b[2] for b in blocks])
    _, by_bic, _ = orc.dlsa(S, ols, n, fit_intercept=True)
    for rank in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
        assert rel(z["ols"], ols) < 1e-10
        assert rel(z["oneshot"], oneshot) < 1e-10
        assert rel(z["S"], S) < 1e-10
        assert rel(z["bic"], by_bic) < 1e-8
