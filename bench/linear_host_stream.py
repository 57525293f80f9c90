"""PCIe-inclusive rate of the linear map step when the rows live in HOST memory (fit_linear_chunks): one pinned chunk of
`rows` x p fp32 handed over `reps` times (so the box needs one chunk of host memory, not the shard), the host -> HBM copy of
chunk i + 1 overlapped with the Gram + X'y kernels of chunk i.  python bench/linear_host_stream.py [rows] [p] [reps]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dlsa_amd
from dlsa_amd import engine

rows = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1 << 19
p = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
Xd, _ = engine.synth(20260101, 0, rows, p, kind=engine.SYNTH_GAUSSIAN, labels=False, dtype=torch.float32)
yd = engine.synth_response(20260101, 0, Xd, sigma=1.0)
Xh = torch.empty((rows, p), dtype=torch.float32).pin_memory(); Xh.copy_(Xd)
yh = torch.empty((rows,), dtype=torch.float32).pin_memory(); yh.copy_(yd)
torch.cuda.synchronize()

def wall(fn):
    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return time.perf_counter() - t, r

# the parts alone: copy, compute (device-resident chunk)
buf = torch.empty_like(Xd)
t_copy, _ = wall(lambda: [buf.copy_(Xh, non_blocking=True) for _ in range(reps)])
t_dev, _ = wall(lambda: dlsa_amd.fit_linear_chunks(((0, Xd, yd) for _ in range(reps)), p))
t_dev, mb_dev = wall(lambda: dlsa_amd.fit_linear_chunks(((0, Xd, yd) for _ in range(reps)), p))
t_host, _ = wall(lambda: dlsa_amd.fit_linear_chunks(((0, Xh, yh) for _ in range(reps)), p))
t_host, mb = wall(lambda: dlsa_amd.fit_linear_chunks(((0, Xh, yh) for _ in range(reps)), p))
same = bool(torch.equal(mb.Sig_inv, mb_dev.Sig_inv))
gb = rows * p * 4 * reps / 1e9
print(json.dumps({"rows_per_chunk": rows, "p": p, "chunks": reps, "GB": gb, "copy_only_s": t_copy, "copy_GBps": gb / t_copy,
                  "compute_only_s": t_dev, "host_stream_s": t_host, "host_stream_rows_per_s": rows * reps / t_host,
                  "host_stream_GBps": gb / t_host, "serial_sum_s": t_copy + t_dev, "blocks_equal_device_run": same}))
