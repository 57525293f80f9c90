"""Hunt for an intermittent hang of the two-rank gloo dry run of bench.py (ranks sharing the one GPU, a parent process holding a GPU
context as pytest does): python bench/stress_two_ranks.py [runs] [watchdog seconds]."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
hold = torch.empty(int(2e9), dtype=torch.float64, device="cuda")          # 16 GB + a context, like a pytest parent
hold.zero_(); torch.cuda.synchronize()
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
wd = sys.argv[2] if len(sys.argv) > 2 else "45"
env = dict(os.environ, DLSA_BENCH_BACKEND="gloo")
for v in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
    env.pop(v, None)
bad = 0
for i in range(runs):
    t = time.time()
    try:
        pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--watchdog-seconds", wd, "--gpus", "2", "--rows-per-gpu", "2000000",
                             "--steps", "3", "--warmup", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=200)
        rc, err = pr.returncode, pr.stderr
    except subprocess.TimeoutExpired as e:
        rc, err = -9, (e.stderr or b"").decode() if isinstance(e.stderr, bytes) else str(e.stderr)
    print("run %d rc=%d %.1f s" % (i, rc, time.time() - t), flush=True)
    if rc != 0:
        bad += 1
        print(err[-8000:], flush=True)
print("STRESS %s: %d of %d runs failed" % ("ok" if bad == 0 else "FAILED", bad, runs))
