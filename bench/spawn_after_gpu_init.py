import multiprocessing as mp, torch, os, subprocess, sys
def f(q): q.put(os.getpid())
if __name__ == "__main__":
    x = torch.ones(10, device="cuda"); torch.cuda.synchronize()
    ctx = mp.get_context("spawn"); q = ctx.Queue(); p = ctx.Process(target=f, args=(q,)); p.start(); print("spawned child said", q.get(timeout=60)); p.join()
    print("subprocess:", subprocess.run([sys.executable, "-c", "print('child ok')"], capture_output=True, text=True).stdout.strip())
