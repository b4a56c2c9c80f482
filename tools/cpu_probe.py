"""How many host cores does this box really give us?  Times a fixed integer loop on 1..N processes."""
import multiprocessing as mp, os, time

def work(_):
    t = time.perf_counter(); x = 0
    for i in range(6_000_000): x += i * i & 7
    return time.perf_counter() - t

if __name__ == "__main__":
    print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
        try: print(f, open(f).read().strip())
        except Exception as e: print(f, "n/a")
    for n in (1, 8, 16, 32, 64, 128):
        t = time.perf_counter()
        with mp.Pool(n) as p: ts = p.map(work, range(n))
        wall = time.perf_counter() - t
        print(f"procs {n:4d}: wall {wall:6.2f}s  mean task {sum(ts)/n:5.2f}s  effective cores {sum(ts[:1])*0+n*min(ts)/max(wall,1e-9):6.1f}", flush=True)
