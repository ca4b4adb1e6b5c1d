"""Probe the host CPU budget of the box (affinity, cgroup quota) and matmul throughput vs thread count."""
import os
import time

import torch

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(p, open(p).read().strip())
    except OSError as e:
        print(p, "n/a")
print("loadavg", open("/proc/loadavg").read().strip())
a, b = torch.randn(2048, 2048), torch.randn(2048, 2048)
for nt in (4, 8, 16, 32, 64, 128):
    if nt > (os.cpu_count() or 1):
        break
    torch.set_num_threads(nt)
    (a @ b)
    t0 = time.time()
    for _ in range(5):
        (a @ b)
    dt = (time.time() - t0) / 5
    print(f"threads {nt:4d}: matmul 2048^3 {dt*1e3:8.1f} ms  {2*2048**3/dt/1e9:8.1f} GFLOP/s", flush=True)
