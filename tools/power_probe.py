"""Socket power and engine clock under a sustained GEMM load, next to an idle reading: the evidence behind "the power cap bounds the GEMM
main loop" (profiles/r03_power_ceiling.txt).  usage (GPU box): python tools/power_probe.py
Runs three loads for ~6 s each (8192^3 f16 GEMM; 16384x4096x1024 GEMM; an HBM-bound LayerNorm) and samples `rocm-smi` from a side
thread every 0.5 s; prints the samples and their means."""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--json"], capture_output=True, text=True,
                             timeout=10).stdout
        import json
        d = json.loads(out)
        card = d[sorted(d)[0]]
        keep = {k: v for k, v in card.items() if any(s in k.lower() for s in ("power", "sclk", "mclk"))}
        return keep
    except Exception as e:                                    # noqa: BLE001
        return {"error": repr(e)}


def sample_while(fn, seconds, label):
    stop, rows = [False], []

    def poll():
        while not stop[0]:
            rows.append(smi())
            time.sleep(0.5)

    th = threading.Thread(target=poll)
    th.start()
    t0 = time.time()
    n = 0
    c0 = K.clock_sample()
    while time.time() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    c1 = K.clock_sample()
    torch.cuda.synchronize()
    dt = time.time() - t0
    stop[0] = True
    th.join()
    print(f"## {label}: {n} launches in {dt:.1f} s ({dt / n * 1e6:.1f} us each), effective engine clock {K.effective_clock_mhz(c0, c1):.0f} MHz")
    for r in rows:
        print("   ", r)
    return n, dt


def main():
    dev = torch.device("cuda:0")
    print("## idle:", smi())
    M = 8192
    x = (torch.rand(M, M, device=dev) * 2 - 1).half()
    w = (torch.rand(M, M, device=dev) * 2 - 1).half()
    y = torch.empty(M, M, device=dev, dtype=torch.float16)
    n, dt = sample_while(lambda: K.gemm(x, w, y, M, M, M, ldx=M, ldw=M), 6.0, "NT 8192^3 f16")
    print(f"   -> {2.0 * M ** 3 * n / dt / 1e12:.0f} TFLOP/s sustained")
    x2 = (torch.rand(16384, 1024, device=dev) * 2 - 1).half()
    w2 = (torch.rand(4096, 1024, device=dev) * 2 - 1).half()
    y2 = torch.empty(16384, 4096, device=dev, dtype=torch.float16)
    n, dt = sample_while(lambda: K.gemm(x2, w2, y2, 16384, 4096, 1024, ldx=1024, ldw=1024), 6.0, "NT 16384x4096x1024 f16")
    print(f"   -> {2.0 * 16384 * 4096 * 1024 * n / dt / 1e12:.0f} TFLOP/s sustained")
    a = torch.randn(16384, 1024, device=dev)
    g, b = torch.ones(1024, device=dev), torch.zeros(1024, device=dev)
    sample_while(lambda: K.layernorm_fwd(a, g, b, torch.float16, 1e-5), 6.0, "LayerNorm forward 16384x1024 (HBM-bound)")
    print("## idle again:", smi())


if __name__ == "__main__":
    main()
