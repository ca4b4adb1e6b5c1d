"""Phase timing of tokmix_bwd_hidden_kernel: s_memtime stamps of the eight waves of workgroup (1, 7) over chunks 4..9.  Needs a debug
build of tokmix.hip with -DFFVC_TM_TIMING linked into a second library and loaded through FFVC_LIB."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import _lib  # noqa: E402
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402

B, T, D, O = 64, 256, 1024, 1024
dt = torch.float16
xn = torch.randn(B, T, D, device="cuda").to(dt)
dy = torch.randn(B, T, D, device="cuda").to(dt)
w1 = (torch.randn(O, T, device="cuda") * 0.05).to(dt)
b1 = torch.randn(O, device="cuda") * 0.1
w2t = (torch.randn(O, T, device="cuda") * 0.05).to(dt)
for _ in range(3):
    h, dh = K.tokmix_bwd_hidden(xn, dy, w1, b1, w2t)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    K.tokmix_bwd_hidden(xn, dy, w1, b1, w2t)
e1.record()
torch.cuda.synchronize()
print("tokmix_bwd_hidden %.1f us" % (e0.elapsed_time(e1) * 100))
buf = (ctypes.c_ulonglong * (8 * 6 * 8))()
lib = _lib.load()
lib.ffvc_debug_tm_stamps.argtypes = [ctypes.c_void_p]
print("rc", lib.ffvc_debug_tm_stamps(buf))
ev = ["top", "vmcnt passed", "barrier passed", "DMA issued", "flush done", "MFMAs issued", "GELU + staging done"]
for wv in (0, 3, 4, 7):
    base = buf[(wv * 6) * 8]
    prev = base
    print(f"wave {wv}")
    for ch in range(6):
        row = []
        for e in range(7):
            v = buf[(wv * 6 + ch) * 8 + e]
            row.append(f"{ev[e]} +{v - prev}")
            prev = v
        print(f"  chunk {4 + ch}: " + " | ".join(row))
