"""Phase timing of conv_row_kernel: s_memtime stamps of the four waves of workgroup 300 over K steps 3..8.  Needs a debug build of
gemm2.hip with -DFFVC_CR_TIMING linked into a second library and loaded through FFVC_LIB."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import _lib  # noqa: E402
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402

B, H, Cin, Cout = 16, 256, 128, 128
x = torch.randn(B, H, H, Cin, device="cuda").half()
w = torch.randn(Cout, 3, 3, Cin, device="cuda").half()
y = torch.empty(B, H, H, Cout, device="cuda", dtype=torch.float16)
for _ in range(3):
    K.gemm(x, w, y, B * H * H, Cout, 9 * Cin, ldw=9 * Cin, x_mode=K.OP_CONV3X3, conv=(H, H, Cin))
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (4 * 6 * 8))()
lib = _lib.load()
lib.ffvc_debug_cr_stamps.argtypes = [ctypes.c_void_p]
print("rc", lib.ffvc_debug_cr_stamps(buf))
row2 = os.environ.get("FFVC_CONV_ROW2", "1") != "0"
ev = (["step start", "half 1 issued", "vmcnt passed", "barrier passed", "half 2 issued", "-", "-"] if row2 else
      ["step start", "DMA issued", "vmcnt(0) passed", "barrier 1 passed", "sub 0 done", "sub 1 done", "barrier 2 passed"])
for wv in range(4):
    if row2:
        k0, k1, k2 = (buf[(wv * 6) * 8 + e] for e in (5, 6, 7))
        print(f"wave {wv}: kernel start -> main loop end {k1 - k0} cycles, epilogue {k2 - k1} cycles")
    base = buf[(wv * 6 + 0) * 8 + 0]
    print(f"wave {wv}: cycles since its step-3 start; per event (delta to previous event)")
    prev = base
    for st in range(6):
        row = []
        for e in range(5 if row2 else 7):
            v = buf[(wv * 6 + st) * 8 + e]
            row.append(f"{ev[e]} {v - base} (+{v - prev})")
            prev = v
        print(f"  step {3 + st} kw={(3 + st) % 3}: " + " | ".join(row))
