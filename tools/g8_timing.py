"""Segment timing of the 8-phase GEMM: s_memtime stamps of wave 0 (group 0) and wave 4 (group 1) of workgroup 0 over K
tiles 4 and 5.  Needs a debug build of gemm2.hip with -DFFVC_G8_TIMING linked into a second library and loaded through
FFVC_LIB (make CXXEXTRA=-DFFVC_G8_TIMING builds the whole library that way)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import _lib  # noqa: E402
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402

M, N, Kd = 4096, 4096, 4096
x = torch.randn(M, Kd, device="cuda").bfloat16()
w = torch.randn(N, Kd, device="cuda").bfloat16()
y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
K.set_option("gemm2_tile", 512)
for _ in range(3):
    K.gemm(x, w, y, M, N, Kd, ldx=Kd, ldw=Kd)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 128)()
lib = _lib.load()
lib.ffvc_debug_g8_stamps.argtypes = [ctypes.c_void_p]
print("rc", lib.ffvc_debug_g8_stamps(buf))
names = {0: "L1 start", 1: "L1 reads issued", 3: "L1 barrier passed", 4: "M1 mfma+dma issued", 5: "M1 barrier passed",
         6: "L2 reads issued", 8: "L2 barrier passed", 9: "M2 mfma+dma issued", 10: "M2 barrier passed",
         11: "L3 reads issued", 13: "L3 barrier passed", 14: "M3 mfma+dma issued", 15: "M3 barrier passed"}
for g in range(2):
    st = [buf[g * 64 + i] for i in range(32)]
    base = st[0]
    print(f"group {g}: K tile 4 then 5, cycles since the tile-4 L1 start, (delta)")
    prev = base
    for i, v in enumerate(st):
        if (i % 16) not in names:
            continue
        print(f"  t{4 + i // 16} {names[i % 16]:20s} {v - base:7d}  (+{v - prev})")
        prev = v
