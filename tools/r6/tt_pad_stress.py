"""Round 6: stress of the padded reduction-major operands (tests/test_gemm_gpu.py::test_padded_reduction_major_operands) — poisoned
allocator cache, many repetitions, location of the worst element on a mismatch.  usage (GPU box): FFVC_TT_PAD=1 python tools/r6/tt_pad_stress.py   (without FFVC_TT_PAD=1 the launches take the register-staged kernel)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from feed_forward_vqgan_clip_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
bad = 0
for rep in range(int(os.environ.get('STRESS_REPS', '40'))):
    for dt in (torch.bfloat16, torch.float16):
        for (M, N, ldx, ldw) in [(3060, 1024, 3064, 1024), (1024, 1020, 1024, 1024), (3060, 1020, 3064, 1024), (3060, 1024, 3060, 1024)]:
            junk = torch.full((8 << 20,), float("nan"), device=dev)      # poison what the next allocations may reuse
            del junk
            rows = 512
            g = torch.Generator(device="cpu").manual_seed(rep * 7 + 1)
            dy = torch.randn(rows, ldx, generator=g).to(dt).to(dev)
            x = torch.randn(rows, ldw, generator=g).to(dt).to(dev)
            if ldx > M:
                dy[:, M:] = float("nan")
            if ldw > N:
                x[:, N:] = float("nan")
            wg = torch.ones(M, N, dtype=torch.float32, device=dev)
            K.gemm(dy, x, wg, M, N, rows, ldx=ldx, ldw=ldw, x_mode=K.OP_TRANS, w_mode=K.OP_TRANS, flags=K.F_ACCUM_OUT)
            ref = dy[:, :M].double().T @ x[:, :N].double() + 1.0
            err = (wg.double() - ref).abs()
            rel = (err.max() / ref.abs().max()).item()
            if not (rel < 2e-5):
                bad += 1
                nbad = int((err > 1e-3 * ref.abs().max()).sum())
                i = int(err.argmax())
                print(f"rep {rep} {dt} M={M} N={N} ldx={ldx} ldw={ldw}: rel {rel:.3e}, {nbad} bad elements, worst at row {i // N} col {i % N}; "
                      f"finite {bool(torch.isfinite(wg).all())}", flush=True)
                rows_bad = torch.nonzero((err > 1e-3 * ref.abs().max()).any(dim=1)).flatten()
                cols_bad = torch.nonzero((err > 1e-3 * ref.abs().max()).any(dim=0)).flatten()
                print("   bad rows", rows_bad[:12].tolist(), "..", rows_bad[-4:].tolist(), "n", rows_bad.numel(),
                      "| bad cols", cols_bad[:12].tolist(), "..", cols_bad[-4:].tolist(), "n", cols_bad.numel(), flush=True)
print("mismatches:", bad)
