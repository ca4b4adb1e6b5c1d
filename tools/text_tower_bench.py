"""Developer micro-benchmark: the frozen fp32-grade CLIP text tower (ViT-B/32's: 12 x 512, 77 tokens) at 64 prompts, with the one-launch fp32
attention (FFVC_ATTN_TEXT=1) and with the GEMM + softmax + GEMM sequence (=0).  usage: FFVC_ATTN_TEXT=0|1 python tools/text_tower_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feed_forward_vqgan_clip_amd import clip as fclip, main as fmain  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

model = fclip.CLIP(fclip.random_state_dict(fclip.VIT_B32, seed=7), torch.float16)
tok = fmain.synthetic_tokens(64, seed=3).cuda()
with torch.no_grad():
    f = model.encode_text(tok)
    t = timeit(lambda: model.encode_text(tok), iters=10)
print(f"FFVC_ATTN_TEXT={os.environ.get('FFVC_ATTN_TEXT', '1')}: encode_text(64 x 77) {t * 1e3:6.2f} ms  checksum {f.double().abs().sum().item():.6f}")
