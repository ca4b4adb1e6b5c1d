"""`python main.py train configs/example.yaml` — same entry point as the reference (main.py:1464-1473)."""
import sys

from feed_forward_vqgan_clip_amd.main import _cli

if __name__ == "__main__":
    sys.exit(_cli(sys.argv[1:]) or 0)
