"""Print a one-line provenance header for files under profiles/: sha256 of the in-tree library, git HEAD, date.
    python tools/stamp.py >> profiles/r03_x.txt        (bench.py only trusts PMC numbers whose stamp matches the library)"""
import datetime
import hashlib
import os
import subprocess

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.environ.get("FFVC_LIB") or os.path.join(root, "feed_forward_vqgan_clip_amd", "lib", "libffvc_hip.so")
sha = hashlib.sha256(open(lib, "rb").read()).hexdigest() if os.path.exists(lib) else "missing"
try:
    head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "n/a"
except OSError:
    head = "n/a"
print(f"# _lib_sha256 {sha}  git {head}  {datetime.datetime.utcnow().strftime('%Y-%m-%dT%H:%MZ')}")
