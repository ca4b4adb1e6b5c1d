"""Build profiles/<round>_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over one bench step.
usage: python tools/pmc_traffic.py <fetch_csv_dir> <write_csv_dir> > profiles/r01_pmc_traffic.json
Counter unit = KiB; FETCH_SIZE is doubled on gfx950 (128-byte requests tallied at 64 B, MI355X_MICROARCH.md)."""
import collections
import csv
import glob
import json
import re
import sys


def per_kernel(d, counter):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            agg[row["Kernel_Name"]] += float(row["Counter_Value"])
            cnt[row["Kernel_Name"]] += 1
    return agg, cnt


CLASSES = {   # bench.py's GEMM classes (dtype suffix stripped) -> kernel-name patterns, demangled or mangled (rocprofv3 does
    # not demangle the _Float16 instantiations: gemm2_kernelIDF16_Li0ELi0E...)
    "gemm_nt": r"gemm[28]p?_kernel<[^,]*, ?0, 0,|gemm2p?_kernelI(DF16_|t)Li0ELi0E|gemm8_kernelI(DF16_|t)Li0E",
    "conv3x3": r"conv_row\d?_kernel|gemm2p?_kernel<[^,]*, ?2, 0,|gemm2p?_kernelI(DF16_|t)Li2ELi0E",
    "gemm_nn": r"gemm2p?_kernel<[^,]*, ?0, 1,|gemm2p?_kernelI(DF16_|t)Li0ELi1E",
    "gemm_tn": r"gemm_kernel<[^,]*, 1, 1|gemm_kernelI(DF16_|t)Li1ELi1E|gemm2p?_kernel<[^,]*, ?1, 1,|gemm2p?_kernelI(DF16_|t)Li1ELi1E",
    "tokmix": r"tokmix_(fwd|bwd_hidden)_kernel",
}
fa, fc = per_kernel(sys.argv[1], "FETCH_SIZE")
wa, wc = per_kernel(sys.argv[2], "WRITE_SIZE")
import hashlib
import os
import subprocess

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = os.path.join(_root, "feed_forward_vqgan_clip_amd", "lib", "libffvc_hip.so")
_sha = hashlib.sha256(open(_lib, "rb").read()).hexdigest() if os.path.exists(_lib) else None
try:
    _head = subprocess.run(["git", "-C", _root, "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip() or None
except OSError:
    _head = None
out = {"_lib_sha256": _sha, "_git_head": _head,
       "_provenance": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) -- python3 bench.py "
                      "--steps 1 --warmup 1 --no-cpu-baseline --no-roofline; per-kernel-class dispatch-weighted averages; "
                      "counter unit KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950), WRITE_SIZE uncorrected."}
for cls, pat in CLASSES.items():
    ks = [k for k in fa if re.search(pat, k)]
    n = sum(fc[k] for k in ks)
    if not n:
        continue
    fetch = sum(fa[k] for k in ks) / n
    write = sum(wa.get(k, 0.0) for k in ks) / max(1, sum(wc.get(k, 0) for k in ks))
    out[cls] = {"kernels": sorted({re.sub(r"^void |\(anonymous namespace\)::|\(ffvc_gemm_desc.*", "", k)[:60] for k in ks}), "dispatches_profiled": n,
                "fetch_kib": round(fetch, 1), "write_kib": round(write, 1),
                "hbm_bytes_per_launch": int((2 * fetch + write) * 1024)}
print(json.dumps(out, indent=1))
