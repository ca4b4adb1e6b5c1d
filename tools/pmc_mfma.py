"""MFMA utilisation of one bench step from a rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES).

    python tools/pmc_mfma.py <counter dir> <steps profiled> <ms per step>

Per kernel: matrix-pipe busy cycles summed over the chip's 1024 SIMDs, as a fraction of the SIMD cycles the kernel was
resident (SQ_BUSY_CYCLES counts per XCD-quadrant: 32 SIMD-cycles per count, calibrated on the 8192^3 GEMM in
profiles/r01_gemm_micro.txt).  Step level: all busy cycles / (1024 SIMDs x step time x 2.4 GHz) — the MFMA utilisation
against the chip's peak clock that BASELINE.json's north_star asks for."""
import collections
import csv
import glob
import re
import sys

d, steps, ms = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for f in files:
    seen = set()
    for row in csv.DictReader(open(f)):
        name = re.sub(r"^void ", "", row["Kernel_Name"])
        name = re.sub(r"\(ffvc_gemm_desc.*|\(float const.*|\(unsigned short.*", "", name)[:90]
        a = agg[name]
        key = (row.get("Dispatch_Id"), name)
        if key not in seen:
            seen.add(key)
            a[0] += 1
        if row["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            a[1] += float(row["Counter_Value"])
        elif row["Counter_Name"] == "SQ_BUSY_CYCLES":
            a[2] += float(row["Counter_Value"])
tot = sum(v[1] for v in agg.values())
print(f"# MFMA busy cycles per step: {tot / steps / 1e6:.1f} M SIMD-cycles = "
      f"{tot / steps / (1024 * ms * 1e-3 * 2.4e9) * 100:.1f} % of 1024 SIMDs x {ms:.1f} ms x 2.4 GHz")
print("# dispatches/step   MFMA busy M-cycles/step   busy % while resident   kernel")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if v[1] <= 0:
        continue
    print(f"{v[0] / steps:10.0f} {v[1] / steps / 1e6:16.1f} {100 * v[1] / max(1.0, 32 * v[2]):16.1f}   {k}")
