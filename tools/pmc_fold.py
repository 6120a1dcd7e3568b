"""Fold a rocprofv3 --pmc counter_collection.csv into per-kernel means (tools/ only).  Usage: python tools/pmc_fold.py <dir> [out.json]
Kernels are keyed by (short name, grid size) so the shapes of one sweep stay apart."""
import collections
import csv
import glob
import json
import re
import sys

src = sys.argv[1]
files = glob.glob(f"{src}/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in files:
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = re.sub(r"\(.*", "", row["Kernel_Name"])
            name = re.sub(r"^void ", "", name)[:90]
            key = f'{name} grid={row.get("Grid_Size", "?")} wg={row.get("Workgroup_Size", "?")}'
            a = acc[key][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
out = {}
for k, cs in acc.items():
    d = {c: v[0] / v[1] for c, v in cs.items()}
    d["dispatches"] = max(v[1] for v in cs.values())
    wc = d.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if c in d:
                d[c + "/WAVE_CYCLES"] = d[c] / wc
    if "SQ_BUSY_CYCLES" in d and "SQ_VALU_MFMA_BUSY_CYCLES" in d and d["SQ_BUSY_CYCLES"]:
        d["MFMA_BUSY/BUSY_CYCLES"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["SQ_BUSY_CYCLES"]
    out[k] = d
text = json.dumps(out, indent=1, sort_keys=True)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text)
print(text[:6000])
