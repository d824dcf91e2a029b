#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output).
Units and the gfx950 correction follow /opt/skills/guides (cdna_hip_programming.md section 7,
MI355X_MICROARCH.md section HBM): both counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of a
wide coalesced read stream, so the read side is doubled (an upper estimate for narrow accesses).
Output: JSON {phase: {"kernel", "fetch_bytes", "write_bytes", "hbm_bytes", "launches"}} per launch averages."""
import csv
import glob
import json
import os
import sys

PHASE = {"k_schur_syrk": "schur", "k_normal_eq<128, 0, true>": "normal_eq", "k_normal_eq": "normal_eq",
         "k_backsubst_cost_step": "backsubst", "k_backsubst": "backsubst", "k_cost": "cost", "k_eval": "eval", "k_reduce_all": "assemble",
         "k_chol_step": "solve"}


def load(d, counter):
    tot, cnt = {}, {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            k = row["Kernel_Name"]
            tot[k] = tot.get(k, 0.0) + float(row["Counter_Value"])
            cnt[k] = cnt.get(k, 0) + 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    name = k.split("(")[0].replace("void ", "").replace("cvxk::", "").replace("cvxs::", "")
    if "k_normal_eq" in name and "0, " not in name and ", 0" not in name:
        continue
    ph = next((v for p, v in PHASE.items() if name.startswith(p)), None)
    if ph is None or ph in out and "normal_eq" not in name:
        continue
    if "k_normal_eq" in name and "128, 1" in name:
        continue
    f = fetch.get(k, (0.0, 0))
    w = write.get(k, (0.0, 0))
    out[ph] = {"kernel": name, "fetch_bytes": 2.0 * f[0] * 1024.0, "write_bytes": w[0] * 1024.0,
               "hbm_bytes": 2.0 * f[0] * 1024.0 + w[0] * 1024.0, "launches": max(f[1], w[1]),
               "note": "FETCH_SIZE doubled (gfx950 wide-read correction); KiB -> bytes"}
print(json.dumps(out, indent=1))
