#!/usr/bin/env python3
"""Copies the summaries of gpurun_out/<round> (tools/profile_all.sh; round tag from $ROUND, default r03) into profiles/: kernel_stats csv per trace,
one text file of mean counter values per dispatch for the kernels of interest per counter run, and
profiles/traffic.json (memory-side bytes per launch of the fused step kernels)."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("ROUND", "r03")
SRC = os.path.join(ROOT, "gpurun_out", ROUND)
DST = os.path.join(ROOT, "profiles")

for tag in ("c4", "c2", "c5", "ops", "k50", "ctx", "train"):
    found = glob.glob(os.path.join(SRC, f"trace_{tag}", "**", "*kernel_stats.csv"), recursive=True)
    if found:       # (a directory that saw several runs: the newest)
        shutil.copy(max(found, key=os.path.getmtime), os.path.join(DST, f"{ROUND}_{tag}_kernel_stats.csv"))
    log = os.path.join(SRC, f"trace_{tag}.log")
    if os.path.exists(log):
        lines = [l for l in open(log) if l.startswith("{") or " ms" in l and not l.startswith(("W2", "E2", "I2"))]
        with open(os.path.join(DST, f"{ROUND}_{tag}_under_rocprof.txt"), "w") as out:
            out.writelines(l for l in lines if "amdgpu.ids" not in l)


def counters(tag, patterns):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for path in glob.glob(os.path.join(SRC, f"pmc_{tag}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            for p in patterns:
                if p in r["Kernel_Name"]:
                    a = acc[p][r["Counter_Name"]]
                    a[0] += float(r["Counter_Value"])
                    a[1] += 1
    return acc


KERNELS = {"c4": ["ring_kernel", "weights_kernel"], "c2": ["diag_kernel"], "c5": ["diag_kernel", "label_step_kernel"],
           "ops": ["diag_kernel", "diag_backward_finish"],
           "ctx": ["label_context_kernel", "context_wgrad_kernel", "context_pre_kernel"]}
traffic = {}
for tag, pats in KERNELS.items():
    acc = counters(tag, pats)
    with open(os.path.join(DST, f"{ROUND}_pmc_{tag}.txt"), "w") as out:
        out.write(f"rocprofv3 --pmc passes of tools/profile_all.sh ({tag}); mean counter value per dispatch\n")
        for p in pats:
            out.write(f"\n== kernels matching '{p}'\n")
            for name in sorted(acc[p]):
                s, n = acc[p][name]
                out.write(f"{name:36s} {s / n:18.1f}   dispatches {n}\n")
    step = {"c4": "ring_kernel", "c2": "diag_kernel"}.get(tag)
    if step and "FETCH_SIZE" in acc[step]:
        fetch = acc[step]["FETCH_SIZE"][0] / acc[step]["FETCH_SIZE"][1]
        write = acc[step]["WRITE_SIZE"][0] / acc[step]["WRITE_SIZE"][1]
        traffic[tag] = {"kernel": step, "fetch_size_kb_per_launch": fetch, "write_size_kb_per_launch": write,
                        "fetch_correction": 2.0, "traffic_bytes_per_launch": int((2 * fetch + write) * 1024),
                        "pixels_per_launch": {"c4": 16 * 1024 * 2048, "c2": 8 * 512 * 512}[tag],
                        "sources": [f"profiles/{ROUND}_pmc_{tag}.txt", f"profiles/{ROUND}_pmc_fetch_calibration.txt"]}
for path in glob.glob(os.path.join(SRC, "pmc_calib", "**", "*counter_collection.csv"), recursive=True):
    with open(os.path.join(DST, f"{ROUND}_pmc_fetch_calibration.txt"), "w") as out:
        out.write("tools/exp/exp_calib_fetch (1 GiB streamed once, one dword per lane and load) under rocprofv3 --pmc FETCH_SIZE\n")
        for r in csv.DictReader(open(path)):
            out.write(f"{r['Kernel_Name'][:60]}  {r['Counter_Name']} = {r['Counter_Value']} KB\n")
traffic["note"] = ("rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes, mean over the kernel's dispatches of one bench run "
                   "(the 1-iteration and 10-iteration calls launch the same kernel on the same data); FETCH_SIZE doubled: on gfx950 it "
                   "reports half the bytes of coalesced dword-per-lane reads (calibration file).  The counters sit on the L2's memory "
                   "side, so reads served by the memory-side cache are included: an upper bound on HBM bytes.")
json.dump(traffic, open(os.path.join(DST, "traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1)[:1500])
