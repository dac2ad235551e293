"""Condense the rocprofv3 outputs of scripts/profile_bench.sh into one per-kernel table."""
import glob
import os
import sys

import pandas as pd

out = sys.argv[1]


def find(sub, pat):
    fs = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return fs[0] if fs else None


def short(n):
    return n.split("(")[0].replace("void ", "")


st = find("trace", "*kernel_stats.csv")
if st:
    s = pd.read_csv(st)
    s["Name"] = s["Name"].map(short)
    print("== kernel stats (all passes of the run: warm-up + timed) ==")
    print(s[["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"]].to_string(index=False))
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    f = find(sub, "*counter_collection.csv")
    if not f:
        print(f"== {sub}: no counter file ==")
        continue
    c = pd.read_csv(f)
    c["Kernel_Name"] = c["Kernel_Name"].map(short)
    piv = c.pivot_table(index=["Kernel_Name", "Dispatch_Id"], columns="Counter_Name", values="Counter_Value", aggfunc="sum")
    g = piv.groupby("Kernel_Name")
    res = g.mean()
    res.insert(0, "dispatches", g.size())
    print(f"== {sub}: mean per dispatch ==")
    with pd.option_context("display.width", 250, "display.max_columns", 30, "display.float_format", "{:.4g}".format):
        print(res.to_string())
    res.to_csv(os.path.join(out, f"{sub}_per_kernel_mean.csv"))
