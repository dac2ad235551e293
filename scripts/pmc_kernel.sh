#!/bin/bash
# SQ counters (two rocprofv3 --pmc passes, program directly after --) for the kernels of one python script.
# usage: scripts/pmc_kernel.sh <tag> <kernel-name-regex> <script.py> [args...]      (run from the repo root via gpurun)
TAG=$1; PAT=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/p1 -- python3 $ROOT/"$@" > $OUT/log1.txt 2>&1 || echo "pass 1 failed"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_LDS SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/p2 -- python3 $ROOT/"$@" > $OUT/log2.txt 2>&1 || echo "pass 2 failed"
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_SMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA --kernel-trace --output-format csv -d $OUT/p3 -- python3 $ROOT/"$@" > $OUT/log3.txt 2>&1 || echo "pass 3 failed"
python3 - "$OUT" "$PAT" <<'PY'
import glob, sys, re
import pandas as pd
out, pat = sys.argv[1], sys.argv[2]
rows = []
for p in ("p1", "p2", "p3"):
    fs = glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(p, "no counter file"); continue
    c = pd.read_csv(fs[0])
    c = c[c.Kernel_Name.str.contains(pat, regex=True)]
    c["K"] = c.Kernel_Name.map(lambda n: n.split("(")[0].replace("void ", "")[:60])
    piv = c.pivot_table(index=["K", "Dispatch_Id", "Start_Timestamp", "End_Timestamp"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()
    piv["dur_us"] = (piv.End_Timestamp - piv.Start_Timestamp) / 1e3
    num = [x for x in piv.columns if x not in ("K", "Dispatch_Id", "Start_Timestamp", "End_Timestamp")]
    g = piv.groupby("K")[num]
    res = g.mean()
    res.insert(0, "dispatches", g.size())
    rows.append(res)
    with pd.option_context("display.width", 250, "display.max_columns", 40, "display.float_format", "{:.5g}".format):
        print(f"== {p}: mean per dispatch ==")
        print(res.T.to_string())
if rows:
    pd.concat(rows, axis=1).to_csv(f"{out}/summary.csv")
PY
