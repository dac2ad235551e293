#!/bin/bash
# stall / issue counters of the GEMM micro-benchmark, two passes; args: variants
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_gemm2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/p1 -- python3 $ROOT/scripts/bench_gemm.py "$@" > $OUT/log1.txt 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $OUT/p2 -- python3 $ROOT/scripts/bench_gemm.py "$@" > $OUT/log2.txt 2>&1
python3 - <<PY
import glob, pandas as pd
for p in ("p1", "p2"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True)[0]
    c = pd.read_csv(f)
    c = c[c.Kernel_Name.str.contains("k_gemm_nt")]
    piv = c.pivot_table(index=["Kernel_Name","Dispatch_Id","Start_Timestamp","End_Timestamp"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()
    piv["dur_us"] = (piv.End_Timestamp - piv.Start_Timestamp) / 1e3
    piv["cu_cyc"] = piv.GRBM_GUI_ACTIVE / 8
    piv["K"] = piv.Kernel_Name.str.extract(r"(k_gemm_nt[_a-z]*)")
    num = [x for x in piv.columns if x not in ("Kernel_Name", "K", "Dispatch_Id", "Start_Timestamp", "End_Timestamp")]
    pd.set_option("display.width", 250)
    print(piv.groupby("K")[num].median().T.to_string())
PY
