#!/bin/bash
# effective clock and MFMA utilisation of the GEMM micro-benchmark
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_gemm
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/scripts/bench_gemm.py "$@" > $OUT/log.txt 2>&1
python3 - <<PY
import glob, pandas as pd
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)[0]
c = pd.read_csv(f)
c = c[c.Kernel_Name.str.contains("k_gemm_nt")]
piv = c.pivot_table(index=["Kernel_Name","Dispatch_Id","Start_Timestamp","End_Timestamp"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()
piv["dur_ns"] = piv.End_Timestamp - piv.Start_Timestamp
piv["clock_GHz"] = piv.GRBM_GUI_ACTIVE / 8 / piv.dur_ns
piv["mfma_util"] = piv.SQ_VALU_MFMA_BUSY_CYCLES / (piv.GRBM_GUI_ACTIVE / 8 * 1024)
piv["lds_conf_frac"] = piv.SQ_LDS_BANK_CONFLICT / piv.SQ_LDS_IDX_ACTIVE
piv["K"] = piv.Kernel_Name.str.extract(r"(k_gemm_nt<[0-9, ]+>)")
print(piv.groupby("K")[["dur_ns","clock_GHz","mfma_util","lds_conf_frac","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_WAVE_CYCLES"]].median().to_string())
PY
