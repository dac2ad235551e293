#!/bin/bash
# LDS / VALU counters of the table-path assembly kernel (scripts/diag_assembly.py)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_assembly
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/scripts/diag_assembly.py "$@" > $OUT/log.txt 2>&1
python3 - <<PY
import glob, pandas as pd
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)[0]
c = pd.read_csv(f)
c = c[c.Kernel_Name.str.contains("k_assemble")]
piv = c.pivot_table(index=["Kernel_Name","Dispatch_Id","Start_Timestamp","End_Timestamp"], columns="Counter_Name", values="Counter_Value", aggfunc="sum").reset_index()
piv["dur_us"] = (piv.End_Timestamp - piv.Start_Timestamp) / 1e3
piv["cu_cycles"] = piv.GRBM_GUI_ACTIVE / 8
piv["lds_busy"] = piv.SQ_LDS_IDX_ACTIVE / (piv.cu_cycles * 256)
piv["lds_conf_frac"] = piv.SQ_LDS_BANK_CONFLICT / piv.SQ_LDS_IDX_ACTIVE
piv["K"] = piv.Kernel_Name.str.extract(r"(k_assemble[a-z_]*<[a-z0-9, ]+>)")
cols = [c for c in ["dur_us","cu_cycles","lds_busy","lds_conf_frac","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_INSTS_VALU","SQ_WAVE_CYCLES","SQ_BUSY_CYCLES"] if c in piv]
print(piv.groupby("K")[cols].median().to_string())
PY
