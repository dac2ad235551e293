"""Does the side-stream panel work overlap the big trailing-update kernels?  (rocprofv3 kernel trace)"""
import glob, sys
import pandas as pd
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
t = pd.read_csv(f)
t["name"] = t.Kernel_Name.str.split("(").str[0].str.replace("void ", "")
t = t.sort_values("Start_Timestamp")
big = t[t.name.str.contains("k_syrk_panels")]
small = t[t.name.str.contains("potrf64|trsm64")]
# take the last factorisation: last 78*2 syrk launches
b = big.tail(156)
lo, hi = b.Start_Timestamp.min(), b.End_Timestamp.max()
s = small[(small.Start_Timestamp >= lo) & (small.End_Timestamp <= hi)]
import numpy as np
bs, be = b.Start_Timestamp.values, b.End_Timestamp.values
inside = 0
for st, en in zip(s.Start_Timestamp.values, s.End_Timestamp.values):
    k = np.searchsorted(bs, st, side="right") - 1
    if k >= 0 and st < be[k]:
        inside += 1
print(f"window {(hi-lo)/1e6:.1f} ms; syrk busy {((be-bs).sum())/1e6:.1f} ms; panel kernels {len(s)}, started while a syrk kernel was running: {inside}")
print("queues:", t.groupby("name").Queue_Id.unique().to_dict() if "Queue_Id" in t else "n/a")
print(b[["name","Start_Timestamp","End_Timestamp"]].assign(dur_us=(b.End_Timestamp-b.Start_Timestamp)/1e3).head(6).to_string())
print(s[["name","Start_Timestamp","End_Timestamp"]].assign(dur_us=(s.End_Timestamp-s.Start_Timestamp)/1e3).head(12).to_string())
