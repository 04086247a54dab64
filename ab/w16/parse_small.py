import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
seq = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
t16 = [d for n, d in seq if "fused_train16" in n]
red = [d for n, d in seq if "reduce16" in n]
for i in range(0, len(t16), 30):
    print("config", i // 30, "fused_train16 median us %.1f" % np.median(t16[i+10:i+30]), "reduce16 %.1f" % np.median(red[i+10:i+30]))
