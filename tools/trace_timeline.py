"""Timeline of the LAST frame in a rocprofv3 kernel_trace.csv: per kernel start offset (us), duration (us), queue.  usage: trace_timeline.py csv n_kernels_per_frame"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2])
fr = rows[-n:]
t0 = int(fr[0]["Start_Timestamp"])
busy_end = 0; gaps = 0.0
for r in fr:
    st, en = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].split("(")[0].replace("void rtf::", "").replace("front_", "")[:28]
    print(f"{st/1e3:9.1f} {(en-st)/1e3:8.1f}  q{r.get('Queue_Id','?'):>3s}  {name}")
print("frame span us", (max(int(r["End_Timestamp"]) for r in fr) - t0) / 1e3)
