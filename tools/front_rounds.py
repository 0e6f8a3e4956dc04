"""Per-round kernel durations of the ray-front pipeline from a rocprofv3 kernel trace directory."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = 0; tot = {}
for r in rows:
    nm = r['Kernel_Name']
    if 'front' not in nm: continue
    short = 'shade0' if 'true>(' in nm and 'shade' in nm else 'shade' if 'shade' in nm else 'trace' if 'trace' in nm else 'resolve'
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot[short] = tot.get(short, 0) + d
    if n < 26: print(f"{short:8s} {d:9.1f} us")
    n += 1
print({k: round(v / frames / 1e3, 2) for k, v in tot.items()}, "ms per frame")
