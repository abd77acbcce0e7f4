"""Prints the kernel timeline (start offset, duration) of the last N kernels of a rocprofv3 kernel trace."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = rows[-n:]
t0 = int(rows[0]['Start_Timestamp'])
prev_end = t0
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0][-48:]
    print(f"{(s - t0) / 1e6:9.3f} ms  dur {(e - s) / 1e6:7.3f} ms  gap {(s - prev_end) / 1e6:7.3f}  q{r.get('Queue_Id', '?')}  {name}")
    prev_end = max(prev_end, e)
