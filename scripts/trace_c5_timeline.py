"""Timeline of the sharded path from a rocprofv3 --kernel-trace --memory-copy-trace run of `bench.py --config c5` (GPU box): the last exchanges'
kernels and copies in start order, times relative to the first, to see what does not overlap.  usage: trace_c5_timeline.py <dir>"""
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"][:48], r.get("Stream_Id", r.get("Queue_Id", ""))))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", r.get("Name", ""))[:24] + " " + str(r.get("Bytes", r.get("Size", ""))), r.get("Stream_Id", "")))
ev.sort()
# the last search kernels
idx = [i for i, e in enumerate(ev) if e[2] == "K" and "search_kernel" in e[3]]
if len(idx) < 6:
    print("few search kernels", len(idx)); sys.exit(0)
lo = idx[-6]; hi = idx[-3]
t0 = ev[lo][0]
for e in ev[lo - 12: hi + 14]:
    dur = (e[1] - e[0]) / 1e6
    if dur < 0.02 and e[2] == "K" and "search" not in e[3] and "lut" not in e[3] and "finalize" not in e[3]:
        tag = "."
    else:
        tag = ""
    print("%9.3f -> %9.3f  %7.3f ms  %s %-50s %s %s" % ((e[0] - t0) / 1e6, (e[1] - t0) / 1e6, dur, e[2], e[3], e[4], tag))
