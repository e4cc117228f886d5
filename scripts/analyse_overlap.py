"""Reads a rocprofv3 kernel trace (csv) and reports, for the search kernels, how much of each dispatch's duration overlapped
another search-kernel dispatch, plus the distinct queues they ran on. usage: analyse_overlap.py <kernel_trace.csv> <out.json> [label]"""
import csv
import json
import sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("void search_kernel<") or r["Kernel_Name"].startswith("search_kernel<")]
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows))
ev = ev[len(ev) // 4:]                      # the steady state (warm-up dropped)
tot = sum(e - s for s, e, _, _ in ev)
ov = 0
for i, (s, e, _, _) in enumerate(ev):
    for s2, e2, _, _ in ev[i + 1:i + 8]:
        if s2 >= e:
            break
        ov += min(e, e2) - s2
span = ev[-1][1] - ev[0][0]
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in ev:                       # union of the intervals
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
out = {"label": sys.argv[3] if len(sys.argv) > 3 else "", "search_kernel_dispatches": len(ev),
       "mean_kernel_us": tot / len(ev) / 1e3, "sum_of_durations_ms": tot / 1e6, "pairwise_overlap_ms": ov / 1e6,
       "overlap_fraction_of_kernel_time": ov / tot, "span_ms": span / 1e6, "union_busy_ms": busy / 1e6,
       "search_kernel_busy_fraction_of_span": busy / span,
       "queues": sorted({q for _, _, q, _ in ev}), "streams": sorted({s for _, _, _, s in ev})}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out))
