"""Per-kernel totals of a rocprofv3 --kernel-trace run plus the wall-clock union of all kernel
intervals (sum of durations / union > 1 means kernels of different streams ran concurrently)."""
import csv, glob, sys, collections
d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:34], r["Stream_Id"]) for r in rows)
# keep the last 45 % of the trace (the timed step, not workload creation / warm-up)
t_lo = iv[0][0] + (iv[-1][1] - iv[0][0]) * (float(sys.argv[2]) if len(sys.argv) > 2 else 0.0)
iv = [x for x in iv if x[0] >= t_lo]
tot = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, q in iv:
  tot[(n, q)][0] += 1; tot[(n, q)][1] += (e - s) / 1e6
union = 0.0; cur_s, cur_e = iv[0][0], iv[0][1]
for s, e, n, q in iv[1:]:
  if s > cur_e:
    union += cur_e - cur_s; cur_s, cur_e = s, e
  else:
    cur_e = max(cur_e, e)
union += cur_e - cur_s
ssum = sum(v[1] for v in tot.values())
print("kernels %d  sum of durations %.1f ms  union %.1f ms  span %.1f ms" % (len(iv), ssum, union / 1e6, (iv[-1][1] - iv[0][0]) / 1e6))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:16]:
  print("%-36s stream %-3s calls %5d  total %8.1f ms  avg %8.1f us" % (k[0], k[1], v[0], v[1], v[1] / v[0] * 1e3))
