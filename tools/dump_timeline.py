"""dev: ordered kernel timeline of the tail of a rocprofv3 --kernel-trace run: start offset, duration,
idle gap before the kernel, grid size.  usage: dump_timeline.py <trace dir> <fraction to skip> [max rows]"""
import csv, glob, sys
d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("psk::", "").replace("at::native::", "").replace(" ", "")[:50],
             int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0), int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 1)) or 1)) for r in rows)
t0 = iv[0][0] + (iv[-1][1] - iv[0][0]) * float(sys.argv[2])
iv = [x for x in iv if x[0] >= t0]
lim = int(sys.argv[3]) if len(sys.argv) > 3 else 400
prev = iv[0][0]
busy = gap = 0.0
for s, e, n, g, w in iv[:lim]:
  print("%9.1f us  dur %8.1f  gap %7.1f  wgs %6d  %s" % ((s - iv[0][0]) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, g // max(w, 1), n))
  busy += (e - s) / 1e3; gap += max(0, s - prev) / 1e3
  prev = max(prev, e)
print("rows %d  busy %.1f us  idle %.1f us" % (min(lim, len(iv)), busy, gap))
