#!/bin/bash
# Dev (round 5): duration of every td_symv / td_row / td_w launch of one cfg3 eigh call (one stream group), averaged per
# tile count T of the trailing matrix, next to the bytes the mat-vec has to read.
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r5_symv_cols
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export PS_DEV_ENV=1 PS_EIGH_TD_STREAMS=${1:-1}
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tools/dev_eigh_one.py > $OUT/run.log 2> $OUT/run.err
cd $GRAFT_REPO_ROOT
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for name in ("td_symv_kernel", "td_row_kernel", "td_w_kernel"):
  d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if name in r["Kernel_Name"]]
  per = len(d) // 2            # two calls in the trace: the second one
  d = d[per:]
  print(name, "launches per call", len(d), "sum %.1f ms" % (sum(d) / 1e3))
  if name != "td_symv_kernel": continue
  n, nt = 2048, 16
  by = collections.defaultdict(list)
  for j, us in enumerate(d):
    by[nt - (j + 1) // 128].append((j, us))
  for T in sorted(by, reverse=True):
    js = by[T]
    avg = sum(u for _, u in js) / len(js)
    # bytes: upper tiles of T x T minus the masked rows of the first tile row (on average half a tile row)
    tiles = T * (T + 1) / 2
    bytes_ = 64 * (tiles - 0.5 * T * 0.5) * 128 * 128 * 4   # rough: half of the first tile row masked on average
    print(f"  T={T:2d}: {len(js):4d} launches, avg {avg:7.1f} us, ~{bytes_ / 1e6:7.1f} MB -> {bytes_ / avg / 1e6:5.2f} TB/s")
PY
rm -rf $OUT/trace
