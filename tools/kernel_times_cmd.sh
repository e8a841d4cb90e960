#!/bin/bash
# Per-kernel totals of an arbitrary command (rocprofv3 kernel trace): tools/kernel_times_cmd.sh python tools/bench_linked.py
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=/tmp/ktc_$$
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- "$@" > $OUT.log 2>&1
python - $OUT <<'PY'
import csv, glob, sys, os
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in [x for x in rows if "atr::" in x["Name"] or "atr_piece" in x["Name"]][:40]:
        print("%-60s calls=%-5s avg_us=%9.1f total_ms=%8.2f %5.1f%%" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
tail -n 2 $OUT.log | cut -c1-300
