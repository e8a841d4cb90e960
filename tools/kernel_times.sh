#!/bin/bash
# Per-kernel average durations of one bench.py run (rocprofv3 kernel trace), atr:: kernels only.
# usage: tools/kernel_times.sh [extra bench args]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=/tmp/kt_$$
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python bench.py --no-cpu-baseline --steps 10 --warmup 2 "$@" > $OUT.log 2>&1
python - $OUT <<'PY'
import csv, glob, sys, os
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "atr::" in r["Name"] and "pack" not in r["Name"]:
            print("%-40s calls=%s avg_us=%.1f" % (r["Name"].split("(")[0][-40:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
