#!/bin/bash
# Start / end of every library kernel of the LAST step of a command, relative to the step's first kernel
# (rocprofv3 kernel trace): which launches overlap.   tools/kernel_timeline_cmd.sh <first kernel name part> <command ...>
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
FIRST=$1; shift
OUT=/tmp/ktl_$$
rocprofv3 --kernel-trace --output-format csv -d $OUT -- "$@" > $OUT.log 2>&1
python - $OUT "$FIRST" <<'PY'
import csv, glob, sys, os
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "atr::" in r["Kernel_Name"] or "atr_piece" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"]]
lo = starts[-1]
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:]:
    print("%-58s %9.1f .. %9.1f us  (%7.1f)  stream %s" % (r["Kernel_Name"].split("(")[0][-58:], (int(r["Start_Timestamp"]) - t0) / 1e3,
          (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Stream_Id", "?")))
PY
