#!/bin/bash
# PMC counters of the library's kernels for an arbitrary command (own rocprofv3 pass, no trace domains):
#   tools/pmc_cmd.sh "<counters>" <command ...>
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
CTRS=$1; shift
OUT=/tmp/pmc_$$
rocprofv3 --pmc $CTRS --output-format csv -d $OUT -- "$@" > $OUT.log 2>&1
python - $OUT <<'PY'
import csv, glob, sys, os, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if ("atr::" in r["Kernel_Name"] or "atr_piece" in r["Kernel_Name"]) and "pack_kernel" not in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][-50:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in acc.items():
    print(name, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())), "launches=%d" % len(next(iter(cs.values()))))
PY
tail -n 1 $OUT.log | cut -c1-200
