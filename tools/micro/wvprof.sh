cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf /tmp/wv && rocprofv3 --kernel-trace --stats -d /tmp/wv --output-format csv -- python tools/micro/wave_paths.py > /tmp/wv.log 2>&1; tail -9 /tmp/wv.log; python - <<PY
import csv,glob
for f in glob.glob("/tmp/wv/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wave" in r["Name"]:
            print(r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
