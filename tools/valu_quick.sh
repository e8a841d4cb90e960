# SQ_INSTS_VALU per kernel of one bench.py config (GPU box): bash tools/valu_quick.sh [config]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
C=${1:-C2}
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d gpurun_out/pmc_v -- python bench.py --config $C --no-cpu-baseline --no-secondary --steps 3 --warmup 1 > /dev/null 2>&1
f=$(find gpurun_out/pmc_v -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "atr::" in r["Kernel_Name"]:
        acc[r["Kernel_Name"].split("(")[0][-44:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, {c: round(sum(x) / len(x) / 1e6, 2) for c, x in v.items()})
PY
rm -rf gpurun_out/pmc_v
