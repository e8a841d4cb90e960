cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_locate.py -m gpu -x -q 2>&1 | tail -1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc_x_$C -- python bench.py --config C2 --no-cpu-baseline --no-secondary --steps 3 --warmup 1 > /dev/null 2>&1
  f=$(find gpurun_out/pmc_x_$C -name "*counter_collection.csv" | head -1)
  python - "$f" $C <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if "atr::" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]:
        acc[r["Kernel_Name"].split("(")[0][-40:]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(sys.argv[2], k, len(v), sum(v) / len(v))
PY
  rm -rf gpurun_out/pmc_x_$C
done
