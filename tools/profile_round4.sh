set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4p
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4p/smoke.log 2>&1
for C in ${CONFIGS:-C2 C3 C4 C5}; do
  bash tools/profile_r.sh r4_$C $C > gpurun_out/r4p/profile_$C.log 2>&1
  cp gpurun_out/prof_r4_$C/summary.txt gpurun_out/r4p/round4_${C}_rocprofv3_summary.txt
done
# the short-batch kernels: a kernel trace of the two micro benchmarks
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/sb && rocprofv3 --kernel-trace --stats -d /tmp/sb --output-format csv -- python tools/micro/wave_paths.py > /tmp/sb.log 2>&1
rm -rf /tmp/sp && rocprofv3 --kernel-trace --stats -d /tmp/sp --output-format csv -- python tools/micro/small_pairs.py > /tmp/sp.log 2>&1
python - > gpurun_out/r4p/round4_short_batch_kernels.txt <<'PY'
import csv, glob
for tag, d in (("tools/micro/wave_paths.py", "/tmp/sb"), ("tools/micro/small_pairs.py", "/tmp/sp")):
    print("== rocprofv3 --kernel-trace --stats --", tag, "(the library's kernels; Name, Calls, AverageNs, MinNs, MaxNs)")
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "atr::" in r["Name"]:
                print(r["Name"][:110], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
tail -3 gpurun_out/r4p/smoke.log; ls gpurun_out/r4p
