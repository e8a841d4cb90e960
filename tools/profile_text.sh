#!/bin/bash
# rocprofv3 kernel-trace summaries of the text pipelines (HBM-resident FASTQ in -> FASTQ out) and of the
# pair aligner: tools/profile_text.sh <tag>   (on the GPU box; outputs under gpurun_out/)
set -u
TAG=${1:-r2t}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() {   # name, command...
  local name=$1; shift
  local out=/tmp/pt_${TAG}_${name}
  rm -rf $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -- "$@" > gpurun_out/${TAG}_${name}.json 2> $out.err
  python - $out "$*" > gpurun_out/${TAG}_${name}_kernels.txt <<'PY'
import csv, glob, os, sys
print("== rocprofv3 --kernel-trace --stats -- %s   (the library's kernels; torch set-up kernels folded into one line)" % sys.argv[2])
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    mine = [r for r in rows if "atr::" in r["Name"]]
    print("%-78s %6s %12s %12s %7s" % ("kernel", "calls", "avg_us", "total_ms", "share"))
    for r in mine:
        print("%-78s %6s %12.1f %12.3f %6.1f%%" % (r["Name"].split("(")[0][-78:], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
    other = tot - sum(float(r["TotalDurationNs"]) for r in mine)
    print("%-78s %6s %12s %12.3f %6.1f%%" % ("(other: torch / runtime kernels)", "", "", other / 1e6, 100 * other / tot))
PY
  tail -c 700 gpurun_out/${TAG}_${name}.json; echo; head -30 gpurun_out/${TAG}_${name}_kernels.txt
}
run fastq_se python tools/bench_fastq.py 10000000 3
run fastq_pe_c3 python tools/bench_fastq_pe.py 5000000 3 C3
run fastq_pe_c5 python tools/bench_fastq_pe.py 1000000 3 C5
run fastq_pe_c3_merge python tools/bench_fastq_pe.py 2000000 3 C3 merge
run pairs python tools/bench_pairs.py
