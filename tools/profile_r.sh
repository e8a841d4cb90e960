#!/bin/bash
# Profiling recipe run on the GPU box (via gpurun): kernel trace + stats, then PMC
# passes in separate runs (rocprofv3 must not combine --pmc with trace domains), then
# keep only the locate_kernel rows so the merged gpurun_out/ stays small.
# usage: tools/profile_r.sh <tag> [config] [extra bench.py args]
set -u
TAG=${1:-r01}
CONFIG=${2:-C2}
EXTRA=${3:-}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
BENCH="python bench.py --config $CONFIG --no-cpu-baseline --no-secondary $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH --steps 10 --warmup 2 > $OUT/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | cut -d' ' -f1)
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -- $BENCH --steps 3 --warmup 1 > $OUT/pmc_$N.log 2>&1
done
# reduce
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
def find(pat):
    return glob.glob(os.path.join(out, pat), recursive=True)
with open(os.path.join(out, "summary.txt"), "w") as fh:
    # the bench line printed by the TRACED run itself: the kernel averages below belong to this ms_per_step (tracing
    # adds a little to both; compare them with each other, not with the untraced line of another lease)
    try:
        import json
        line = [l for l in open(os.path.join(out, "trace.log")).read().splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        fh.write("== bench line of this traced run: value=%.4g %s ms_per_step=%.4f kernel_ms(HIP events)=%.4f steps=%d\n" % (
            d["value"], d["unit"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["steps"]))
    except Exception as e:                                        # noqa: BLE001
        fh.write("== (no bench line in trace.log: %r)\n" % (e,))
    for f in find("trace/**/*kernel_stats.csv"):
        fh.write("== kernel_stats (%s): the library's kernels; everything else (torch kernels of the synthetic-data\n"
                 "   generation and of bench.py's set-up, outside the timed steps) folded into one line\n" % f)
        rows = list(csv.reader(open(f)))
        fh.write(",".join(rows[0]) + "\n")
        other_calls = other_ns = 0
        for r in rows[1:]:
            if ("atr::" in r[0] or "atr_piece" in r[0]):
                fh.write(",".join('"%s"' % x if i == 0 else x for i, x in enumerate(r)) + "\n")
            else:
                other_calls += int(r[1]); other_ns += int(r[2])
        fh.write('"(other: torch / runtime kernels, set-up only)",%d,%d\n' % (other_calls, other_ns))
    for f in find("trace/**/*kernel_trace.csv"):
        rows = list(csv.DictReader(open(f)))
        durs = collections.defaultdict(list)
        meta = {}
        for r in rows:
            name = r["Kernel_Name"]
            if ("atr::" in name or "atr_piece" in name):
                durs[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                meta[name] = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")}
        fh.write("== kernel_trace reduced\n")
        for name, d in durs.items():
            fh.write("%s\n  launches=%d avg_ns=%.0f min_ns=%d max_ns=%d %s\n" % (name, len(d), sum(d) / len(d), min(d), max(d), meta[name]))
    for f in find("pmc_*/**/*counter_collection.csv"):
        rows = list(csv.DictReader(open(f)))
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            if ("atr::" in r["Kernel_Name"] or "atr_piece" in r["Kernel_Name"]) and "pack_kernel" not in r["Kernel_Name"]:
                acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        fh.write("== pmc %s\n" % f)
        for name, cs in acc.items():
            fh.write("%s\n" % name)
            for c, v in cs.items():
                fh.write("  %s: launches=%d avg=%.1f sum=%.1f\n" % (c, len(v), sum(v) / len(v), sum(v)))
PY
# drop the bulky raw files
find $OUT -name "*.csv" -size +200k -delete
find $OUT -name "*.csv" | grep -v -e kernel_stats -e domain_stats | xargs rm -f
cat $OUT/summary.txt | head -80
du -sh $OUT
