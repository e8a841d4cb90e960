#!/usr/bin/env python3
"""Reduce the FETCH_SIZE / WRITE_SIZE sections of a tools/profile_r.sh summary into
profiles/traffic_<config>.json (the `roofline.traffic` figure of bench.py).

    python tools/reduce_traffic.py profiles/round2_c2_rocprofv3_summary.txt C2 10000000

Counter unit: KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3 section):
FETCH_SIZE reports half of a wide coalesced stream of 16 bytes per lane, so the kernels that stream
the packed batch that way (the pre-passes, the insert sweep) count 2 x FETCH_SIZE; every other
kernel counts the raw value -- calibrated as the guide asks on known byte counts: scatter_kernel
read 40.0 MB of window words + 8.4 MB of histogram with 4-byte loads and FETCH_SIZE said 52.3 MB;
pack_kernel wrote 800 MB and WRITE_SIZE said 800 MB."""
import json
import os
import re
import sys

STREAMING = ("filter_kernel", "insert_kernel")        # 16-byte-per-lane coalesced streams of the packed batch
# stream AND gather (bench.py MIXED_KERNELS): raw FETCH_SIZE + the uncounted half of the known stream, bytes per unit
MIXED = {"insert_correct_kernel": 128.0, "atr_piece_spec": 40.0, "piece_filter_kernel": 40.0}


def short(name):
    if name.startswith("atr_piece"):
        return name.strip()
    m = re.search(r"atr::(\w+)(<[^>]*>)?", name)
    return "atr::" + m.group(1) + (m.group(2) or "")


def main():
    src = sys.argv[1]
    config = sys.argv[2] if len(sys.argv) > 2 else "C2"
    units = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000
    # bench.py calls per PMC pass (tools/profile_r.sh: --steps 3 --warmup 1); kernels launched several
    # times per call (the per-adapter band / window kernels of C4) are summed per call
    calls = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    per = {}
    counter = None
    kernel = None
    for line in open(src):
        if line.startswith("== pmc"):
            counter = "FETCH_SIZE" if "pmc_FETCH_SIZE" in line else "WRITE_SIZE" if "pmc_WRITE_SIZE" in line else None
            continue
        if line.startswith("=="):
            counter = None
        if counter is None:
            continue
        if ("atr::" in line or line.startswith("atr_piece")) and not line.startswith(" "):
            kernel = short(line)
        m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE): launches=(\d+) avg=([\d.]+)", line)
        if m and kernel:
            per.setdefault(kernel, {})[m.group(1)] = float(m.group(3)) * int(m.group(2)) / calls
    # VALU wave-instructions per launch (SQ_INSTS_VALU pass), for the issue-bound view of the roofline
    valu = {}
    kernel = None
    section = False
    for line in open(src):
        if line.startswith("== pmc"):
            section = "pmc_SQ_WAVES" in line
            continue
        if line.startswith("=="):
            section = False
        if not section:
            continue
        if "atr::" in line and not line.startswith(" "):
            kernel = short(line)
        m = re.match(r"\s+SQ_INSTS_VALU: launches=(\d+) avg=([\d.]+)", line)
        if m and kernel:
            valu[kernel] = float(m.group(2)) * int(m.group(1)) / calls
    total = 0.0
    for k, v in per.items():
        half = next((h for key, h in MIXED.items() if key in k), None)
        if half is not None:
            total += (v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0 + half * units
            continue
        f = 2.0 if any(s in k for s in STREAMING) else 1.0
        total += (f * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0
    out = {"workload": config, "units_per_launch": units, "hbm_bytes_per_launch": total,
           "hbm_bytes_per_unit": total / units,
           "source": "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)" % src,
           "method": "(2*FETCH_SIZE + WRITE_SIZE) KiB for the kernels that stream the packed batch with 16-byte loads per lane "
                     "(gfx950 FETCH_SIZE counts half of such a stream), raw FETCH_SIZE + WRITE_SIZE for all others (calibrated: "
                     "scatter_kernel 48.4 MB of 4-byte loads read as 52.3 MB)",
           "per_kernel": per,
           "valu_wave_insts_per_launch": sum(valu.values()), "valu_wave_insts_per_kernel": valu}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "traffic_%s.json" % config), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({k: out[k] for k in ("hbm_bytes_per_launch", "hbm_bytes_per_unit")}))


if __name__ == "__main__":
    main()
