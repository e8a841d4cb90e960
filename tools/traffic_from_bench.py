#!/usr/bin/env python3
"""profiles/traffic_<config>.json -- what bench.py falls back to when it cannot run its own PMC passes (N > 1,
--no-live-counters, no rocprofv3) -- from a default bench line whose traffic WAS measured live by that invocation.
usage: tools/traffic_from_bench.py profiles/round6_bench_default_all_configs.json"""
import json
import os
import sys

src = sys.argv[1]
line = json.loads(open(src).read().strip().splitlines()[-1])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
configs = {"C2": line}
configs.update(line.get("configs", {}))
for name, res in configs.items():
    roof = res.get("roofline", {})
    if not roof.get("traffic") or "this invocation" not in roof.get("traffic_source", ""):
        print(name, "no live traffic in this line: left alone")
        continue
    units = res["config"].get("reads_per_gpu") or res["config"].get("pairs_per_gpu")
    out = {"workload": name, "units_per_launch": units, "hbm_bytes_per_launch": roof["traffic"],
           "hbm_bytes_per_unit": roof["traffic"] / units,
           "source": "%s (the traffic bench.py measured live in that invocation: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                     "separate passes, the rule of bench.py's live_counters)" % os.path.relpath(os.path.abspath(src), root)}
    if roof.get("valu_wave_insts_per_launch"):
        out["valu_wave_insts_per_launch"] = roof["valu_wave_insts_per_launch"]
    with open(os.path.join(root, "profiles", "traffic_%s.json" % name), "w") as fh:
        json.dump(out, fh, indent=1)
    print(name, "%.3f GB per launch, %.1f B per unit" % (roof["traffic"] / 1e9, roof["traffic"] / units))
