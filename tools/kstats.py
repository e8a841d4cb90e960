#!/usr/bin/env python3
"""Resource counts and an instruction census of every kernel in a gfx950 assembly file (hipcc -S --cuda-device-only):
SGPR / VGPR counts, spills, scratch, LDS, and the static number of instructions per class inside each kernel body."""
import collections
import re
import sys

POISON = ("v_alignbit", "v_min_u32", "v_max_u32", "v_min_i32", "v_max_i32", "v_lshlrev_b32", "v_lshl_add", "v_lshl_or", "v_bfe", "v_cndmask",
          "v_mad_u32", "v_min3", "v_max3", "v_readlane", "v_writelane", "v_readfirstlane", "v_mov_b32_dpp", "v_perm", "v_mul", "v_ffb", "v_bcnt", "v_lshlrev_b64", "v_lshrrev_b64")


def main(path, want=None):
    text = open(path).read()
    meta = text[text.index("amdhsa.kernels"):] if "amdhsa.kernels" in text else ""
    info = {}
    for k in meta.split("- .agpr_count")[1:]:
        g = lambda f: re.search(r"\." + f + r":\s+(\S+)", k).group(1)
        info[g("name")] = dict(sgpr=g("sgpr_count"), sspill=g("sgpr_spill_count"), vgpr=g("vgpr_count"), vspill=g("vgpr_spill_count"),
                               scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size"))
    for name, d in info.items():
        if want and want not in name:
            continue
        m = re.search(r"^%s:.*?\n(.*?)^\s*s_endpgm" % re.escape(name), text, re.S | re.M)
        census = collections.Counter()
        if m:
            for line in m.group(1).splitlines():
                t = line.strip().split()
                if not t or t[0].endswith(":") or t[0].startswith((".", ";")):
                    continue
                op = t[0]
                cls = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
                census[cls] += 1
                if op.startswith(POISON):
                    census["valu_4cycle_class"] += 1
                if op in ("s_nop", "s_waitcnt", "s_cbranch_scc1", "s_cbranch_scc0", "s_branch", "s_cbranch_vccz", "s_cbranch_vccnz", "s_cbranch_execz", "s_cbranch_execnz", "s_bitcmp1_b32"):
                    census[op] += 1
        print("%s\n   sgpr %s (spilled %s)  vgpr %s (spilled %s)  scratch %s B  lds %s B" % (name[:110], d["sgpr"], d["sspill"], d["vgpr"], d["vspill"], d["scratch"], d["lds"]))
        print("   static: " + "  ".join("%s %d" % kv for kv in sorted(census.items())))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
