#!/usr/bin/env python3
"""VALU instruction mix of the kernels in a hipcc -S listing, by issue class (profiles/round4_valu_issue_sparse.txt).
    hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S x.hip -o x.s && python tools/isa_mix.py x.s [name-substring]"""
import collections
import re
import sys

FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32",
        "v_lshrrev_b32", "v_ashrrev_i32", "v_bitop3_b32", "v_min_u16", "v_max_u16", "v_add_u16", "v_sub_u16",
        "v_lshlrev_b16", "v_lshrrev_b16"}
BENIGN = {"v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_subbrev_co_u32", "v_subrev_co_u32"}


def main():
    txt = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    for m in re.finditer(r"^(\w+):\s*; @\1\n(.*?)s_endpgm", txt, re.M | re.S):
        name, body = m.group(1), m.group(2)
        if want not in name:
            continue
        ops = collections.Counter()
        for line in body.splitlines():
            f = line.split()
            if f and f[0].startswith(("v_", "ds_", "s_set_gpr", "global_", "buffer_")):
                ops[re.sub(r"_e(32|64)$|_sdwa$|_dpp$", "", f[0])] += 1
        valu = {k: v for k, v in ops.items() if k.startswith("v_")}
        fast = sum(v for k, v in valu.items() if k in FAST)
        ben = sum(v for k, v in valu.items() if k in BENIGN)
        tot = sum(valu.values())
        print("%s\n  VALU %d: fast %d, carry %d, other (4-cycle, possibly poisoning) %d" % (name[:90], tot, fast, ben, tot - fast - ben))
        print("  other:", sorted(((k, v) for k, v in valu.items() if k not in FAST and k not in BENIGN), key=lambda x: -x[1]))
        print("  mem:", sorted(((k, v) for k, v in ops.items() if not k.startswith("v_")), key=lambda x: -x[1]))


main()
