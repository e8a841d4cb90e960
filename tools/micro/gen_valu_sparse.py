#!/usr/bin/env python3
"""Generates tools/micro/valu_sparse.hip: the round-4 issue-rate measurements (verdict item 2a).

  * A^n S streams, n = 1, 3, 7, 15, 31: n two-cycle ops (v_add_u32 / v_xor_b32 / v_and_b32 / v_or_b32 in rotation)
    followed by ONE op of the four-cycle class S, for S in {v_alignbit_b32, v_add_co_u32, v_bitop3_b32 (three
    distinct VGPRs), v_add_u32 with an SGPR operand, v_min_u32, v_lshlrev_b32, v_lshl_add_u32};
  * the same streams at 1, 2, 4 and 8 waves per SIMD (is the slow mode a property of the wave or of the SIMD?);
  * operand-count probes: three-operand ops whose sources repeat, rotate (v_alignbit x, x), ops with literals;
  * 16-bit ops, v_lshlrev_b32 by a literal, v_add x, x (the shift the Myers recurrence needs);
  * the indexed v_mov (s_set_gpr_idx_on / v_mov_b32 / s_set_gpr_idx_off) and ds_read_b32 next to two-cycle ops.
Every kernel: 8 independent chains (a0..a7), one asm block of BODY instructions, ITERS iterations.
"""
import sys

A_OPS = ["v_add_u32 {d}, {d}, {b}", "v_xor_b32 {d}, {d}, {b}", "v_and_b32 {d}, {d}, {c}", "v_or_b32 {d}, {d}, {b}"]
S_OPS = {
    "alignbit": "v_alignbit_b32 {d}, {d}, {b}, 31",
    "add_co": "v_add_co_u32 {d}, vcc, {d}, {b}",
    "bitop3": "v_bitop3_b32 {d}, {d}, {b}, {c} bitop3:0xde",
    "add_sgpr": "v_add_u32 {d}, {s}, {d}",
    "min_u32": "v_min_u32 {d}, {d}, {b}",
    "lshl_lit": "v_lshlrev_b32 {d}, 1, {d}",
    "lshl_add": "v_lshl_add_u32 {d}, {d}, 1, {b}",
}
SINGLE = {   # one op type per kernel (8 chains): name -> template
    "add_xx (x+x)": "v_add_u32 {d}, {d}, {d}",
    "lshlrev lit1": "v_lshlrev_b32 {d}, 1, {d}",
    "lshrrev lit31": "v_lshrrev_b32 {d}, 31, {d}",
    "ashrrev lit31": "v_ashrrev_i32 {d}, 31, {d}",
    "sub": "v_sub_u32 {d}, {d}, {b}",
    "subrev": "v_subrev_u32 {d}, {b}, {d}",
    "and_lit": "v_and_b32 {d}, 0x7f7f7f7f, {d}",
    "or_inline": "v_or_b32 {d}, 1, {d}",
    "bitop3 x,b,b": "v_bitop3_b32 {d}, {d}, {b}, {b} bitop3:0xde",
    "bitop3 x,x,b": "v_bitop3_b32 {d}, {d}, {d}, {b} bitop3:0xde",
    "bitop3 x,b,c": "v_bitop3_b32 {d}, {d}, {b}, {c} bitop3:0xde",
    "bitop3 x,b,lit": "v_bitop3_b32 {d}, {d}, {b}, 15 bitop3:0xde",
    "and_or x,b,b": "v_and_or_b32 {d}, {d}, {b}, {b}",
    "and_or x,b,c": "v_and_or_b32 {d}, {d}, {b}, {c}",
    "add3 x,b,b": "v_add3_u32 {d}, {d}, {b}, {b}",
    "alignbit x,x (rot)": "v_alignbit_b32 {d}, {d}, {d}, 31",
    "alignbit x,b": "v_alignbit_b32 {d}, {d}, {b}, 31",
    "alignbit x,b,vgpr": "v_alignbit_b32 {d}, {d}, {b}, {c}",
    "min_u16": "v_min_u16 {d}, {d}, {b}",
    "max_u16": "v_max_u16 {d}, {d}, {b}",
    "add_u16": "v_add_u16 {d}, {d}, {b}",
    "sub_u16": "v_sub_u16 {d}, {d}, {b}",
    "lshlrev_b16": "v_lshlrev_b16 {d}, 1, {d}",
    "lshrrev_b16": "v_lshrrev_b16 {d}, 1, {d}",
    "min_i32": "v_min_i32 {d}, {d}, {b}",
    "max_u32": "v_max_u32 {d}, {d}, {b}",
    "bfe_u32": "v_bfe_u32 {d}, {d}, 3, 8",
    "bcnt": "v_bcnt_u32_b32 {d}, {d}, {b}",
    "ffbl": "v_ffbl_b32 {d}, {d}",
    "cndmask sgpr": "v_cndmask_b32 {d}, {d}, {b}, s[6:7]",
    "mov": "v_mov_b32 {d}, {b}",
    "mov_dpp shr1": "v_mov_b32_dpp {d}, {d} wave_shr:1 row_mask:0xf bank_mask:0xf",
    "add sdwa": "v_add_u32_sdwa {d}, {d}, {b} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD",
}

def chain(i):
    return "%%%d" % (i % 8)

def emit_kernel(name, lines, extra_clobber=""):
    body = "\\n\\t".join(lines) + "\\n\\t"
    return ('__global__ __launch_bounds__(256) void %s(uint32_t *out, int iters, uint32_t seed) {\n'
            '    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;\n'
            '    uint32_t b = seed | 1u, c = seed * 3u + 5u;\n'
            '    uint32_t sg = __builtin_amdgcn_readfirstlane(seed + 1u);\n'
            '    for (int i = 0; i < iters; ++i)\n'
            '        asm volatile("%s" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) '
            ': "v"(b), "v"(c), "s"(sg) : "vcc", "s4", "s5", "s6", "s7", "scc"%s);\n'
            '    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);\n}\n'
            % (name, body, extra_clobber))

def fmt(t, i):
    return t.format(d=chain(i), b="%8", c="%9", s="%10")

kernels = []   # (kname, label, nvalu, group)
src = []

def add(label, lines, nvalu, group):
    kname = "k%d" % len(kernels)
    kernels.append((kname, label, nvalu, group))
    src.append(emit_kernel(kname, lines))

# pure streams
add("A only", [fmt(A_OPS[i % 4], i) for i in range(64)], 64, "mix")
for sname, st in S_OPS.items():
    add("S only: " + sname, [fmt(st, i) for i in range(64)], 64, "mix")
    for n in (1, 3, 7, 15, 31):
        lines = []
        total = 64 if n < 31 else 64
        i = 0
        while len(lines) < total:
            for _ in range(n):
                lines.append(fmt(A_OPS[i % 4], i)); i += 1
            lines.append(fmt(st, i)); i += 1
        add("A^%d S: %s" % (n, sname), lines, len(lines), "mix")
for label, t in SINGLE.items():
    add("single: " + label, [fmt(t, i) for i in range(64)], 64, "single")
# indexed mov: s_set_gpr_idx_on / v_mov / off  (index 0: reads the register itself)
lines = []
for i in range(32):
    lines += ["s_set_gpr_idx_on s4, gpr_idx(SRC0)", "v_mov_b32 %s, %s" % (chain(i), chain(i)), "s_set_gpr_idx_off", fmt(A_OPS[i % 4], i + 1)]
add("indexed v_mov + A (pairs)", ["s_mov_b32 s4, 0"] + lines, 64, "special")
lines = []
for i in range(16):
    lines += ["s_set_gpr_idx_on s4, gpr_idx(SRC0)", "v_mov_b32 %s, %s" % (chain(i), chain(i)), "s_set_gpr_idx_off"] + [fmt(A_OPS[(i + j) % 4], i + 1 + j) for j in range(3)]
add("indexed v_mov + AAA", ["s_mov_b32 s4, 0"] + lines, 64, "special")
lines = []
for i in range(64):
    lines += ["s_set_gpr_idx_on s4, gpr_idx(SRC0)", "v_mov_b32 %s, %s" % (chain(i), chain(i)), "s_set_gpr_idx_off"]
add("indexed v_mov only", ["s_mov_b32 s4, 0"] + lines, 64, "special")
# Myers column, two codings (synthetic eq = b rotated by the loop; no memory): ops counted as VALU per column
cur = []   # current form: bitop3 + carry chain + alignbit + lshl_add + min  (state pv=%0 mv=%1 score=%2 hits=%3 best=%4, temps %5 %6 %7)
for col in range(4):
    cur += [
        "v_or_b32 %5, %8, %1",            # xv = eq | mv
        "v_and_b32 %6, %8, %0",           # t = eq & pv
        "v_add_u32 %6, %6, %0",           # s = t + pv
        "v_bitop3_b32 %6, %6, %0, %8 bitop3:0xbe",   # xh = (s ^ pv) | eq
        "v_bitop3_b32 %7, %1, %6, %0 bitop3:0xf1",   # ph = mv | ~(xh | pv)   (LUT value immaterial for timing)
        "v_and_b32 %6, %0, %6",           # mh = pv & xh
        "v_add_co_u32 %7, s[6:7], %7, %7",
        "v_add_co_u32 %6, vcc, %6, %6",
        "s_nop 0",
        "v_addc_co_u32 %2, s[6:7], 0, %2, s[6:7]",
        "v_subbrev_co_u32 %2, vcc, 0, %2, vcc",
        "v_bitop3_b32 %0, %6, %5, %7 bitop3:0xf1",   # pv = mh | ~(xv | ph)
        "v_and_b32 %1, %7, %5",           # mv = ph & xv
        "v_alignbit_b32 %3, %3, %2, 31",  # hits
        "v_lshl_add_u32 %5, %2, 10, %10", # key (SGPR tag)
        "v_min_u32 %4, %4, %5",
    ]
add("myers column, current coding (16 VALU)", cur, 64, "myers")
two = []   # two-cycle ops only (23 VALU per column)
for col in range(2):
    two += [
        "v_or_b32 %5, %8, %1",            # xv
        "v_and_b32 %6, %8, %0",           # t
        "v_add_u32 %6, %6, %0",           # s
        "v_xor_b32 %6, %6, %0",           # s ^ pv
        "v_or_b32 %6, %6, %8",            # xh
        "v_or_b32 %7, %6, %0",            # xh | pv
        "v_not_b32 %7, %7",
        "v_or_b32 %7, %7, %1",            # ph
        "v_and_b32 %6, %0, %6",           # mh
        "v_lshrrev_b32 %9, 31, %7",       # pout   (uses %9 = c as a temp: timing only)
        "v_add_u32 %2, %2, %9",
        "v_lshrrev_b32 %9, 31, %6",       # mout
        "v_sub_u32 %2, %2, %9",
        "v_add_u32 %7, %7, %7",           # ph << 1
        "v_add_u32 %6, %6, %6",           # mh << 1
        "v_or_b32 %0, %5, %7",            # xv | ph
        "v_not_b32 %0, %0",
        "v_or_b32 %0, %0, %6",            # pv
        "v_and_b32 %1, %7, %5",           # mv
        "v_add_u32 %3, %3, %3",           # hits <<= 1
        "v_lshrrev_b32 %9, 31, %2",
        "v_add_u32 %3, %3, %9",           # hits |= sign(score)
        "v_not_b32 %9, %2",               # exact-hit vector: second hit-bit word (D - 1) >> 31, folded later
        "v_lshrrev_b32 %9, 31, %9",
        "v_add_u32 %4, %4, %4",
        "v_add_u32 %4, %4, %9",
    ]
add("myers column, two-cycle ops only (26 VALU)", two, 52, "myers")


# round-4 coding of the column (filter_core.hpp, filter_step + filter_shift_out2): bitop3 + carry chains only.
# state pv=%0 mv=%1 score=%2 hits=%3 nz=%4, temps %5 %6 %7, eq = %8, k = %9 (c), carries s[6:7] and vcc
def new_col(st, tmp, nops=True, flags=True, sc="s[6:7]"):
    pv, mv, score, hits, nz = st
    xv, a, b2 = tmp
    L = ["v_or_b32 %s, %%8, %s" % (xv, mv),
         "v_and_b32 %s, %%8, %s" % (a, pv),
         "v_add_u32 %s, %s, %s" % (a, a, pv),
         "v_bitop3_b32 %s, %s, %s, %%8 bitop3:0xbe" % (a, a, pv),      # xh
         "v_bitop3_b32 %s, %s, %s, %s bitop3:0xf1" % (b2, mv, a, pv),  # ph
         "v_and_b32 %s, %s, %s" % (a, pv, a),                          # mh
         "v_add_co_u32 %s, %s, %s, %s" % (b2, sc, b2, b2),
         "v_add_co_u32 %s, vcc, %s, %s" % (a, a, a)]
    if nops: L.append("s_nop 0")
    L += ["v_addc_co_u32 %s, %s, 0, %s, %s" % (score, sc, score, sc),
          "v_subbrev_co_u32 %s, vcc, 0, %s, vcc" % (score, score)]
    if flags:
        L += ["v_add_co_u32 %s, %s, %s, %s" % (pv, sc, score, score),      # (pv, mv are dead here: used as the unread sums)
              "v_add_co_u32 %s, vcc, %s, %%9" % (mv, score)]
        if nops: L.append("s_nop 0")
        L += ["v_addc_co_u32 %s, %s, %s, %s, %s" % (hits, sc, hits, hits, sc),
              "v_addc_co_u32 %s, vcc, %s, %s, vcc" % (nz, nz, nz)]
    L += ["v_bitop3_b32 %s, %s, %s, %s bitop3:0xf1" % (pv, a, xv, b2),    # pv
          "v_and_b32 %s, %s, %s" % (mv, b2, xv)]                          # mv
    return L
S5 = ("%0", "%1", "%2", "%3", "%4"); T3 = ("%5", "%6", "%7")
nv = lambda L: sum(1 for x in L if x.startswith("v_"))
L = new_col(S5, T3) * 4
add("myers column, round-4 coding (16 VALU, 2 s_nop)", L, nv(L), "myers")
L = new_col(S5, T3, nops=False) * 4
add("  the same without the s_nops (hazard-unsafe: timing only)", L, nv(L), "myers")
L = new_col(S5, T3, flags=False) * 4
add("  the same without the two flag vectors (12 VALU)", L, nv(L), "myers")
L = new_col(S5, T3, sc="vcc", nops=True) * 4
add("  the same, every carry through vcc (serialised)", L, nv(L), "myers")
# "+v"(c) needed for the two-cycle coding: c is written.  Patch: make c an in/out operand everywhere (harmless).
out = ['// GENERATED by tools/micro/gen_valu_sparse.py -- do not edit.  Build and run on the GPU box:',
       '//   hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_sparse tools/micro/valu_sparse.hip && /tmp/valu_sparse',
       '#include <hip/hip_runtime.h>', '#include <stdio.h>', '#include <stdint.h>', '#include <string.h>']
out += [s.replace('"v"(b), "v"(c), "s"(sg)', '"v"(b), "v"(c), "s"(sg)') for s in src]
out.append('typedef void (*kern)(uint32_t *, int, uint32_t);')
out.append('struct K { kern k; const char *label; int nvalu; const char *group; };')
out.append('static const K KS[] = {')
for kname, label, nvalu, group in kernels:
    out.append('    {%s, "%s", %d, "%s"},' % (kname, label, nvalu, group))
out.append('};')
out.append(r'''
static double run(kern k, uint32_t *d, int nv, int blocks, int threads, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, iters, 3u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, iters, 3u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    // SIMD cycles per VALU instruction: waves per SIMD = blocks * (threads / 64) / 1024 (one round when <= 8)
    const double waves_per_simd = (double)blocks * (threads / 64) / 1024.0;
    const double valu_per_simd = waves_per_simd * iters * nv;
    return ms * 1e-3 * 2.4e9 / valu_per_simd;
}
int main(int argc, char **argv) {
    uint32_t *d;
    (void)hipMalloc(&d, 8192 * 256 * 4);
    const int n = sizeof(KS) / sizeof(KS[0]);
    printf("# SIMD cycles per VALU instruction at an assumed 2.4 GHz; columns: waves per SIMD 8 / 4 / 2 / 1\n");
    for (int i = 0; i < n; ++i) {
        if (argc > 1 && strcmp(argv[1], KS[i].group) != 0) continue;          // ./valu_sparse myers: one group only
        const bool all = strcmp(KS[i].group, "mix") == 0 || strcmp(KS[i].group, "myers") == 0 || strcmp(KS[i].group, "special") == 0;
        const double c8 = run(KS[i].k, d, KS[i].nvalu, 8192, 256, 2048);
        if (all) {
            const double c4 = run(KS[i].k, d, KS[i].nvalu, 4096, 256, 2048);
            const double c2 = run(KS[i].k, d, KS[i].nvalu, 2048, 256, 2048);
            const double c1 = run(KS[i].k, d, KS[i].nvalu, 1024, 256, 2048);
            printf("%-48s %6.2f %6.2f %6.2f %6.2f\n", KS[i].label, c8, c4, c2, c1);
        } else {
            printf("%-48s %6.2f\n", KS[i].label, c8);
        }
        fflush(stdout);
    }
    return 0;
}
''')
open(sys.argv[1] if len(sys.argv) > 1 else "tools/micro/valu_sparse.hip", "w").write("\n".join(out))
