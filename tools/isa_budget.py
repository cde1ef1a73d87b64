#!/usr/bin/env python3
"""Static issue budget of one kernel from the compiler's assembly listing (hipcc -S --cuda-device-only):
vector instructions by issue class (2 cycles per wave64 instruction: f32 add/mul/fma, moves, 32-bit add/sub,
and/or/xor, right shifts; 4 cycles: everything else, v_rcp 8 -- the classes tools/micro/valu_classes.hip
measures), scalar ALU instructions, LDS and memory instructions, split at the workgroup barriers.

    python tools/isa_budget.py <listing.s> <kernel symbol substring> [--phases]
"""
import re
import sys

TWO = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_add_u32",
       "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32",
       "v_not_b32", "v_fmamk_f32", "v_fmaak_f32", "v_bitop3_b32", "v_mac_f32", "v_madak_f32", "v_madmk_f32",
       "v_add_co_u32", "v_sub_co_u32", "v_addc_co_u32", "v_xnor_b32", "v_accvgpr_write_b32", "v_accvgpr_read_b32",
       "v_readfirstlane_b32", "v_readlane_b32", "v_writelane_b32", "v_nop"}


def classify(m):
    base = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", m)
    form = m[len(base):]
    if m.startswith("v_"):
        if base == "v_rcp_f32":
            return "v8"
        if base in TWO and form in ("", "_e32", "_e64"):
            return "v2"
        return "v4"
    if m.startswith("s_"):
        if m in ("s_nop", "s_waitcnt", "s_barrier", "s_endpgm", "s_sleep", "s_setprio", "s_waitcnt_vscnt"):
            return "s1"
        if m.startswith("s_cbranch") or m == "s_branch":
            return "sbr"
        if m.startswith("s_load") or m.startswith("s_buffer_load") or m == "s_memtime":
            return "smem"
        return "salu"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    phases = "--phases" in sys.argv
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^[A-Za-z_][\w$.]*:", l) and key in l.split(":")[0]:
            start = i
            break
    if start is None:
        raise SystemExit("kernel not found")
    sects, cur = [], {}
    hist = {}
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".set ") or s.startswith(".Lfunc_end"):
            break
        m = re.match(r"^([a-z][a-z0-9_]+)\b", s)
        if not m:
            continue
        op = m.group(1)
        if not op.startswith(("v_", "s_", "ds_", "buffer_", "global_", "flat_", "scratch_")):
            continue
        c = classify(op)
        cur[c] = cur.get(c, 0) + 1
        hist[op] = hist.get(op, 0) + 1
        if op == "s_barrier":
            sects.append(cur)
            cur = {}
    sects.append(cur)
    tot = {}
    for s in sects:
        for k, v in s.items():
            tot[k] = tot.get(k, 0) + v

    def show(name, s):
        v2, v4, v8 = s.get("v2", 0), s.get("v4", 0), s.get("v8", 0)
        cyc = 2 * v2 + 4 * v4 + 8 * v8
        print("%-14s valu %5d (2-cycle %5d, 4-cycle %5d, 8-cycle %3d) = %6d issue cycles; salu %4d branch %3d lds %4d vmem %3d wait/nop %3d"
              % (name, v2 + v4 + v8, v2, v4, v8, cyc, s.get("salu", 0), s.get("sbr", 0), s.get("lds", 0), s.get("vmem", 0),
                 s.get("s1", 0)))
    if phases:
        for i, s in enumerate(sects):
            show("section %d" % i, s)
    show("static total", tot)
    if "--hist" in sys.argv:
        for k, v in sorted(hist.items(), key=lambda kv: -kv[1])[:60]:
            print("  %5d %s (%s)" % (v, k, classify(k)))


if __name__ == "__main__":
    main()
