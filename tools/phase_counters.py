"""Counters of the coding kernel by phase, from counter runs of the full kernel and of the three ablated builds
(-DJPG_ABLATE=1: without the coding phase; =2: without transform and coding; =3: loads only, the pixels XOR-ed into
LDS): the difference between one build and the next is what the phase left out executes.

usage: python tools/phase_counters.py <full summary> <ablate1 summary> <ablate2 summary> <ablate3 summary>
"""
import re
import sys


def counters(path):
    vals, on = {}, False
    for l in open(path):
        if not l.startswith(" "):
            on = "k_coded" in l
            continue
        m = re.match(r"\s+(\S+)\s+n=\s*\d+\s+mean=\s*([\d.]+)", l)
        if on and m:
            vals[m.group(1)] = float(m.group(2))
    return vals


full, a1, a2, a3 = (counters(p) for p in sys.argv[1:5])
zero = {k: 0.0 for k in full}
phases = [("loads, table, window clearing", a3, zero), ("colour conversion + plane stores", a2, a3),
          ("transform (+ MCU before the tile)", a1, a2), ("mask, coding, placement, stores", full, a1)]
keys = ["SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_CVT",
        "SQ_INSTS_VALU_INT32", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS",
        "SQ_WAIT_INST_LDS"]
print("per launch of the counter run, millions; each row = what the phase adds to the build without it\n")
print("%-36s" % "phase" + "".join("%12s" % k.replace("SQ_INSTS_", "").replace("SQ_", "")[:11] for k in keys))
for name, hi, lo in phases:
    print("%-36s" % name + "".join("%12.3f" % ((hi.get(k, 0) - lo.get(k, 0)) / 1e6) for k in keys))
print("%-36s" % "whole kernel" + "".join("%12.3f" % (full.get(k, 0) / 1e6) for k in keys))
print()
for name, hi, lo in phases:
    act = hi.get("SQ_LDS_IDX_ACTIVE", 0) - lo.get("SQ_LDS_IDX_ACTIVE", 0)
    con = hi.get("SQ_LDS_BANK_CONFLICT", 0) - lo.get("SQ_LDS_BANK_CONFLICT", 0)
    print("%-36s LDS bank-conflict cycles %.3f M of %.3f M LDS cycles (%.1f %%); %.1f %% of the kernel's conflict cycles"
          % (name, con / 1e6, act / 1e6, 100 * con / act if act else 0, 100 * con / full["SQ_LDS_BANK_CONFLICT"]))
