"""Puts one record together for bench.py (roofline.traffic, roofline_valu) from a counter run of the coding kernel:
HBM bytes from FETCH_SIZE / WRITE_SIZE with the gfx950 correction, issue cycles from tools/valu_budget.py.

usage: python tools/pmc_record.py <pmc summary.txt> <budget.json> <bench line.json> <out.json> [--images-per-launch 4]
"""
import argparse
import json
import re


def counters(path, kernel):
    vals, on = {}, False
    for l in open(path):
        if not l.startswith(" "):
            on = kernel in l
            continue
        m = re.match(r"\s+(\S+)\s+n=\s*\d+\s+mean=\s*([\d.]+)", l)
        if on and m:
            vals[m.group(1)] = float(m.group(2))
    return vals


ap = argparse.ArgumentParser()
ap.add_argument("summary"); ap.add_argument("budget"); ap.add_argument("bench"); ap.add_argument("out")
ap.add_argument("--images-per-launch", type=int, default=4)
ap.add_argument("--kernel", default="k_coded")
ap.add_argument("--command", default="tools/pmc_configs.sh")
a = ap.parse_args()
p = counters(a.summary, a.kernel)
b = json.load(open(a.budget))
line = json.load(open(a.bench))
n = a.images_per_launch
roof = line["roofline"]
alg_image = roof["algorithmic_bytes"] / roof["images_per_launch"]
# FETCH_SIZE / WRITE_SIZE are in KB (1024 bytes); gfx950 tallies a 128-byte read request as 64 bytes, so
# FETCH_SIZE is doubled; WRITE_SIZE is taken as it is
rd = p["FETCH_SIZE"] * 1024 * 2 / n
wr = p["WRITE_SIZE"] * 1024 / n
rec = {
    "workload": b["workload"], "kernel": a.kernel,
    "command": "%s (tools/pmc_coded.sh: rocprofv3 --pmc, one process per counter set, over bench.py --steps 2 --warmup 1 "
               "--batch 8 --resident 4 --per-launch 4: launches of %d images); summary in %s" % (a.command, n, a.summary.replace("gpurun_out/", "profiles/")),
    "FETCH_SIZE_KB_per_launch": p["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": p["WRITE_SIZE"],
    "corrections": "FETCH_SIZE doubled (gfx950 tallies 128-byte read requests at 64 bytes); WRITE_SIZE taken as is; 1 KB = 1024 bytes",
    "hbm_read_bytes_per_launch": int(rd), "hbm_write_bytes_per_launch": int(wr), "hbm_bytes_per_launch": int(rd + wr),
    "algorithmic_bytes_per_launch": int(alg_image),
    "note": "*_per_launch figures are PER IMAGE (the counter run's launches cover %d images; bench.py multiplies by its own "
            "images per launch). Read: the pixel rows once, plus, per tile, the partial cache lines of the MCU before it. "
            "Written: the coded bits only." % n,
    "images_per_launch_in_the_counter_run": n,
    "valu_instructions_per_launch": b["valu_instructions_per_launch"],
    "valu_issue_cycles_per_launch": b["valu_issue_cycles_per_launch"],
    "simds": b["simds"], "clock_ghz": b["clock_ghz"],
    "source": "%s (tools/valu_budget.py: per-class counts of the counter run x issue costs from tools/micro/valu_classes.hip)" % a.budget.replace("gpurun_out/", "profiles/"),
    "valu": b["per_launch"],
    "frac_bounds_other_all_2_or_all_4_cycles": b.get("frac_bounds_other_all_2_or_all_4_cycles"),
    "scalar_alu_instructions_per_launch": b["scalar_alu_instructions_per_launch"],
    "scalar_issue_cycles_per_launch": 3.0 * b["scalar_alu_instructions_per_launch"],
    "scalar_note": "a scalar ALU instruction takes 4 cycles of its SIMD's issue time beside a 2-cycle vector instruction of "
                   "another wave and 2 beside a 4-cycle one (tools/micro/valu_classes.hip, modes s_*_then_v_*); priced at 3",
    "lds": {k: p.get(k) for k in ("SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS")},
    "lds_bank_conflict_share_of_lds_cycles": round(p["SQ_LDS_BANK_CONFLICT"] / p["SQ_LDS_IDX_ACTIVE"], 4) if p.get("SQ_LDS_IDX_ACTIVE") else None,
    "waves_per_simd": b.get("waves_per_simd"),
    "wave_time_waiting_for_memory_or_lds": b.get("wave_time_waiting_for_memory_or_lds"),
}
json.dump(rec, open(a.out, "w"), indent=1)
print("%s: traffic %.1f MB per image vs %.1f MB algorithmic (x%.3f); issue %.3f valu only" % (
    rec["workload"], (rd + wr) / 1e6, alg_image / 1e6, (rd + wr) / alg_image, b["frac_in_the_counter_run"]))
