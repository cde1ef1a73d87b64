"""Vector-ALU issue budget of the coding kernel.

Puts together (a) the issue cost of every instruction class on a gfx950 SIMD, measured by
tools/micro/valu_classes.hip (wave64: 2 cycles for the 32-lane-per-clock class, 4 cycles for the 16-lane class,
8 for reciprocals), (b) the kernel's dynamic instruction counts per class from `rocprofv3 --pmc`
(tools/pmc_coded.sh: SQ_INSTS_VALU and its ADD/MUL/FMA_F32 and CVT parts, which the hardware counts exactly) and
(c), for the instructions those counters do not tell apart, the share of 4-cycle instructions among them in the
kernel's ISA.  Result: issue cycles the kernel needs per image against the cycles its SIMDs have while it runs.

usage: python tools/valu_budget.py <valu_classes.txt> <pmc summary.txt> <kernel symbol substring> <out prefix>
       [--images-per-launch N] [--workload "..."]
"""
import argparse
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F32_ARITH = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32",
             "v_mac_f32", "v_madak_f32", "v_madmk_f32")


def measured_costs(path):
    """mnemonic -> cycles per wave-instruction per SIMD, rounded to the class (2, 4, 8)."""
    out, raw = {}, {}
    for line in open(path):
        m = re.match(r"(\S+)\s+\d+ waves/SIMD:\s+([\d.]+) ns", line)
        if not m:
            continue
        ns = float(m.group(2))
        raw[m.group(1)] = ns
        cls = 2 if ns < 1.3 else (4 if ns < 2.6 else (8 if ns < 5 else None))
        name = re.sub(r"_(sdwa|dpp|literal|sgpr|half_count|vcc_set_once|bytes|byte\d?|word\d|row_shr\d|like_mov_sgpr|op_sel|sgpr_pair_mask).*$", "", m.group(1))
        if cls:
            out.setdefault(name, (cls, ns))
        if "sdwa" in m.group(1):
            out["*sdwa"] = (cls, ns)
        if "dpp" in m.group(1):
            out["*dpp"] = (cls, ns)
    # v_cndmask_b32: the 4-byte encoding reads its mask from vcc and takes what is left of the pair
    # (v_cmp, v_cndmask) after the compare; the 8-byte encoding with a scalar register pair as mask is measured
    # on its own.  (A stream of 256 selects that all read a vcc written once by scalar instructions measures 23
    # cycles per instruction, as does a stream of v_nop: streams the kernels do not contain.)
    if "v_cmp_then_v_cndmask_b32_vcc" in raw and "v_cmp_lt_u32" in raw:
        ns = raw["v_cmp_then_v_cndmask_b32_vcc"] - raw["v_cmp_lt_u32"]
        out["v_cndmask_b32_e32"] = (2 if ns < 1.3 else 4, ns)
    if "v_cndmask_b32_sgpr_pair_mask" in raw:
        ns = raw["v_cndmask_b32_sgpr_pair_mask"]
        out["v_cndmask_b32_e64"] = (2 if ns < 1.3 else 4, ns)
    out.pop("v_cndmask_b32", None)
    out.pop("v_nop", None)
    return out


def kernel_asm(symbol):
    """ISA of the kernel, compiled with the library's own flags."""
    import importlib.util
    sp = importlib.util.spec_from_file_location("_jpeg_build", os.path.join(ROOT, "jpeg-encoder_amd", "build.py"))
    b = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(b)
    flags = [f for f in b.HIP_FLAGS if f not in ("-Wall",)]
    out = "/tmp/valu_budget_coded.s"
    subprocess.run([b.HIPCC] + flags + ["--offload-device-only", "-S", os.path.join(b.CSRC, "coded.hip"), "-o", out],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lines, on = [], False
    for l in open(out):
        if re.match(r"^_Z\S*%s\S*:" % re.escape(symbol), l):
            on = True
        if on:
            lines.append(l)
            if "s_endpgm" in l:
                break
    return lines


def static_classes(lines, costs):
    """(counts per mnemonic, cost per mnemonic) of the kernel's vector-ALU instructions."""
    counts, unknown = {}, set()
    for l in lines:
        m = re.match(r"\s+(v_\w+)", l)
        if not m:
            continue
        op = m.group(1)
        base = re.sub(r"_(e32|e64)$", "", op)
        kind = "sdwa" if base.endswith("_sdwa") else ("dpp" if base.endswith("_dpp") else "")
        base = re.sub(r"_(sdwa|dpp)$", "", base)
        if kind:
            cost = costs["*" + kind][0]
        elif base == "v_cndmask_b32" and ("v_cndmask_b32_e32" in costs or "v_cndmask_b32_e64" in costs):
            # the disassembly names the encoding; without a suffix the mask operand tells (vcc: 4-byte encoding)
            e64 = op.endswith("_e64") or (not op.endswith("_e32") and not re.search(r",\s*vcc\s*$", l.split(";")[0].strip()))
            cost = costs.get("v_cndmask_b32_e64" if e64 else "v_cndmask_b32_e32", (2, 0))[0]
            base += "_sgpr_mask" if e64 else "_vcc"
        elif base in costs and costs[base][0]:
            cost = costs[base][0]
        elif re.match(r"v_cvt_f32_ubyte\d", base):
            cost = costs["v_cvt_f32_ubyte0"][0]
        elif base.startswith("v_pk_") and not base.endswith("_f32") and "v_pk_add_u16" in costs:
            cost = costs["v_pk_add_u16"][0]                     # packed 16-bit integer forms: one class
        elif base.startswith("v_cmp") or base.startswith("v_cmpx"):
            cost = costs["v_cmp_lt_u32"][0]
        else:
            # not measured: priced with the cheaper class, so that the budget errs on the low side
            cost = 2
            unknown.add(base)
        key = base + ("_" + kind if kind else "")
        c = counts.setdefault(key, [0, cost])
        c[0] += 1
    return counts, sorted(unknown)


def pmc(path, kernel):
    vals, on = {}, False
    for l in open(path):
        if not l.startswith(" "):
            on = kernel in l
            continue
        m = re.match(r"\s+(\S+)\s+n=\s*\d+\s+mean=\s*([\d.]+)", l)
        if on and m:
            vals[m.group(1)] = float(m.group(2))
    return vals


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("classes"); ap.add_argument("pmc"); ap.add_argument("symbol"); ap.add_argument("out")
    ap.add_argument("--images-per-launch", type=int, default=4)
    ap.add_argument("--workload", default="7680x4320 4:2:0 real-dct")
    ap.add_argument("--kernel", default="k_coded")
    ap.add_argument("--simds", type=int, default=1024)
    ap.add_argument("--xcds", type=int, default=8)
    a = ap.parse_args()
    costs = measured_costs(a.classes)
    lines = kernel_asm(a.symbol)
    counts, unknown = static_classes(lines, costs)
    p = pmc(a.pmc, a.kernel)
    n = a.images_per_launch
    total = p["SQ_INSTS_VALU"]
    f32 = p["SQ_INSTS_VALU_ADD_F32"] + p["SQ_INSTS_VALU_MUL_F32"] + p["SQ_INSTS_VALU_FMA_F32"]
    cvt = p["SQ_INSTS_VALU_CVT"]
    rest = total - f32 - cvt
    # the rest: split by the ISA's share of 4-cycle instructions among everything that is neither f32 arithmetic nor a conversion
    r2 = sum(c for k, (c, cost) in counts.items() if cost == 2 and not k.startswith(F32_ARITH) and not k.startswith("v_cvt"))
    r4 = sum(c for k, (c, cost) in counts.items() if cost >= 4 and not k.startswith(F32_ARITH) and not k.startswith("v_cvt"))
    share4 = r4 / (r2 + r4)
    need = 2 * f32 + 4 * cvt + rest * (2 * (1 - share4) + 4 * share4)
    # bounds that do not depend on the static share: every other instruction in the cheaper / the dearer class
    need_lo, need_hi = 2 * f32 + 4 * cvt + 2 * rest, 2 * f32 + 4 * cvt + 4 * rest
    cycles = p["GRBM_GUI_ACTIVE"] / a.xcds                       # shader cycles the kernel ran (per XCD)
    have = a.simds * cycles
    # Scalar instructions: next to 2-cycle vector instructions a scalar ALU instruction takes 4 cycles of the SIMD's
    # issue time that no vector instruction of another wave fills (tools/micro/valu_classes.hip, the modes
    # "s_..._then_v_mul_f32": 6 cycles per pair at four waves per SIMD, against 2 for the vector instruction alone;
    # 3 for s_nop or s_waitcnt and a vector instruction); next to a 4-cycle vector instruction it takes 2.
    salu = p.get("SQ_INSTS_SALU", 0.0)
    need_with_salu = need + 3.0 * salu                          # between the 2 and the 4 cycles measured
    rec = {
        "workload": a.workload, "kernel": a.kernel, "images_per_launch_in_the_counter_run": n,
        "source": "tools/valu_budget.py over %s and %s" % (os.path.basename(a.classes), a.pmc),
        "issue_cost_cycles": {"32 lanes per clock (f32 add/mul/fma, mov, add/sub/and/or/xor, right shifts)": 2,
                              "16 lanes per clock (conversions, left shifts, compares, min/max, 3-operand integer, SDWA, DPP, perm, mul24/lo/hi)": 4,
                              "v_rcp_f32": 8},
        "per_launch": {"SQ_INSTS_VALU": total, "f32_add_mul_fma": f32, "conversions": cvt, "other": rest,
                       "share_of_4_cycle_instructions_among_other_in_the_ISA": round(share4, 4),
                       "issue_cycles_needed": need, "kernel_cycles": cycles, "issue_cycles_available": have},
        "valu_instructions_per_launch": total / n,            # per image (bench.py multiplies by its images per launch)
        "valu_issue_cycles_per_launch": need / n,
        "simds": a.simds, "clock_ghz": 2.4,
        "frac_in_the_counter_run": round(need / have, 4),
        "frac_bounds_other_all_2_or_all_4_cycles": [round(need_lo / have, 4), round(need_hi / have, 4)],
        "SQ_INSTS_VALU_INT32": p.get("SQ_INSTS_VALU_INT32"),
        "scalar_alu_instructions_per_launch": salu / n,
        "frac_in_the_counter_run_with_scalar_instructions_at_3_cycles": round(need_with_salu / have, 4) if salu else None,
        "mean_issue_cycles_per_instruction": round(need / total, 3),
        "not_measured_priced_at_2_cycles": unknown,
        "waves_per_simd": round(p.get("SQ_WAVE_CYCLES", 0) * 4 / have, 3) if p.get("SQ_WAVE_CYCLES") else None,
        "wave_time_waiting_for_memory_or_lds": round(p.get("SQ_WAIT_INST_ANY", 0) / p["SQ_WAVE_CYCLES"], 3) if p.get("SQ_WAVE_CYCLES") else None,
    }
    json.dump(rec, open(a.out + ".json", "w"), indent=1)
    with open(a.out + ".txt", "w") as f:
        f.write("Vector-ALU issue budget of %s, %s (launches of %d images)\n\n" % (a.kernel, a.workload, n))
        f.write("dynamic counts per launch (rocprofv3 --pmc):\n")
        f.write("  f32 add/mul/fma   %12.0f  x 2 cycles\n  conversions       %12.0f  x 4 cycles\n  other             %12.0f  x %.2f cycles (%.1f %% of them 4-cycle instructions in the ISA)\n"
                % (f32, cvt, rest, 2 * (1 - share4) + 4 * share4, 100 * share4))
        f.write("  total             %12.0f  -> %.1f M issue cycles needed; %d SIMDs x %.0f cycles = %.1f M available: %.3f\n\n"
                % (total, need / 1e6, a.simds, cycles, have / 1e6, need / have))
        if salu:
            f.write("scalar ALU instructions %12.0f per launch: next to vector instructions of another wave they take 4 cycles of the\n"
                    "  SIMD's issue time beside a 2-cycle vector instruction and 2 beside a 4-cycle one (tools/micro/valu_classes.hip,\n"
                    "  modes s_*_then_v_*); at 3 cycles each: %.1f M issue cycles needed in all: %.3f\n\n"
                    % (salu, need_with_salu / 1e6, need_with_salu / have))
        f.write("ISA of the kernel (static counts, all paths), vector-ALU instructions by issue cost:\n")
        for cost in (2, 4, 8):
            items = sorted(((c, k) for k, (c, cc) in counts.items() if cc == cost), reverse=True)
            if items:
                f.write("  %d cycles: %d instructions: %s\n" % (cost, sum(c for c, _ in items), ", ".join("%s %d" % (k, c) for c, k in items)))
        f.write("\nbounds (every other instruction at 2 / at 4 cycles): %.3f .. %.3f\n" % (need_lo / have, need_hi / have))
        f.write("instruction kinds in the ISA without a measurement (priced at 2 cycles): %s\n" % (", ".join(unknown) or "none"))
    print(open(a.out + ".txt").read())


if __name__ == "__main__":
    main()
