#!/usr/bin/env python3
"""Static per-phase instruction budget of one search-kernel instantiation.

Compiles csrc/search_d<D>.hip for gfx950 with -DDR_PHASE_MARK (search_kernel.hpp: the PH(i) phase boundaries become
comments in the otherwise unchanged ISA; no GPU needed), cuts the chosen kernel's ISA at the marks and counts the
instructions of every phase by class, in program order, with the compiler's loop annotations: instructions inside a
loop nested deeper than the expansion loop are counted apart (they repeat per expansion).  STATIC counts: an expansion
executes a phase's straight-line part at most once (branches skip parts of it) and its inner loops several times; the
dynamic totals per expansion come from the SQ counters (profiles/r04/sq_counters.json) and are printed beside the sum.

  python3 scripts/phase_budget.py [--dim 128] [--kernel "128, true, 0, 2, 16, true, 64, true, true, 0"] [--out FILE]
"""
import argparse, json, pathlib, re, subprocess, sys, tempfile

ROOT = pathlib.Path(__file__).resolve().parent.parent
PHASES = ["setup+LUT", "pop/stop", "adjacency", "visited", "ADC", "exact", "decisions", "output"]


def classify(mn):
    if mn.startswith("v_dot") or mn.startswith("v_mfma"): return "valu_dot"
    if mn.startswith("v_readlane") or mn.startswith("v_readfirstlane") or mn.startswith("v_writelane"): return "valu_lane"
    if mn.startswith("v_cmp") or mn.startswith("v_cmpx"): return "valu_cmp"
    if mn.startswith("v_"): return "valu"
    if mn.startswith("ds_bpermute") or mn.startswith("ds_permute") or mn.startswith("ds_swizzle"): return "lds_permute"
    if mn.startswith("ds_"): return "lds"
    if mn.startswith(("global_", "buffer_", "flat_")): return "vmem"
    if mn.startswith("scratch_"): return "scratch"
    if mn.startswith(("s_load", "s_buffer_load", "s_store", "s_memtime", "s_dcache")): return "smem"
    if mn.startswith("s_waitcnt"): return "waitcnt"
    if mn.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_call")): return "branch"
    if mn.startswith(("s_barrier", "s_nop", "s_sleep", "s_endpgm", "s_setprio", "s_sethalt", "s_code_end")): return "misc"
    if mn.startswith("s_"): return "salu"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--kernel", default="128, true, 0, 2, 16, true, 64, true, true, 0",
                    help="template arguments of search_kernel<...> (default: variant 13 at D = 128, the bench kernel)")
    ap.add_argument("--asm", default=None, help="use this -S output instead of compiling")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    if args.asm:
        text = pathlib.Path(args.asm).read_text()
    else:
        with tempfile.TemporaryDirectory() as td:
            out = pathlib.Path(td) / "k.s"
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-DDR_PHASE_MARK",
                   "--cuda-device-only", "-S", str(ROOT / "diskrag_amd" / "csrc" / ("search_d%d.hip" % args.dim)), "-o", str(out)]
            if args.dim in (768, 960, 1536):
                cmd.insert(5, "-fno-slp-vectorize")
            subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
            text = out.read_text()
    # mangled name of search_kernel<...>(SearchParams): Li<n>E for ints, Lb<0|1>E for bools
    targs = [a.strip() for a in args.kernel.split(",")]
    mangled = "_Z13search_kernelI" + "".join(("Lb%dE" % (a == "true")) if a in ("true", "false") else ("Li%sE" % a) for a in targs) + "Ev12SearchParams"
    lines = text.splitlines()
    try:
        start = next(i for i, l in enumerate(lines) if l.startswith(mangled + ":"))
    except StopIteration:
        sys.exit("kernel %s not found" % mangled)
    end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i])
    body = lines[start:end]
    meta = {}
    for l in body:
        m = re.match(r"\s*\.amdhsa_(next_free_vgpr|next_free_sgpr|accum_offset|group_segment_fixed_size|private_segment_fixed_size)\s+(\S+)", l)
        if m: meta[m.group(1)] = m.group(2)
    for l in lines[end:end + 80]:
        m = re.match(r";\s*(NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize|codeLenInByte|SGPRBlocks|NumSgprs)[^:]*:\s*(\S+)", l)
        if m: meta[m.group(1)] = m.group(2)

    # walk: phase index = the phase whose END mark comes next in program order (before BEGIN: "prologue")
    cur, depth = "prologue", 0
    order = ["prologue"] + PHASES + ["epilogue"]
    counts = {k: {"top": {}, "inner": {}} for k in order}
    nxt = 0
    exp_depth = None       # loop depth of the expansion loop body = depth at the first "pop/stop" instruction
    for l in body:
        s = l.strip()
        if s.startswith("; DR_PHASE_BEGIN"):
            cur = PHASES[0]; continue
        m = re.match(r"; DR_PHASE_END (\d+)", s)
        if m:
            i = int(m.group(1)); cur = PHASES[i + 1] if i + 1 < len(PHASES) else "epilogue"; continue
        m = re.search(r"Depth=(\d+)", s)
        if s.startswith(";") or s.startswith(".LBB") or s.startswith("//"):
            if m and ("Loop Header" in s or "in Loop" in s or "Inner Loop" in s or "Parent Loop" in s or "Child Loop" in s):
                if "Child Loop" not in s and "Parent Loop" not in s:
                    depth = int(m.group(1))
            elif re.match(r"\.LBB\d+_\d+:", s) or s.startswith("; %bb."):
                if not m: depth = 0 if "Loop" not in s else depth
            continue
        if not s or s.startswith(".") or s.endswith(":"):
            continue
        mn = s.split()[0]
        if cur == "pop/stop" and exp_depth is None:
            exp_depth = depth
        # (the expansion loop's body is itself one loop deeper: the row is walked in chunks of 64 slots -- one chunk at R <= 64)
        where = "inner" if (exp_depth is not None and depth > exp_depth + 1 and cur in PHASES[1:7]) else "top"
        c = counts[cur][where]
        k = classify(mn)
        c[k] = c.get(k, 0) + 1
    def tot(c, keys): return sum(c.get(k, 0) for k in keys)
    VALU = ("valu", "valu_dot", "valu_lane", "valu_cmp"); SALU = ("salu", "branch", "waitcnt", "misc")
    rows = []
    for ph in order:
        t, n = counts[ph]["top"], counts[ph]["inner"]
        rows.append({"phase": ph, "valu": tot(t, VALU), "valu_inner_loops": tot(n, VALU), "salu": tot(t, SALU), "salu_inner_loops": tot(n, SALU),
                     "smem": tot(t, ("smem",)) + tot(n, ("smem",)), "lds": tot(t, ("lds", "lds_permute")), "lds_inner_loops": tot(n, ("lds", "lds_permute")),
                     "vmem": tot(t, ("vmem",)) + tot(n, ("vmem",)), "scratch": tot(t, ("scratch",)) + tot(n, ("scratch",)),
                     "detail_top": t, "detail_inner": n})
    res = {"kernel": "search_kernel<%s>" % args.kernel, "mangled": mangled, "meta": meta, "expansion_loop_depth": exp_depth, "phases": rows,
           "note": "static counts in program order between the PH(i) marks (-DDR_PHASE_MARK); *_inner_loops = inside loops nested deeper than the expansion loop"}
    if args.out:
        pathlib.Path(args.out).write_text(json.dumps(res, indent=1))
    print("kernel search_kernel<%s>  %s" % (args.kernel, meta))
    print("%-10s %6s %8s %6s %8s %5s %5s %8s %5s" % ("phase", "VALU", "(+loops)", "SALU", "(+loops)", "SMEM", "LDS", "(+loops)", "VMEM"))
    for r in rows:
        print("%-10s %6d %8d %6d %8d %5d %5d %8d %5d" % (r["phase"], r["valu"], r["valu_inner_loops"], r["salu"], r["salu_inner_loops"], r["smem"], r["lds"],
                                                     r["lds_inner_loops"], r["vmem"]))
    # the ADC mark sits on a conditional path (`if (need_adc)`): the compiler lays the row loads and dot products of the "exact"
    # phase out on either side of it, so the two are only meaningful together
    ra, rb = rows[order.index("ADC")], rows[order.index("exact")]
    print("%-10s %6d %8d %6d %8d %5d %5d %8d %5d   <- ADC + exact together (the mark between them is on a conditional path)" % (
        "rows", ra["valu"] + rb["valu"], ra["valu_inner_loops"] + rb["valu_inner_loops"], ra["salu"] + rb["salu"],
        ra["salu_inner_loops"] + rb["salu_inner_loops"], ra["smem"] + rb["smem"], ra["lds"] + rb["lds"], ra["lds_inner_loops"] + rb["lds_inner_loops"],
        ra["vmem"] + rb["vmem"]))
    per_exp = [r for r in rows if r["phase"] in PHASES[1:7]]
    print("per expansion, static once-through: VALU %d (+%d in inner loops), SALU %d (+%d), LDS %d (+%d), VMEM %d" % (
        sum(r["valu"] for r in per_exp), sum(r["valu_inner_loops"] for r in per_exp), sum(r["salu"] for r in per_exp),
        sum(r["salu_inner_loops"] for r in per_exp), sum(r["lds"] for r in per_exp), sum(r["lds_inner_loops"] for r in per_exp), sum(r["vmem"] for r in per_exp)))


if __name__ == "__main__":
    main()
