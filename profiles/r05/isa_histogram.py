"""Dynamic opcode histogram of cells_kernel: the compiler's assembly (line tables on: every instruction carries its source
line and the chain of calls it was inlined through) x the trip census of the run (profiles/r05/census_run.py, a
-DTRX_CENSUS build).  VERDICT round 4, item 6: "of cells_kernel's VALU issue 58 % is fp64 arithmetic, 10 % INT32, and 32 %
are moves / selects / compares / conversions nobody has itemised".

    sh profiles/r05/isa_histogram.sh        (builds, runs the census on the GPU box, writes profiles/r05/isa_histogram_*.txt)

An instruction is filed under (stage, function): stage = where in cells_body (trx_kernels.hip) the outermost inlined frame
sits -- window pass / cell plans / pair table / pair loop / cell finalisation / batch set-up / kernel set-up -- and function
= the innermost frame (ma_flux, cel_pair's loop, kepler_step, ...).  A wave executes a basic block when ANY lane needs it,
and the 64 pairs of a trip are a mix of cases, so within a stage every instruction is weighted with the stage's trip count
-- except the places the census counts separately: the AGM loop of cel_pair, the two geometric cases of ma_flux, the
full Kepler solve of a plan."""
import collections
import json
import re
import sys


def classify(op):
    if op.startswith("v_"):
        if op.startswith("v_fma_f64") or op.startswith("v_fmac_f64"):
            return "fp64 fma"
        if op.startswith("v_mul_f64"):
            return "fp64 mul"
        if op.startswith("v_add_f64"):
            return "fp64 add"
        if re.match(r"v_(rcp|rsq|sqrt)_f64", op):
            return "fp64 rcp/rsq seed"
        if re.match(r"v_cmpx?_\w+_f64", op) or op.startswith("v_cmp_class_f64"):
            return "fp64 compare"
        if re.match(r"v_(min|max|ldexp|frexp_\w+|rndne|trunc|floor|ceil|fract|div_\w+|trig_preop)_f64", op):
            return "fp64 min/max/ldexp/rndne"
        if op.startswith("v_cvt"):
            return "convert"
        if re.search(r"_f32|_f16", op):
            return "fp32"
        if op.startswith("v_cndmask"):
            return "select (v_cndmask)"
        if op.startswith("v_mov") or op.startswith("v_accvgpr") or op.startswith("v_swap"):
            return "move (v_mov)"
        if re.match(r"v_(readlane|writelane)", op):
            return "sgpr spill traffic (v_readlane/v_writelane)"
        if re.match(r"v_(readfirstlane|permlane|perm_|bpermute|mbcnt)", op) or "dpp" in op:
            return "cross-lane (readfirstlane, mbcnt, dpp)"
        if op.startswith("v_cmp"):
            return "int compare"
        if re.match(r"v_(and|or|xor|not|bfe|bfi|bfm|lshl|lshr|ashr|lshlrev|lshrrev|ashrrev|alignbit|alignbyte|and_or|or3|xad|lshl_or|lshl_add)", op):
            return "int logic/shift"
        if re.match(r"v_(add|sub|subrev|mul|mad|addc|subb|min|max|mul_hi|mul_lo|add3|add_lshl|sad)", op):
            return "int arithmetic"
        return "other valu (" + op + ")"
    return None


def stage_of(line, R):
    for name, lo, hi in R:
        if lo <= line <= hi:
            return name
    return "kernel set-up"


def main():
    path, frag, census_path, regions_path = sys.argv[1:5]
    pmc_valu = float(sys.argv[5]) if len(sys.argv) > 5 else None
    cen = json.load(open(census_path))
    reg = json.load(open(regions_path))
    R = [(r["name"], r["lo"], r["hi"]) for r in reg["stages"]]
    fn_ranges = [(f["name"], f["lo"], f["hi"]) for f in reg["functions"]]
    kfile = reg["kernel_file"]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(frag) + r"\S*:", l))
    weights = {
        "window pass": cen["window_trip"], "cell plans": cen["chunk0"] + cen["chunk1"], "pair table": cen["pass"],
        "pair loop": cen["pair_trip"], "cell finalisation": cen["chunk0"] + cen["chunk1"], "batch set-up": cen["batch"],
        "kernel set-up": cen["batch"] / 8.0,
    }
    special = {"cel_pair AGM loop": cen["agm_trip"], "ma_flux (disc inside the limb)": cen["inside_trip"],
               "ma_flux (disc crossing the limb)": cen["crossing_trip"], "ma_flux (common)": cen["flux_trip"],
               "cel_pair (set-up and result)": cen["flux_trip"], "atan_pos_tab": cen["crossing_trip"],
               "kepler_full": None}
    table = collections.defaultdict(collections.Counter)        # (stage, fn) -> class -> static count
    other = collections.Counter()
    cur = (kfile, 0, [])
    i = start + 1
    while ".end_amdhsa_kernel" not in lines[i] and not lines[i].startswith(".Lfunc_end"):
        l = lines[i]
        t = l.strip()
        m = re.match(r"\.loc\s+\d+\s+(\d+)\s+\d+.*?;\s*(\S+?):(\d+):\d+(.*)$", t)
        if m:
            chain = [(m.group(2), int(m.group(3)))] + [(a, int(b)) for a, b in re.findall(r"@\[\s*(\S+?):(\d+):\d+", m.group(4))]
            cur = chain
        elif t and not t.startswith(";") and not t.startswith(".") and not re.match(r"^\S+:$", t):
            op = t.split()[0]
            c = classify(op)
            chain = cur if isinstance(cur, list) else []
            # stage: the outermost frame inside cells_body
            stage = "kernel set-up"
            for f, ln in reversed(chain):
                if f.endswith(kfile):
                    s_ = stage_of(ln, R)
                    if s_ != "kernel set-up":
                        stage = s_
                        break
            # function: among the frames below the stage, the innermost one the census has a count of its own for
            # (kepler_full, the AGM loop, the two cases of ma_flux, its arctangents); else the outermost named function
            # of trx_device.hpp; the small helpers (reciprocal, square root, fma_k, range reduction) count for their caller
            fn = "(stage's own lines)"
            named = []
            for f, ln in chain:
                if f.endswith(kfile):
                    break
                hit = next((name for name, lo, hi in fn_ranges if lo <= ln <= hi), None)
                named.append(hit)
            if named:
                spec = [h for h in named if h in special]
                fn = spec[0] if spec else next((h for h in reversed(named) if h), "device math (helpers called from the stage)")
            if c is None:
                other[(stage, "salu" if op.startswith("s_") and not re.match(r"s_(waitcnt|nop|cbranch|branch|load|buffer)", op)
                       else ("lds" if op.startswith("ds_") else ("vmem" if re.match(r"(global|flat|buffer|scratch)_", op) else "other")))] += 1
            else:
                table[(stage, fn)][c] += 1
        i += 1
    # dynamic counts
    dyn = collections.Counter()
    dyn_stage = collections.defaultdict(collections.Counter)
    rows = []
    for (stage, fn), cnts in table.items():
        w = weights[stage]
        if fn in special and special[fn] is not None and stage == "pair loop":
            w = special[fn]
        if fn == "kepler_full":
            w = cen["kepler_full_plan"] * 3.0 if stage == "cell plans" else cen["kepler_full_pair"] * 3.0      # ~3 iterations
        n_static = sum(cnts.values())
        rows.append((stage, fn, n_static, w, cnts))
        for c, v in cnts.items():
            dyn[c] += v * w
            dyn_stage[stage][c] += v * w
    total = sum(dyn.values())
    print("kernel %s" % lines[start].split(":")[0])
    print("census (per launch, %d rows x %d points, %s grid): %s" % (cen["rows"], cen["n_time"], cen["grid"],
          ", ".join("%s %.3g" % (k, cen[k]) for k in ("batch", "window_trip", "chunk0", "chunk1", "pass", "pair_trip", "pair_lanes",
                                                      "flux_trip", "flux_lanes", "agm_trip", "inside_trip", "crossing_trip",
                                                      "kepler_full_plan"))))
    print("lanes per pair trip %.1f of 64; AGM loop trips (two Bulirsch steps each) per flux trip %.2f"
          % (cen["pair_lanes"] / cen["pair_trip"], cen["agm_trip"] / cen["flux_trip"]))
    print("\nVALU wave-instructions per launch, model: %.3e%s" % (total, ("   measured (SQ_INSTS_VALU): %.3e   model / measured %.3f"
          % (pmc_valu, total / pmc_valu)) if pmc_valu else ""))
    print("\n%-46s %12s %7s" % ("class", "per launch", "share"))
    for c, v in dyn.most_common():
        print("%-46s %12.3e %6.1f %%" % (c, v, 100.0 * v / total))
    f64 = sum(v for c, v in dyn.items() if c in ("fp64 fma", "fp64 mul", "fp64 add"))
    print("%-46s %12.3e %6.1f %%" % ("= fp64 fma + mul + add", f64, 100.0 * f64 / total))
    print("\nby stage:")
    for stage, cnts in sorted(dyn_stage.items(), key=lambda kv: -sum(kv[1].values())):
        st = sum(cnts.values())
        f = sum(v for c, v in cnts.items() if c in ("fp64 fma", "fp64 mul", "fp64 add"))
        print("  %-20s %10.3e (%4.1f %% of VALU)  fp64 arithmetic %4.1f %%; largest other classes: %s"
              % (stage, st, 100.0 * st / total, 100.0 * f / st,
                 ", ".join("%s %.1f %%" % (c, 100.0 * v / st) for c, v in cnts.most_common(8) if c not in ("fp64 fma", "fp64 mul", "fp64 add"))))
    print("\nby (stage, function): static VALU instructions x trips")
    for stage, fn, n_static, w, cnts in sorted(rows, key=lambda r: -r[2] * r[3])[:28]:
        d = n_static * w
        top = ", ".join("%s %d" % (c, v) for c, v in cnts.most_common(6))
        print("  %-18s %-34s %5d x %9.3e = %9.3e (%4.1f %%)  [%s]" % (stage, fn, n_static, w, d, 100.0 * d / total, top))
    print("\nnon-VALU instructions (static, by stage): %s" % dict(other))


main()
