#!/usr/bin/env python3
"""Regression guard for the workloads bench.py reports next to the headline: compares every `other_workloads.*.ms` (and the headline's
ms_per_step) of a fresh bench line with the last committed full line (profiles/r*_bench_default.json, highest round) and fails when one got
slower by more than the tolerance. Run by tools/gpu_round_end.sh after the default bench run, so a round that speeds one codec up cannot
silently slow its neighbour down (round 4 did: BC5 +19 %, BC3 +21 %, found only by the next review).

usage: tools/perf_guard.py <fresh bench line .json> [--baseline profiles/rNN_bench_default.json] [--tolerance 0.05] [--floor-ms 0.1]
Times below --floor-ms are compared with an absolute slack of 5 us instead: a 50 us kernel timed over 20 calls moves by that much between
boxes and runs with unchanged code (bc_decode_kernel<2>, same 1 235 instructions per wave: 0.0528 / 0.0480 ms in the kernel traces of
profiles/r05_kernels.md / r06_kernels.md, 0.048 / 0.052 ms in the bench lines of the same two rounds). Exit code 1 = regression."""
import glob, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def opt(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


def load_line(path):
    txt = open(path).read().strip()
    try:
        return json.loads(txt)
    except json.JSONDecodeError:
        for line in reversed(txt.splitlines()):          # a log whose last JSON line is the bench line
            line = line.strip()
            if line.startswith("{"):
                return json.loads(line)
        raise


def times(line):
    out = {"headline.ms_per_step": line.get("ms_per_step")}
    for k, v in (line.get("other_workloads") or {}).items():
        if isinstance(v, dict):
            if isinstance(v.get("ms"), (int, float)):
                out[k + ".ms"] = v["ms"]
            elif isinstance(v.get("Mtexels_s"), (int, float)) and v["Mtexels_s"] > 0:
                out[k + ".us_per_Mtexel"] = round(1e6 / v["Mtexels_s"], 3)       # a rate (the cfg5 shard's image count is a parameter): compared as time per texel
    return {k: v for k, v in out.items() if isinstance(v, (int, float))}


def main():
    fresh_path = sys.argv[1]
    base_path = opt("--baseline", None)
    if not base_path:
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default.json")), key=lambda p: int(re.search(r"r(\d+)_", os.path.basename(p)).group(1)))
        cands = [c for c in cands if os.path.abspath(c) != os.path.abspath(fresh_path)]
        if not cands:
            print("perf_guard: no committed baseline under profiles/"); return 0
        base_path = cands[-1]
    tol = float(opt("--tolerance", "0.05")); floor = float(opt("--floor-ms", "0.1"))
    fresh, base = times(load_line(fresh_path)), times(load_line(base_path))
    bad = []
    print("perf_guard: %s against %s (tolerance %.0f %%)" % (fresh_path, os.path.relpath(base_path, ROOT), tol * 100))
    for k in sorted(base):
        if k not in fresh:
            print("  %-40s missing in the fresh line" % k); continue
        b, f = base[k], fresh[k]
        slow = (f > b + 0.005) if (k.endswith(".ms") and b < floor) else (f > b * (1 + tol))
        print("  %-40s %10.3f -> %10.3f  %+6.1f %%%s" % (k, b, f, 100.0 * (f - b) / b if b else 0.0, "   <-- SLOWER" if slow else ""))
        if slow:
            bad.append(k)
    if bad:
        print("perf_guard: FAILED - slower than the last committed round: " + ", ".join(bad)); return 1
    print("perf_guard: ok"); return 0


if __name__ == "__main__":
    sys.exit(main())
