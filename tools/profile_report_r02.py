#!/usr/bin/env python3
"""Turns the CSVs written by tools/profile_r02.sh into a markdown summary for profiles/.

usage: tools/profile_report_r02.py gpurun_out/<dir> "<title>" profiles/<name>.md [--algo kernel_substring=bytes ...] [--json profiles/pmc_traffic.json --dominant <mark name>]

VALU utilisation. gfx950 issues a wave64 VALU instruction over 2 cycles for a handful of opcodes (v_add/sub_u32, v_and/or/xor_b32,
v_lshrrev/ashrrev_b32, v_mov_b32, v_add/mul/fma_f32 ...) and over 4 cycles for everything else the search kernels use (v_dot4, v_cmp,
v_cndmask, v_max*, v_mad*, v_mul*, every VOP3-only opcode; transcendental 8) - profiles/r02_valu_rates.md. The fraction reported is
    (VALU instructions per SIMD) x (mean issue cycles per instruction) / (kernel duration x effective clock)
with the mean issue cost taken from the kernel's own code: its disassembly (hipcc -S) is classified opcode by opcode with the measured
table, instructions weighted 8^(loop depth) so that the inner loops - where nearly all dynamic instructions are - dominate. It is an
estimate of the dynamic mix, bounded by construction between the all-2-cycle and all-4-cycle readings, which are printed next to it.
The effective clock is GRBM_GUI_ACTIVE / duration where that counter was collected, else 2.4 GHz."""
import collections, csv, glob, json, os, re, subprocess, sys

D, title, out_md = sys.argv[1], sys.argv[2], sys.argv[3]
args = sys.argv[4:]
algo = {}
json_out = None; dominant_mark = None
i = 0
while i < len(args):
    if args[i] == "--algo":
        k, v = args[i + 1].split("="); algo[k] = float(v); i += 2
    elif args[i] == "--json":
        json_out = args[i + 1]; i += 2
    elif args[i] == "--dominant":
        dominant_mark = args[i + 1]; i += 2
    else:
        i += 1
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS, NOMINAL_HZ = 1024, 2.4e9

FAST = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32", "v_add_f32", "v_sub_f32",
        "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mac_f32", "v_not_b32", "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_accvgpr")
SLOW8 = ("v_rcp_", "v_sqrt_", "v_rsq_", "v_exp_", "v_log_", "v_sin_", "v_cos_")


def short(k):
    return k.replace('dxtex::(anonymous namespace)::', '').replace('void ', '').split('(')[0]


def cost_of(op):
    if any(op.startswith(p) for p in SLOW8):
        return 8.1
    base = op.replace("_e32", "").replace("_e64", "").replace("_sdwa", "").replace("_dpp", "")
    if op.endswith("_e64") or op.endswith("_sdwa") or op.endswith("_dpp"):
        return 4.15                                           # VOP3 / SDWA / DPP encodings issue at the slow rate
    return 2.15 if base in FAST else 4.15


def static_mix():
    """kernel short name -> (mean issue cycles per VALU instruction, weighted fraction of fast opcodes)"""
    out = {}
    for src in glob.glob(os.path.join(ROOT, "directxtex_amd", "csrc", "*.hip")):
        s_path = "/tmp/_mix_" + os.path.basename(src) + ".s"
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                            "-I" + os.path.join(ROOT, "directxtex_amd", "csrc"), "-I" + os.path.join(ROOT, "include"), "-x", "hip", "--cuda-device-only", "-S", src, "-o", s_path],
                           capture_output=True, text=True)
        if r.returncode != 0:
            continue
        name = None; depth = 0; acc = None
        for line in open(s_path):
            m = re.match(r"^(_Z\w+):", line)
            if m:
                try:
                    dem = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
                except Exception:
                    dem = m.group(1)
                name = short(dem); acc = [0.0, 0.0, 0.0]; depth = 0; out[name] = acc
                continue
            if name is None:
                continue
            if ".amdhsa_kernel" in line:
                name = None; continue
            dm = re.search(r"Depth[ =](\d+)", line)
            if line.startswith(".LBB"):
                depth = int(dm.group(1)) if dm else 0
                if "Loop Header" not in line and "in Loop" not in line:
                    depth = 0
                continue
            t = line.strip().split()
            if t and t[0].startswith("v_") and not t[0].startswith("v_mfma"):
                wgt = 8.0 ** depth
                c = cost_of(t[0])
                acc[0] += wgt * c; acc[1] += wgt; acc[2] += wgt * (1.0 if c < 3 else 0.0)
    return {k: (v[0] / v[1], v[2] / v[1]) for k, v in out.items() if v[1] > 0}


stats = list(csv.DictReader(open(os.path.join(D, 'trace_kernel_stats.csv'))))
meta = {}
for r in csv.DictReader(open(os.path.join(D, 'pass1_counter_collection.csv'))):
    meta.setdefault(short(r['Kernel_Name']), (r['VGPR_Count'], r['Accum_VGPR_Count'], r['SGPR_Count'], r['LDS_Block_Size'], r['Scratch_Size'], r['Grid_Size'], r['Workgroup_Size']))
cnt = collections.defaultdict(lambda: collections.defaultdict(float)); launches = collections.defaultdict(lambda: collections.defaultdict(int))
for f in ('pass1', 'pass2', 'fetch', 'write'):
    p = os.path.join(D, f + '_counter_collection.csv')
    if not os.path.exists(p):
        continue
    for r in csv.DictReader(open(p)):
        k = short(r['Kernel_Name']); cnt[k][r['Counter_Name']] += float(r['Counter_Value']); launches[k][r['Counter_Name']] += 1
mix = static_mix()
avg_ms = {short(r['Name']): float(r['AverageNs']) / 1e6 for r in stats}

L = [f"# {title}\n"]
L.append("Collected by `tools/profile_r02.sh` (one `rocprofv3 --kernel-trace --stats` run, then separate `--kernel-trace --pmc` passes: SQ instruction counters,")
L.append("SQ wait counters + GRBM_GUI_ACTIVE, TCC FETCH_SIZE, TCC WRITE_SIZE); summarised by `tools/profile_report_r02.py`.\n")
L.append("## Kernel trace (`--stats`)\n")
L.append("(rocprofv3's VGPR column is the kernel descriptor's granulated count read with the wave32 granule: the registers a wave64 lane really holds are twice that, as `hipcc -Rpass-analysis=kernel-resource-usage` / tools/kernel_resources.sh report.)\n")
L.append("| kernel | calls | total ms | avg ms | % | VGPR | AGPR | SGPR | LDS B | scratch B | grid | wg |")
L.append("|---|---|---|---|---|---|---|---|---|---|---|---|")
for r in stats:
    k = short(r['Name']); m = meta.get(k, ('?',) * 7)
    if float(r['Percentage']) < 0.02:
        continue
    L.append(f"| `{k}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e6:.4f} | {float(r['Percentage']):.2f} | " + " | ".join(m) + " |")
L.append("\n## VALU issue utilisation per launch (see the header of tools/profile_report_r02.py and profiles/r02_valu_rates.md)\n")
L.append("The static mix is an estimate (a few per cent either way; the clock is taken at its 2.4 GHz maximum, profiled passes run lower): readings above 1 are shown as 1.00 = the SIMDs' VALU")
L.append("issue slots are full. The two bracketing columns do not depend on the mix: a kernel whose all-4-cycle reading is below 1 has idle issue slots for certain.\n")
L.append("| kernel | avg ms | waves | VALU insts / wave | active lanes per VALU inst | G lane-ops/s | fast-opcode share (static, loop-weighted) | mean issue cyc / inst | VALU issue utilisation | if every inst were 2 cyc / 4 cyc | clock GHz |")
L.append("|---|---|---|---|---|---|---|---|---|---|---|")
order = [short(r['Name']) for r in stats]
util = {}
for k in order:
    v = cnt.get(k)
    if not v or not v.get('SQ_WAVE_CYCLES') or k not in avg_ms:
        continue
    n = max(1, launches[k]['SQ_WAVES'])
    insts = v['SQ_INSTS_VALU'] / n
    if insts < 1e5:
        continue
    secs = avg_ms[k] * 1e-3
    gui = v.get('GRBM_GUI_ACTIVE', 0.0) / max(1, launches[k].get('GRBM_GUI_ACTIVE', 1))
    clock = NOMINAL_HZ
    lanes = v['SQ_THREAD_CYCLES_VALU'] / max(1, v['SQ_ACTIVE_INST_VALU'])
    cyc, fast = mix.get(k, (4.15, 0.0))
    per_simd = insts / SIMDS
    u = per_simd * cyc / (secs * clock); u2 = per_simd * 2.15 / (secs * clock); u4 = per_simd * 4.15 / (secs * clock)
    util[k] = (u, lanes, insts, secs)
    L.append("| `%s` | %.4f | %d | %.0f | %.1f | %.0f | %.2f | %.2f | **%.2f** | %.2f / %.2f | %.2f |" % (
        k, avg_ms[k], v['SQ_WAVES'] / n, insts / max(1, v['SQ_WAVES'] / n), lanes, insts * lanes / secs / 1e9, fast, cyc, min(u, 1.0), u2, u4, clock / 1e9))
L.append("\n## HBM traffic per launch (TCC `FETCH_SIZE`, `WRITE_SIZE`, separate passes; rocprofv3 reports KiB)\n")
L.append("gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts the 128-B requests of wide (16 B per lane) streaming reads as 64 B. Kernels marked x2")
L.append("read their image with 16-byte loads and have FETCH_SIZE doubled; the BC6H / BC7 search kernels read 4 B per lane, where the counter matched a known byte count")
L.append("in round 1 (`bc7_rough_kernel` reads the 64 MiB source once: 64.2 MiB reported).\n")
L.append("| kernel | avg ms | FETCH_SIZE MiB | WRITE_SIZE MiB | HBM traffic MiB | traffic GB/s | algorithmic MiB | algorithmic GB/s | fraction of 8 TB/s (algorithmic) | traffic / algorithmic |")
L.append("|---|---|---|---|---|---|---|---|---|---|")
WIDE = ("bc15_encode", "bc_decode", "resize_", "convert_kernel")
traffic = {}
for k in order:
    v = cnt.get(k)
    if not v or 'FETCH_SIZE' not in v or k not in avg_ms:
        continue
    f = v['FETCH_SIZE'] / max(1, launches[k]['FETCH_SIZE']) / 1024.0; w = v.get('WRITE_SIZE', 0.0) / max(1, launches[k].get('WRITE_SIZE', 1)) / 1024.0
    x2 = any(t in k for t in WIDE)
    if x2:
        f *= 2
    tot = f + w
    traffic[k] = tot * 1048576.0
    if tot < 1.0 and avg_ms[k] < 0.02:
        continue
    ab = next((b for s, b in algo.items() if s in k), None)
    secs = avg_ms[k] * 1e-3
    L.append("| `%s`%s | %.4f | %.1f | %.1f | %.1f | %.0f | %s | %s | %s | %s |" % (
        k, " (x2)" if x2 else "", avg_ms[k], f, w, tot, tot * 1048576 / secs / 1e9,
        "%.1f" % (ab / 1048576) if ab else "-", "%.0f" % (ab / secs / 1e9) if ab else "-", "%.4f" % (ab / secs / 8e12) if ab else "-", "%.2f" % (tot * 1048576 / ab) if ab else "-"))
open(out_md, 'w').write("\n".join(L) + "\n")
if json_out:
    dom = order[0]
    u = util.get(dom, (0, 0, 0, 0))
    import hashlib
    h = hashlib.sha256()
    for name in ("bc7_encode.hip", "bc7_core.h", "search_common.h"):       # the stamp bench.py checks (kernel_sources_sha256)
        h.update(open(os.path.join(ROOT, "directxtex_amd", "csrc", name), "rb").read())
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    json.dump({"kernel": dominant_mark or dom, "rocprof_kernel": dom, "hbm_bytes_per_launch": int(traffic.get(dom, 0)),
               "sources_sha256": h.hexdigest(), "git_head_at_report": head,
               "valu": {"issue_utilisation": round(min(u[0], 1.0), 3), "active_lanes_per_valu_inst": round(u[1], 1), "valu_insts_per_launch": int(u[2]),
                        "kernel_avg_ms_profiled": round(u[3] * 1e3, 3), "mean_issue_cycles_per_inst": round(mix.get(dom, (4.15, 0))[0], 2),
                        "lane_ops_per_s": round(u[2] * u[1] / max(u[3], 1e-9), 0),
                        "method": "SQ_INSTS_VALU per SIMD x loop-weighted static issue cost (profiles/r02_valu_rates.md) / (duration x 2.4 GHz)"},
               "source": f"{out_md}: FETCH_SIZE + WRITE_SIZE, separate --pmc passes, KiB -> bytes"}, open(json_out, 'w'), indent=1)
print(open(out_md).read()[:5000])
