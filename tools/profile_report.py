#!/usr/bin/env python3
"""Markdown summary of a tools/profile.sh directory (round 4: + where the wave cycles go and what the LDS does) for workloads whose kernels are launched at several sizes (the `others` set:
BC1 / BC3 / BC5 encode, BC6H cfg3, the decoders, Convert, the cfg4 mip chains and BC3 of the chain). Unlike tools/profile_report_r02.py
(one row per kernel NAME, right for the BC7 pipeline where every launch of a name does the same work) rows here are (kernel, grid size):
the 4096^2 / 8192^2 instance of a kernel is not averaged with the 1 x 1 level of a mip chain.

usage: tools/profile_report.py gpurun_out/<dir> "<title>" profiles/<name>.md [--by-name] [--algo-bytes N] [--json profiles/pmc_traffic.json --dominant <mark name>]
--by-name merges the launches of a kernel name (the BC7 pipeline, where every launch of a name does the same work); --algo-bytes N = the
algorithmic bytes of one image for every kernel of such a pipeline (BC7 cfg2: 83886080); --json writes the dominant kernel's traffic and SQ
counters in the form bench.py falls back to when it cannot run rocprofv3 itself.

VALU utilisation. gfx950 issues a wave64 VALU instruction over 2 cycles for a handful of opcodes (v_add/sub_u32, v_and/or/xor_b32,
v_lshrrev/ashrrev_b32, v_mov_b32, v_add/mul/fma_f32 ...) and over 4 cycles for everything else the search kernels use (v_dot4, v_cmp,
v_cndmask, v_max*, v_mad*, v_mul*, every VOP3-only opcode; transcendental 8) - profiles/r02_valu_rates.md. The fraction reported is
    (VALU instructions per SIMD) x (mean issue cycles per instruction) / (kernel duration x 2.4 GHz)
with the mean issue cost taken from the kernel's own code: its disassembly (hipcc -S) is classified opcode by opcode with the measured
table, instructions weighted 8^(loop depth) so that the inner loops - where nearly all dynamic instructions are - dominate. It is an
estimate of the dynamic mix, bounded by construction between the all-2-cycle and all-4-cycle readings, which are printed next to it.
NOTE on kernels that run side by side (the late modes 4 / 5 of the BC7 pipeline on three streams): the durations come from the plain
kernel trace, where they overlap and share the SIMDs, so each one's utilisation is its SHARE of the issue slots during its span, not
what it reaches alone."""
import collections, csv, os, re, subprocess, sys, glob

D, title, out_md = sys.argv[1], sys.argv[2], sys.argv[3]
BY_NAME = "--by-name" in sys.argv


def opt(name, default=None):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


ALGO_ALL = float(opt("--algo-bytes", 0) or 0)
JSON_OUT, DOMINANT = opt("--json"), opt("--dominant")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS, HZ = 1024, 2.4e9
FAST = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32", "v_add_f32", "v_sub_f32",
        "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mac_f32", "v_not_b32", "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_accvgpr")
SLOW8 = ("v_rcp_", "v_sqrt_", "v_rsq_", "v_exp_", "v_log_", "v_sin_", "v_cos_")
MiB = 1048576.0
# algorithmic bytes (bytes that must be read + written) of the full-size instance of each kernel: (kernel substring, grid) -> (bytes, what)
ALGO = [
    ("bc15_encode_kernel<1, false, true>", 1048576, 72 * MiB, "BC1 4096^2: 64 MiB RGBA8 in, 8 MiB out"),
    ("bc15_encode_kernel<3, false, true>", 1048576, 80 * MiB, "BC3 4096^2"),
    ("bc15_encode_kernel<3, false, true>", 4194304, 320 * MiB, "BC3 8192^2 (cfg4 level 0)"),
    ("bc15_encode_kernel<5, false, true>", 1048576, 80 * MiB, "BC5 4096^2"),
    ("bc_decode_kernel<0>", 1048576, 76 * MiB, "BC1 (72 MiB) and BC3 (80 MiB) 4096^2 -> RGBA8, alternating"),
    ("bc_decode_kernel<2>", 1048576, 144 * MiB, "BC6H 4096^2 -> RGBA16F"),
    ("bc_decode_kernel<3>", 1048576, 80 * MiB, "BC7 4096^2 (arbitrary blocks) -> RGBA8"),
    ("convert_quad_kernel", 1048576, 192 * MiB, "RGBA8 -> RGBA16F 4096^2"),
    ("resize_box_half_rgba8_kernel", 8388608, 320 * MiB, "8192^2 -> 4096^2"),
    ("resize_cubic_half_rgba8_kernel", 2097152, 320 * MiB, "8192^2 -> 4096^2"),
]


def short(k):
    return k.replace('dxtex::(anonymous namespace)::', '').replace('void ', '').split('(')[0]


def cost_of(op):
    if any(op.startswith(p) for p in SLOW8):
        return 8.1
    base = op.replace("_e32", "").replace("_e64", "").replace("_sdwa", "").replace("_dpp", "")
    if op.endswith("_e64") or op.endswith("_sdwa") or op.endswith("_dpp"):
        return 4.15
    return 2.15 if base in FAST else 4.15


def static_mix():
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
                dem = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
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
                wgt = 8.0 ** depth; c = cost_of(t[0])
                acc[0] += wgt * c; acc[1] += wgt; acc[2] += wgt * (1.0 if c < 3 else 0.0)
    return {k: (v[0] / v[1], v[2] / v[1]) for k, v in out.items() if v[1] > 0}


# durations per (kernel, grid) from the plain trace
dur = collections.defaultdict(list); meta = {}
for r in csv.DictReader(open(os.path.join(D, 'trace_kernel_trace.csv'))):
    g = 0 if BY_NAME else int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
    k = (short(r['Kernel_Name']), g)
    dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
    meta[k] = (r['VGPR_Count'], r['SGPR_Count'], r['LDS_Block_Size'], r['Scratch_Size'], r['Workgroup_Size_X'])
cnt = collections.defaultdict(lambda: collections.defaultdict(float)); nl = collections.defaultdict(lambda: collections.defaultdict(int))
for f in ('pass1', 'pass2', 'pass3', 'fetch', 'write'):
    p = os.path.join(D, f + '_counter_collection.csv')
    if os.path.exists(p):
        for r in csv.DictReader(open(p)):
            k = (short(r['Kernel_Name']), 0 if BY_NAME else int(r['Grid_Size']))
            cnt[k][r['Counter_Name']] += float(r['Counter_Value']); nl[k][r['Counter_Name']] += 1
mix = static_mix()
total = sum(sum(v) for v in dur.values())
rows = sorted(dur.items(), key=lambda kv: -sum(kv[1]))

L = [f"# {title}\n"]
L.append("Collected by `tools/profile.sh <tag> <workload>` (`tools/prof_workloads.py <workload> 3`: one `rocprofv3 --kernel-trace --stats` run, then separate `--kernel-trace --pmc` passes with one")
L.append("repetition: SQ instruction counters, SQ wait counters + GRBM_GUI_ACTIVE, the LDS pass, TCC FETCH_SIZE, TCC WRITE_SIZE), summarised by `tools/profile_report.py`. Rows are (kernel, grid size):")
L.append("a kernel launched for several image sizes (mip levels, the BC3 of a chain) has a row per size. The BC6H kernels are launched once per mode (10 launches per image).\n")
L.append("## Kernel trace\n")
L.append("(rocprofv3's VGPR column is read with the wave32 granule: a wave64 lane holds twice that, see tools/kernel_resources.sh.)\n")
L.append("| kernel | grid (threads) | calls | avg ms | min ms | total ms | % | VGPR | SGPR | LDS B | scratch B | wg |")
L.append("|---|---|---|---|---|---|---|---|---|---|---|---|")
for (k, g), v in rows:
    if sum(v) / total < 0.0005 and max(v) < 0.02:
        continue
    m = meta[(k, g)]
    L.append(f"| `{k}` | {g} | {len(v)} | {sum(v)/len(v):.4f} | {min(v):.4f} | {sum(v):.3f} | {100*sum(v)/total:.2f} | " + " | ".join(m) + " |")
L.append("\n## VALU issue utilisation (method: header of tools/profile_report.py, issue costs of profiles/r02_valu_rates.md)\n")
L.append("Round 6: the last column is MEASURED - `SQ_ACTIVE_INST_VALU` (quad-cycles in which a wave had a VALU instruction executing, summed over the waves) x 4 /")
L.append("(1024 SIMDs x kernel cycles at 2.4 GHz): the VALU pipes' busy share, independent of any opcode price list. The static-mix estimate left of it weighs the")
L.append("kernel's disassembly and under-prices kernels whose hot loops are `v_dot4` / `v_max` (4 cycles) next to much cold 2-cycle straight-line code.\n")
L.append("| kernel | grid | avg ms | waves | VALU insts / wave | active lanes per VALU inst | mean issue cyc / inst (static mix) | VALU issue utilisation | if every inst were 2 cyc / 4 cyc | VALU busy, measured |")
L.append("|---|---|---|---|---|---|---|---|---|---|")
for (k, g), v in rows:
    c = cnt.get((k, g))
    if not c or not c.get('SQ_INSTS_VALU'):
        continue
    n = max(1, nl[(k, g)]['SQ_WAVES'])
    insts = c['SQ_INSTS_VALU'] / n
    if insts < 2e5:
        continue
    secs = sum(v) / len(v) * 1e-3
    lanes = c['SQ_THREAD_CYCLES_VALU'] / max(1, c['SQ_ACTIVE_INST_VALU'])
    cyc = mix.get(k, (4.15, 0))[0]
    ps = insts / SIMDS
    busy = (c['SQ_ACTIVE_INST_VALU'] / n) * 4.0 / (SIMDS * secs * HZ)
    L.append("| `%s` | %d | %.4f | %d | %.0f | %.1f | %.2f | **%.2f** | %.2f / %.2f | **%.2f** |" % (
        k, g, secs * 1e3, c['SQ_WAVES'] / n, insts / max(1, c['SQ_WAVES'] / n), lanes, cyc, min(1.0, ps * cyc / (secs * HZ)), ps * 2.15 / (secs * HZ), ps * 4.15 / (secs * HZ), busy))
# ---- where the wave cycles go (pass 2) and what the LDS does (pass 3) -------------------------------------------------------------
L.append("\n## Where the wave cycles go, and the LDS (SQ wait / activity counters, quad-cycles; `tools/profile.sh` passes 2 and 3)\n")
L.append("`SQ_WAIT_ANY` = wave parked on `s_waitcnt` / a barrier (memory or LDS latency not hidden); `SQ_WAIT_INST_ANY` = a ready instruction could not issue")
L.append("(the pipe it needs is taken - at high occupancy mostly by OTHER waves of the SIMD, i.e. the SIMD is saturated - or a dependency stall); `SQ_ACTIVE_INST_ANY` = issuing.")
L.append("The three are disjoint and add up to about `SQ_WAVE_CYCLES` (MI355X_MICROARCH.md, PMC section). LDS: `SQ_LDS_BANK_CONFLICT` = extra cycles lost to bank conflicts out of")
L.append("`SQ_LDS_IDX_ACTIVE` = all cycles the LDS arrays were busy; `SQ_WAIT_INST_LDS` = issue stalls on the LDS pipe (a part of WAIT_INST_ANY).\n")
L.append("| kernel | grid | avg ms | issue util. | parked (WAIT_ANY) | issue-stalled (WAIT_INST_ANY) | issuing (ACTIVE_INST_ANY) | of which VALU / LDS / VMEM / scalar | LDS insts per wave | LDS bank-conflict share of LDS cycles | LDS busy share of kernel time | LDS issue stall share of wave cycles | VMEM rd / wr insts per wave | reading |")
L.append("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for (k, g), v in rows:
    c = cnt.get((k, g))
    if not c or not c.get('SQ_WAIT_ANY') or not c.get('SQ_INSTS_VALU'):
        continue
    n1 = max(1, nl[(k, g)]['SQ_WAVES']); n2 = max(1, nl[(k, g)]['SQ_WAIT_ANY']); n3 = max(1, nl[(k, g)].get('SQ_LDS_IDX_ACTIVE', 1))
    insts = c['SQ_INSTS_VALU'] / n1
    if insts < 2e5:
        continue
    secs = sum(v) / len(v) * 1e-3
    waves = c['SQ_WAVES'] / n1
    wc = c.get('SQ_WAVE_CYCLES', 0.0)
    # SQ_WAVE_CYCLES is collected in pass 1 and pass 2: average the two
    wcn = max(1, nl[(k, g)].get('SQ_WAVE_CYCLES', 1)); wave_cyc = wc / wcn
    wa, wi, ac = c['SQ_WAIT_ANY'] / n2, c['SQ_WAIT_INST_ANY'] / n2, c['SQ_ACTIVE_INST_ANY'] / n2
    tot3 = max(1.0, wa + wi + ac)
    a_valu = c.get('SQ_ACTIVE_INST_VALU', 0.0) / n1
    a_lds = c.get('SQ_ACTIVE_INST_LDS', 0.0) / n3; a_vmem = c.get('SQ_ACTIVE_INST_VMEM', 0.0) / n3; a_sca = c.get('SQ_ACTIVE_INST_SCA', 0.0) / n3
    lds_i = c.get('SQ_INSTS_LDS', 0.0) / n2 / max(1, waves)
    bank = c.get('SQ_LDS_BANK_CONFLICT', 0.0) / n3; idx = c.get('SQ_LDS_IDX_ACTIVE', 0.0) / n3
    w_lds = c.get('SQ_WAIT_INST_LDS', 0.0) / n3
    gui = c.get('GRBM_GUI_ACTIVE', 0.0) / max(1, nl[(k, g)].get('GRBM_GUI_ACTIVE', 1))
    cyc = mix.get(k, (4.15, 0))[0]
    util = min(1.0, insts / SIMDS * cyc / (secs * HZ))
    # LDS busy share: IDX_ACTIVE is summed over the CUs' LDS arrays (256), in cycles
    lds_busy = idx / 256.0 / max(1.0, gui) if gui else 0.0
    park, stall, issue = wa / tot3, wi / tot3, ac / tot3
    if util >= 0.9:
        reading = "VALU issue-bound: the SIMDs issue nearly every slot"
    elif park >= 0.5:
        reading = "latency-bound: waves sit in s_waitcnt / barriers more than half of their time" + (" (LDS round trips)" if lds_i > 200 and bank / max(1.0, idx) < 0.1 else "")
    elif stall >= 0.5 and util >= 0.6:
        reading = "issue-bound per SIMD (ready waves queue for the VALU) but lanes idle inside the instructions - divergence, see active lanes"
    elif stall >= 0.4:
        reading = "waves ready but not issuing: dependency / pipe stalls at low occupancy"
    else:
        reading = "mixed: neither parked nor stalled dominates"
    if idx and bank / idx >= 0.15:
        reading += "; LDS bank conflicts cost %.0f %% of its cycles" % (100 * bank / idx)
    L.append("| `%s` | %d | %.4f | %.2f | %.2f | %.2f | %.2f | %.2f / %.2f / %.2f / %.2f | %.0f | %s | %s | %.3f | %.0f / %.0f | %s |" % (
        k, g, secs * 1e3, util, park, stall, issue,
        a_valu / max(1.0, ac), a_lds / max(1.0, ac), a_vmem / max(1.0, ac), a_sca / max(1.0, ac), lds_i,
        ("%.3f" % (bank / idx)) if idx else "-", ("%.2f" % lds_busy) if idx and gui else "-", w_lds / max(1.0, wave_cyc),
        c.get('SQ_INSTS_VMEM_RD', 0.0) / n2 / max(1, waves), c.get('SQ_INSTS_VMEM_WR', 0.0) / n2 / max(1, waves), reading))
L.append("\n## HBM traffic per launch (TCC `FETCH_SIZE`, `WRITE_SIZE`, separate passes; rocprofv3 reports KiB)\n")
L.append("gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts the 128-B requests of wide (16 B per lane) streaming reads as 64 B; kernels marked x2 read their")
L.append("image with 16-byte loads and have FETCH_SIZE doubled. Algorithmic bytes (SURVEY 8d) are listed for the full-size instance of each kernel.\n")
L.append("| kernel | grid | avg ms | FETCH MiB | WRITE MiB | traffic MiB | traffic GB/s | algorithmic MiB | algorithmic GB/s | fraction of 8 TB/s | traffic / algorithmic | what |")
L.append("|---|---|---|---|---|---|---|---|---|---|---|---|")
WIDE = ("bc15_encode_kernel", "bc_decode", "resize_box_half")
TRAFFIC = {}; STEP_TRAFFIC = [0.0]
for (k, g), v in rows:
    c = cnt.get((k, g))
    if not c or 'FETCH_SIZE' not in c:
        continue
    f = c['FETCH_SIZE'] / max(1, nl[(k, g)]['FETCH_SIZE']) / 1024.0; w = c.get('WRITE_SIZE', 0.0) / max(1, nl[(k, g)].get('WRITE_SIZE', 1)) / 1024.0
    x2 = any(t in k for t in WIDE)
    if x2:
        f *= 2
    tot = f + w
    secs = sum(v) / len(v) * 1e-3
    a = next(((b, what) for s, gg, b, what in ALGO if s in k and gg == g), None)
    if not a and ALGO_ALL:
        a = (ALGO_ALL, "")
    TRAFFIC[k] = tot * MiB; STEP_TRAFFIC[0] += tot * MiB * len(v)
    if tot < 4.0 and (not a or ALGO_ALL):
        continue
    L.append("| `%s`%s | %d | %.4f | %.1f | %.1f | %.1f | %.0f | %s | %s | %s | %s | %s |" % (
        k, " (x2)" if x2 else "", g, secs * 1e3, f, w, tot, tot * MiB / secs / 1e9,
        "%.0f" % (a[0] / MiB) if a else "-", "%.0f" % (a[0] / secs / 1e9) if a else "-", "%.3f" % (a[0] / secs / 8e12) if a else "-",
        "%.2f" % (tot * MiB / a[0]) if a else "-", a[1] if a else ""))
if ALGO_ALL:
    reps = max(len(v) for v in dur.values() if v)
    nimg = max(1, min(len(v) for (k, g), v in rows[:5]))        # the big kernels run once per image
    L.append("\nAll launches of one image: %.1f MiB of HBM traffic = %.1f x the algorithmic bytes." % (STEP_TRAFFIC[0] / nimg / MiB, STEP_TRAFFIC[0] / nimg / ALGO_ALL))
open(out_md, 'w').write("\n".join(L) + "\n")
if JSON_OUT:
    import hashlib, json
    (dom, g), v = rows[0]
    c = cnt[(dom, g)]; n = max(1, nl[(dom, g)]['SQ_WAVES'])
    insts = c.get('SQ_INSTS_VALU', 0.0) / n; secs = sum(v) / len(v) * 1e-3
    cyc = mix.get(dom, (4.15, 0))[0]
    h = hashlib.sha256()
    for name in ("bc7_encode.hip", "bc7_core.h", "search_common.h"):       # the stamp bench.py checks (kernel_sources_sha256)
        h.update(open(os.path.join(ROOT, "directxtex_amd", "csrc", name), "rb").read())
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    lanes = c.get('SQ_THREAD_CYCLES_VALU', 0.0) / max(1.0, c.get('SQ_ACTIVE_INST_VALU', 0.0))
    n3 = max(1, nl[(dom, g)].get('SQ_LDS_IDX_ACTIVE', 1))
    json.dump({"kernel": DOMINANT or dom, "rocprof_kernel": dom, "hbm_bytes_per_launch": int(TRAFFIC.get(dom, 0)),
               "sources_sha256": h.hexdigest(), "git_head_at_report": head,
               "valu": {"issue_utilisation": round(min(1.0, insts / SIMDS * cyc / (secs * HZ)), 3), "active_lanes_per_valu_inst": round(lanes, 1),
                        "valu_insts_per_launch": int(insts), "kernel_avg_ms_profiled": round(secs * 1e3, 3), "mean_issue_cycles_per_inst": round(cyc, 2),
                        "lds_bank_conflict_share": round(c.get('SQ_LDS_BANK_CONFLICT', 0.0) / max(1.0, c.get('SQ_LDS_IDX_ACTIVE', 0.0)), 3),
                        "method": "SQ_INSTS_VALU per SIMD x loop-weighted static issue cost (profiles/r02_valu_rates.md) / (duration x 2.4 GHz)"},
               "traffic_whole_step_over_algorithmic": round(STEP_TRAFFIC[0] / max(1, min(len(v) for (k, g), v in rows[:5])) / ALGO_ALL, 1) if ALGO_ALL else None,
               "source": f"{out_md}: FETCH_SIZE + WRITE_SIZE, separate --pmc passes, KiB -> bytes"}, open(JSON_OUT, 'w'), indent=1)
print("\n".join(L)[:6000])
