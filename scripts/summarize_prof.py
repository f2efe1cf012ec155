#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel-trace stats + separate PMC passes) into a short text summary of this
library's kernels (names starting with tce::), with the derived figures DESIGN.md quotes.

rocprofv3 prints FETCH_SIZE / WRITE_SIZE in KiB, and on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide
coalesced streaming read (/opt/skills/guides/MI355X_MICROARCH.md, "HBM"): bytes = value * 1024 * 2.  The same
correction reproduces the algorithmic bytes of the GEMV to 0.5 %; WRITE_SIZE is uncalibrated (shown with the same
factor, for scale only)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
MINE = "tce::"
import re
FILTER = re.compile(os.environ.get("TCE_PROF_FILTER", ""))  # only kernels whose name matches (one summary file per kernel family: profile_r5.sh)


def mine(name):
    return "tce::" in name and bool(FILTER.search(name))


KEY_GRID = os.environ.get("TCE_PROF_KEY_GRID", "0") == "1"  # one row per (kernel, grid size): a driver that runs several shapes through one kernel (profile_r5.sh)


def short(name, row=None):
    i = name.find(MINE)
    s = name[i:i + 110] if i >= 0 else name[:110]
    if KEY_GRID and row is not None:
        if row.get("Grid_Size_X"):  # kernel trace: per dimension; counter collection: the product
            gs = int(row["Grid_Size_X"]) * int(row.get("Grid_Size_Y") or 1) * int(row.get("Grid_Size_Z") or 1)
        else:
            gs = row.get("Grid_Size")
        if gs:
            s += f" [grid {gs}]"
    return s


def segmented(rows, name_col):
    """KEY_GRID: the driver runs its cases one after the other, a few launches each -- two cases may share (kernel, grid) (512 x 768 x 768 and 512 x 768 x 3072): a key's
    dispatches are cut into runs of consecutive dispatches (in dispatch order, among this library's kernels) and the run's index joins the key: "... #0", "... #1"."""
    rows = sorted(rows, key=lambda r: int(r["Dispatch_Id"]))
    seen, last, out = defaultdict(int), None, []
    for r in rows:
        k = short(r[name_col], r)
        if k != last:
            seen[k] += 1
            last = k
        out.append((k + f" #{seen[k] - 1}", r))
    return out


avg_ns = {}
for f in sorted(glob.glob(os.path.join(root, "kt", "**", "*kernel_stats.csv"), recursive=True)):
    print("== kernel stats (rocprofv3 --kernel-trace --stats):", os.path.relpath(f, root))
    for row in csv.DictReader(open(f)):
        if mine(row.get("Name", "")):
            avg_ns[short(row["Name"])] = float(row["AverageNs"])
            print("  ", short(row["Name"]), "| calls", row.get("Calls"), "| avg ns", row.get("AverageNs"), "| min", row.get("MinNs"), "| max", row.get("MaxNs"),
                  "| stddev", row.get("StdDev"))
for f in sorted(glob.glob(os.path.join(root, "kt", "**", "*kernel_trace.csv"), recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if mine(r["Kernel_Name"])]
    by = defaultdict(list)
    for k_, r in (segmented(rows, "Kernel_Name") if KEY_GRID else [(r["Kernel_Name"], r) for r in rows]):
        by[k_].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r))
    print("== kernel trace:", os.path.relpath(f, root), len(rows), "dispatches of this library")
    for k, v in sorted(by.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
        d = sorted(e - s for s, e, _ in v)
        v.sort(key=lambda t: t[0])
        gaps = sorted(v[i + 1][0] - v[i][1] for i in range(len(v) - 1))
        r0 = v[0][2]
        if KEY_GRID:
            avg_ns[k] = sum(d) / len(d)
        print(f"   {k if KEY_GRID else short(k)}\n      n={len(d)} dur ns min/med/mean/max = {d[0]}/{d[len(d)//2]}/{sum(d)//len(d)}/{d[-1]}  gap ns med = {gaps[len(gaps)//2] if gaps else None}"
              f"  grid={r0.get('Grid_Size_X', r0.get('Grid_Size'))} wg={r0.get('Workgroup_Size_X', r0.get('Workgroup_Size'))} vgpr={r0.get('VGPR_Count')} sgpr={r0.get('SGPR_Count')} lds={r0.get('LDS_Block_Size')}")
med = defaultdict(dict)
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        acc = defaultdict(lambda: defaultdict(list))
        n = 0
        prow = [r for r in csv.DictReader(open(f)) if mine(r["Kernel_Name"])]
        if KEY_GRID:  # one row per (dispatch, counter): segment on the dispatches, then spread the keys over their counter rows
            firsts = {}
            for r in prow:
                firsts.setdefault(r["Dispatch_Id"], r)
            keyof = {r["Dispatch_Id"]: k_ for k_, r in segmented(list(firsts.values()), "Kernel_Name")}
        for r in prow:
            acc[keyof[r["Dispatch_Id"]] if KEY_GRID else short(r["Kernel_Name"], r)][r["Counter_Name"]].append(float(r["Counter_Value"]))
            n += 1
        print("== pmc:", os.path.relpath(f, root), n, "rows of this library")
        for k, cs in acc.items():
            print("   ", k)
            for c, vals in cs.items():
                vals.sort()
                med[k][c] = vals[len(vals) // 2]
                print(f"       {c}: n={len(vals)} median={vals[len(vals)//2]:.5g} mean={sum(vals)/len(vals):.5g}")
print("== derived (per launch, medians)")
for k, m in med.items():
    out = []
    if "FETCH_SIZE" in m:
        out.append(f"HBM read traffic = FETCH_SIZE*1024*2 = {m['FETCH_SIZE'] * 2048 / 1e6:.2f} MB")
    if "WRITE_SIZE" in m:
        out.append(f"HBM write traffic = {m['WRITE_SIZE'] * 2048 / 1e6:.3f} MB")
    if "SQ_INSTS_VALU" in m and "SQ_WAVES" in m:
        out.append(f"VALU instructions per wave = {m['SQ_INSTS_VALU'] / m['SQ_WAVES']:.0f}")
    if "SQ_INSTS_VALU" in m and m.get("SQ_INSTS_MFMA"):
        out.append(f"SQ_INSTS_VALU / SQ_INSTS_MFMA = {m['SQ_INSTS_VALU'] / m['SQ_INSTS_MFMA']:.2f} (the counter includes the MFMAs: {m['SQ_INSTS_VALU'] / m['SQ_INSTS_MFMA'] - 1:.2f} other vector instructions per MFMA)")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("GRBM_GUI_ACTIVE"):
        # SQ_VALU_MFMA_BUSY_CYCLES sums over the chip's 1024 SIMDs in units of 4 cycles? -- printed raw beside the launch's cycles; the ratio quoted in DESIGN is busy / (GRBM_GUI_ACTIVE x SIMDs)
        out.append(f"SQ_VALU_MFMA_BUSY_CYCLES = {m['SQ_VALU_MFMA_BUSY_CYCLES']:.4g}, GRBM_GUI_ACTIVE = {m['GRBM_GUI_ACTIVE']:.4g} (summed over the 8 XCDs): matrix pipe busy = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / 8.0 * 1024):.2f} of the launch's SIMD-cycles (1024 SIMDs x the launch's cycles)")
    if "SQ_BUSY_CYCLES" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" not in m:
        out.append(f"SQ_VALU_MFMA_BUSY_CYCLES = {m['SQ_VALU_MFMA_BUSY_CYCLES']:.4g}")
    if "SQ_INSTS_LDS" in m and m.get("SQ_INSTS_MFMA"):
        out.append(f"LDS instructions per MFMA = {m['SQ_INSTS_LDS'] / m['SQ_INSTS_MFMA']:.2f}; bank-conflict cycles / LDS active cycles = {m.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, m.get('SQ_ACTIVE_INST_LDS', 1)):.3f}")
    if "SQ_WAIT_ANY" in m and "SQ_WAVE_CYCLES" in m:
        out.append(f"wave-cycles parked on s_waitcnt/barrier = {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.2f}, stalled at issue = {m.get('SQ_WAIT_INST_ANY', 0) / m['SQ_WAVE_CYCLES']:.2f}, "
                   f"issuing = {m.get('SQ_ACTIVE_INST_ANY', 0) / m['SQ_WAVE_CYCLES']:.2f}")
    if k in avg_ns and "FETCH_SIZE" in m:
        out.append(f"traffic / avg duration = {m['FETCH_SIZE'] * 2048 / avg_ns[k]:.0f} GB/s")
    print("   ", k)
    for o in out:
        print("       " + o)

# machine-readable traffic figure for bench.py's roofline.traffic (dominant GEMV kernel only)
import json
for k, m in med.items():
    if "w4a16_gemv" in k and "FETCH_SIZE" in m:
        n = 0
        for d in glob.glob(os.path.join(root, "pmc_fetch", "**", "*counter_collection.csv"), recursive=True):
            n += sum(1 for r in csv.DictReader(open(d)) if mine(r["Kernel_Name"]) and r["Counter_Name"] == "FETCH_SIZE")
        import hashlib
        csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tinychatengine_amd", "csrc")
        srcs = ["w4a16_gemv_i8.hip", "w4a16_mfma_layout.hpp"] if "gemv_i8" in k else ["w4a16_gemv.hip"]
        sha = hashlib.sha256(b"".join(open(os.path.join(csrc, f), "rb").read() for f in srcs)).hexdigest()
        json.dump({"kernel": k, "kernel_sources": srcs, "kernel_sources_sha256": sha,
                   "hbm_read_bytes_per_launch": int(m["FETCH_SIZE"] * 2048), "fetch_size_kib_median": m["FETCH_SIZE"],
                   "dispatches": n, "algorithmic_bytes_per_launch": int(os.environ.get("TCE_ALGO_BYTES", "46910464")),
                   "correction": "FETCH_SIZE [KiB] x 1024 x 2 (gfx950 reports half of a wide coalesced streaming read)"},
                  open(os.path.join(root, "traffic.json"), "w"), indent=1)
        break
