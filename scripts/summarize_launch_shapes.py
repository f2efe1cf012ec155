#!/usr/bin/env python3
"""Per launch shape of a decode token: rocprofv3 kernel durations (graph-replayed, same command and box as the HIP-event numbers printed
by bench.py --shapes-only), the separate PMC passes (FETCH_SIZE x 1024 x 2 on gfx950; SQ wave-cycle split), algorithmic bytes and the
fraction of the 8 TB/s HBM peak -- so that every `frac_of_8TBs` of the bench line's decode_launch_shapes can be recomputed from this file.

    python scripts/summarize_launch_shapes.py <profile dir> <bench --shapes-only json (graph)>

A dispatch is told apart by (kernel name, grid size): the launch shapes of the baseline-named workload all have different grids."""
import csv, glob, json, os, sys
from collections import defaultdict

root, bench_json = sys.argv[1], sys.argv[2]
bench = json.loads([l for l in open(bench_json) if l.startswith("{")][-1])["decode_launch_shapes"]


def rows_of(pattern):
    out = []
    for f in sorted(glob.glob(os.path.join(root, pattern), recursive=True)):
        out += [r for r in csv.DictReader(open(f)) if "tce::" in r.get("Kernel_Name", "")]
    return out


def key(r):
    name = r["Kernel_Name"]
    i = name.find("tce::")
    name = name[i:i + 90]
    return name, int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0)


kt = defaultdict(list)
for r in rows_of("kt/**/*kernel_trace.csv"):
    kt[key(r)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
pmc = defaultdict(lambda: defaultdict(list))
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    for r in rows_of(os.path.relpath(d, root) + "/**/*counter_collection.csv"):
        pmc[key(r)][r["Counter_Name"]].append(float(r["Counter_Value"]))

# expected (grid threads, workgroup) -> launch label + algorithmic bytes, from bench's own table (bytes) and the dispatcher's geometry
labels = {}
for row in bench["linears"]:
    labels[row["launch"]] = row
print(f"# per-launch-shape summary; HIP-event numbers: {bench['mode']}")
print("# kernel | grid threads x workgroup | n | rocprofv3 dur us min / median / mean | HBM read MB = FETCH_SIZE*1024*2 (median) | wave-cycles: parked / stalled at issue / issuing")
for k, d in sorted(kt.items(), key=lambda kv: -sum(kv[1])):
    d.sort()
    m = {c: sorted(v)[len(v) // 2] for c, v in pmc.get(k, {}).items()}
    extra = ""
    if "FETCH_SIZE" in m:
        extra += f" | HBM read {m['FETCH_SIZE'] * 2048 / 1e6:.2f} MB"
    if "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"]:
        wc = m["SQ_WAVE_CYCLES"]
        extra += f" | parked {m.get('SQ_WAIT_ANY', 0) / wc:.2f} stalled {m.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} issuing {m.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f}"
    if "SQ_INSTS_VALU" in m and "SQ_WAVES" in m and m["SQ_WAVES"]:
        extra += f" | VALU per wave {m['SQ_INSTS_VALU'] / m['SQ_WAVES']:.0f}"
    print(f"{k[0]} | {k[1]} x {k[2]} | n={len(d)} | {d[0] / 1e3:.2f} / {d[len(d) // 2] / 1e3:.2f} / {sum(d) / len(d) / 1e3:.2f}{extra}")
print("\n# bench.py --shapes-only on the same box, same session (HIP events around graph replays; includes inter-kernel gaps):")
for row in bench["linears"]:
    print(f"  {row['launch']}: {row['us']} us, {row['bytes']} B -> {row['GBs']} GB/s = {row['frac_of_8TBs']} of 8 TB/s")
for row in bench.get("attention_step", []):
    print(f"  {row['launch']}: {row['us']} us, cache {row['kv_cache_bytes']} B -> {row['GBs']} GB/s; cut {row['cut']}")
print("\n# to recompute a fraction: algorithmic bytes (above) / rocprofv3 median duration of the matching (kernel, grid) row / 8e12.")
print("# grids of the baseline-named workload: qkv 12288 rows -> 768 workgroups x 256; o 4096 -> 512 x 256; gate+up 22016 -> 2752 x 256; down 4096 x 11008 -> 256 x 512; lm_head 32000 -> 2000 x 256")
