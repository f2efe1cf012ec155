"""Per-kernel summary (name, grid, count, median / mean / min duration in us) of a rocprofv3 results database (rocprofv3 --kernel-trace -d DIR -o NAME writes
DIR/NAME_results.db on ROCm 7.2).  Only this library's kernels (tce::) unless --all.

    python scripts/summarize_kernel_db.py gpurun_out/prefill_prof/prefill_results.db [--all]
"""
import collections
import sqlite3
import statistics
import sys

db = sqlite3.connect(sys.argv[1])
show_all = "--all" in sys.argv
rows = list(db.cursor().execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start"))
agg = collections.OrderedDict()
for name, s, e, gx, gy, gz, wx in rows:
    if "tce::" not in name and not show_all:
        continue
    short = name.replace("tce::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    agg.setdefault((short, gx, gy, gz, wx), []).append((e - s) / 1e3)
print("# kernel | grid threads | workgroup | n | us median / mean / min")
for k, v in agg.items():
    print(f"{k[0][:70]:70s} grid {str(k[1]) + 'x' + str(k[2]) + 'x' + str(k[3]):16s} wg {k[4]:4d} n={len(v):4d}  {statistics.median(v):8.2f} / {statistics.mean(v):8.2f} / {min(v):8.2f}")
