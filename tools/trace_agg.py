"""Group a rocprofv3 *_kernel_trace.csv by (kernel, grid) and print average durations (for per-shape kernel timing)."""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-48:]
    key = (n, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""))
    agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    if len(v) >= int(sys.argv[2]) if len(sys.argv) > 2 else 10:
        print(k, len(v), "avg %.1f us" % (sum(v) / len(v)))
