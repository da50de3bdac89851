"""Matrix-pipe utilisation per kernel from one rocprofv3 PMC pass
   (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv):
   usage: python tools/mfma_util.py <dir with *counter_collection.csv> [out.json]"""
import csv, glob, json, os, re, sys
from collections import defaultdict

per = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for fn in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    with open(fn) as f:
        for r in csv.DictReader(f):
            k = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]).split("(")[0]
            per[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                cnt[k] += 1
out = {}
for k, c in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_INSTS_MFMA", 0))[:12]:
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    if gui <= 0 or c.get("SQ_INSTS_MFMA", 0) <= 0:
        continue
    # SQ_VALU_MFMA_BUSY_CYCLES is summed over all 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs (cross-check: a
    # v_mfma_f32_32x32x16_bf16 occupies the pipe 32 cycles, so busy ~= insts * 32 / 1024 SIMD-cycles per launch)
    out[k] = {"launches": cnt[k], "mfma_insts_per_launch": c["SQ_INSTS_MFMA"] / max(cnt[k], 1),
              "active_cycles_per_launch": gui / 8.0 / max(cnt[k], 1),
              "mfma_busy_frac_of_active_cycles": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8.0 * 1024.0),
              "mfma_busy_frac_from_inst_count": c["SQ_INSTS_MFMA"] * 32.0 / 1024.0 / (gui / 8.0)}
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
