"""Summary of tools/probes/power_or_stall.py run under one rocprofv3 --pmc pass. Usage:
  python tools/probes/power_or_stall_summary.py <dir with *_counter_collection.csv> <manifest.json> <out.md>
The GEMM dispatches (kernel names gemm_nt_*) appear in launch order = manifest order; the first two launches of a group warm up."""
import collections
import csv
import glob
import json
import statistics
import sys

d, man, out = sys.argv[1:4]
rows = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
disp = collections.OrderedDict()
for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
    if "gemm_nt_" not in r["Kernel_Name"]:
        continue
    e = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"].split("(")[0].replace("void (anonymous namespace)::", ""),
                                                 "wall_us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                                 "vgpr": r["VGPR_Count"], "agpr": r["Accum_VGPR_Count"], "wg": r["Workgroup_Size"]})
    e[r["Counter_Name"]] = float(r["Counter_Value"])
disp = list(disp.values())
groups = json.load(open(man))
lines = ["| shape | operands | main loop | kernel | wall us | TFLOP/s | GPU cycles (GRBM_GUI_ACTIVE / 8) | effective clock GHz | MFMA-busy cycles / SIMD | MFMA busy | issue stall (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES) | parked (SQ_WAIT_ANY / SQ_WAVE_CYCLES) | MFMA instructions |",
         "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
i = 0
med = statistics.median
for g in groups:
    ds = disp[i:i + g["launches"]][2:]
    i += g["launches"]
    if not ds:
        continue
    wall = med(x["wall_us"] for x in ds)
    cyc = med(x.get("GRBM_GUI_ACTIVE", 0) for x in ds) / 8
    busy = med(x.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for x in ds)
    wc = med(x.get("SQ_WAVE_CYCLES", 0) for x in ds) or 1
    lines.append(f"| {g['shape']} {g['M']}x{g['N']}x{g['K']} | {g['fill']} | {g['variant']} | `{ds[0]['name'][:40]}` ({ds[0]['wg']} thr, {ds[0]['vgpr']}+{ds[0]['agpr']} regs) | {wall:.1f} | "
                 f"{2.0 * g['M'] * g['N'] * g['K'] / wall / 1e6:.0f} | {cyc:.0f} | {cyc / wall / 1e3:.3f} | {busy / 1024:.0f} | {busy / (cyc * 1024):.3f} | "
                 f"{med(x.get('SQ_WAIT_INST_ANY', 0) for x in ds) / wc:.2f} | {med(x.get('SQ_WAIT_ANY', 0) for x in ds) / wc:.2f} | {med(x.get('SQ_INSTS_MFMA', 0) for x in ds):.0f} |")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
