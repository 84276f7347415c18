"""Per-kernel PMC summary of tools/pmc_step.sh passes -> markdown. Usage:
  python tools/pmc_summary.py gpurun_out/<tag>_pmc_ profiles/NAME.md "title"
Reads every <prefix><i>/**/p_counter_collection.csv (rocprofv3 csv: one row per dispatch and counter), keeps the dispatches
of the LAST optimizer step, averages each counter per kernel name (and per grid size for the GEMM kernels: one kernel
serves several shapes). FETCH_SIZE is doubled (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md HBM section);
FETCH/WRITE are KiB."""
import collections
import csv
import glob
import sys

prefix, dst, title = sys.argv[1:4]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(prefix + "*/**/*counter_collection.csv", recursive=True)):
    rows = list(csv.DictReader(open(f)))
    if not rows:
        continue
    # dispatches of the last optimizer step = after the second-to-last adamw dispatch
    ids = sorted({int(r["Dispatch_Id"]) for r in rows if "adamw" in r["Kernel_Name"]})
    runs = []  # the optimizer is a run of consecutive AdamW dispatches (one per weight class)
    for i in ids:
        if runs and i == runs[-1][1] + 1:
            runs[-1][1] = i
        else:
            runs.append([i, i])
    lo = runs[-2][1] if len(runs) >= 2 else 0
    hi = runs[-1][1] if runs else 1 << 60
    # forward / backward by position relative to the step's loss kernel (the persistent 256x256 launches of the gate|up
    # forward and of the down-proj dgrad share kernel name and grid)
    ce = [int(r["Dispatch_Id"]) for r in rows if lo < int(r["Dispatch_Id"]) <= hi and "::ce_" in r["Kernel_Name"].replace("void ", "::")]
    ce_id = max(ce) if ce else lo
    for r in rows:
        d = int(r["Dispatch_Id"])
        if not (lo < d <= hi):
            continue
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if name.startswith("gemm") or name.startswith("reduce"):
            name += f" [{int(r['Grid_Size']) // max(1, int(r['Workgroup_Size']))} blocks]"
        if name.startswith("gemm_nt_256") or name.startswith("gemm_kernel"):
            name += " fwd" if d < ce_id else " bwd"
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["FETCH_SIZE", "WRITE_SIZE", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
        "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"]
with open(dst, "w") as f:
    f.write(f"# {title}\n\n")
    f.write("`tools/pmc_step.sh`: `rocprofv3 --kernel-trace --pmc <set>` over `tools/one_step.py` (three optimizer steps; one counter set per "
            "pass; counter collection serialises the dispatches of the two backward streams), dispatches of the last optimizer step, mean per launch. "
            "`read MB` = FETCH_SIZE x 2 (gfx950 correction) x 1024 B; `written MB` = WRITE_SIZE x 1024 B. `mfma busy` = "
            "SQ_VALU_MFMA_BUSY_CYCLES / (GPU cycles x 4 SIMDs x 256 CUs), GPU cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs); `wait` = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves parked in "
            "s_waitcnt / barrier), `stall` = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (issue stalls); `lds conflict` = SQ_LDS_BANK_CONFLICT / "
            "SQ_LDS_IDX_ACTIVE. Launches with fewer than 256 one-per-CU blocks (the 256x224 kernels planned as background launches) show "
            "their chip-level figure and, in brackets, the figure per OCCUPIED CU (x 256 / blocks).\n\n")
    f.write("| kernel | launches | read MB | written MB | GPU cycles | mfma busy | wait | stall | lds conflict |\n|---|---|---|---|---|---|---|---|---|\n")
    def mean(name, c):
        v = agg[name].get(c)
        return sum(v) / len(v) if v else None
    order = sorted(agg, key=lambda n: -(mean(n, "GRBM_GUI_ACTIVE") or 0) * len(agg[n].get("GRBM_GUI_ACTIVE", [1])))
    for n in order:
        if n.startswith("at::") or "rocclr" in n:
            continue
        fe, wr, gr = mean(n, "FETCH_SIZE"), mean(n, "WRITE_SIZE"), mean(n, "GRBM_GUI_ACTIVE")
        gr = gr / 8 if gr is not None else None
        mb, wc, wa, wi = mean(n, "SQ_VALU_MFMA_BUSY_CYCLES"), mean(n, "SQ_WAVE_CYCLES"), mean(n, "SQ_WAIT_ANY"), mean(n, "SQ_WAIT_INST_ANY")
        lc, li = mean(n, "SQ_LDS_BANK_CONFLICT"), mean(n, "SQ_LDS_IDX_ACTIVE")
        k = len(next(iter(agg[n].values())))
        fmt = lambda x, s="{:.1f}": "" if x is None else s.format(x)  # noqa: E731
        busy = fmt(mb / (gr * 1024) if mb and gr else None, '{:.3f}')
        import re
        mblk = re.search(r"\[(\d+) blocks\]", n)
        if busy and mblk and "_224_" in n and int(mblk.group(1)) < 256:
            busy += f" ({mb / (gr * 1024) * 256 / int(mblk.group(1)):.2f})"
        f.write(f"| `{n[:70]}` | {k} | {fmt(fe * 2 * 1024 / 1e6 if fe is not None else None)} | {fmt(wr * 1024 / 1e6 if wr is not None else None)} | "
                f"{fmt(gr, '{:.0f}')} | {busy} | {fmt(wa / wc if wa and wc else None, '{:.2f}')} | "
                f"{fmt(wi / wc if wi and wc else None, '{:.2f}')} | {fmt(lc / li if lc is not None and li else None, '{:.3f}')} |\n")
print(open(dst).read()[:3000])
