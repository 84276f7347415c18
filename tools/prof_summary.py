"""rocprofv3 --stats CSV -> markdown summary under profiles/. Usage:
python tools/prof_summary.py gpurun_out/prof3/r1_kernel_stats.csv profiles/NAME.md "title" [steps]"""
import csv
import sys

src, dst, title = sys.argv[1:4]
import os
CMD = os.environ.get("PROF_CMD", "`rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline "
                     "--no-extras` on 1x MI355X (7 optimizer steps of Slam-358M B=8 T=1024, plus the probes bench.py runs after the timed region - 90 "
                     "gate|up GEMM launches of the roofline probe, the RMSNorm / AdamW / gradient-norm launches of `hbm_kernels` - and model init). "
                     "With the wgrad side stream the kernels of the two streams overlap: per-kernel times are launch durations, their sum is more "
                     "than the step.")
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 7
rows = list(csv.DictReader(open(src)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(dst, "w") as f:
    f.write(f"# {title}\n\n")
    f.write(CMD + "\n\n")
    f.write(f"Total kernel time {tot/1e6:.1f} ms -> {tot/1e6/steps:.1f} ms of kernels per optimizer step.\n\n")
    f.write("| kernel | calls | total ms | avg us | % | ms/step |\n|---|---|---|---|---|---|\n")
    for r in rows[:34]:
        n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        f.write(f"| `{n[:100]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | "
                f"{float(r['Percentage']):.2f} | {float(r['TotalDurationNs'])/1e6/steps:.2f} |\n")
# optional: kernel_trace.csv of the same run -> GEMM launches split by grid (one kernel name serves every shape)
if len(sys.argv) > 5:
    import collections
    names = {4864: "gate|up fwd on the 128x128 kernel", 2432: "down dgrad (+fused SwiGLU bwd), N4864 K896",
             576: "qkv fwd (+bias, RoPE), N1152 K896", 448: "N896 shapes: o fwd/dgrad, down fwd, qkv dgrad, gate|up dgrad", 256: "LM head fwd, N512"}
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(sys.argv[5])):
        n = r["Kernel_Name"]
        if "gemm_kernel" in n or "gemm_tn_bal" in n or "gemm_tn_224" in n or "gemm_nt_256" in n:
            kind = "wgrad (balanced TN)" if "tn_bal" in n else "wgrad 256x224 8-phase" if "tn_224" in n else "NT 256x256 8-phase" if "gemm_nt_256" in n else "NT"
            agg[(kind, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    with open(dst, "a") as f:
        f.write("\n## GEMM launches by grid (from the kernel trace of the same run)\n\n| kernel | blocks | launches | avg us | total ms | shape |\n|---|---|---|---|---|---|\n")
        for (kind, blocks), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            f.write(f"| {kind} | {blocks} | {len(v)} | {sum(v)/len(v):.1f} | {sum(v)/1e3:.1f} | {names.get(blocks, '') if kind == 'NT' else 'gate|up fwd (+fused SwiGLU), M8192 N9728 K896  <- bench.py roofline kernel' if blocks == 1216 else 'down dgrad (+fused SwiGLU bwd), N4864 K896' if blocks == 608 and '256' in kind else ''} |\n")
print(open(dst).read()[:2500])
