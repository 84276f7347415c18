"""rocprofv3 --stats CSV -> markdown summary under profiles/. Usage:
python tools/prof_summary.py gpurun_out/prof3/r1_kernel_stats.csv profiles/NAME.md "title" [steps]"""
import csv
import sys

src, dst, title = sys.argv[1:4]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 7
rows = list(csv.DictReader(open(src)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(dst, "w") as f:
    f.write(f"# {title}\n\n")
    f.write("`rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline` "
            f"on 1x MI355X ({steps} optimizer steps of Slam-358M B=8 T=1024, plus the roofline probe's 23 gate|up GEMM launches "
            "and model init).\n\n")
    f.write(f"Total kernel time {tot/1e6:.1f} ms -> {tot/1e6/steps:.1f} ms of kernels per optimizer step.\n\n")
    f.write("| kernel | calls | total ms | avg us | % | ms/step |\n|---|---|---|---|---|---|\n")
    for r in rows[:34]:
        n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        f.write(f"| `{n[:100]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | "
                f"{float(r['Percentage']):.2f} | {float(r['TotalDurationNs'])/1e6/steps:.2f} |\n")
print(open(dst).read()[:2500])
