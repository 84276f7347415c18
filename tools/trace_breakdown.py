"""Step breakdown from a rocprofv3 kernel trace of `bench.py` (one row per launch family, per optimizer step):
python tools/trace_breakdown.py gpurun_out/fin_prof/r3_kernel_trace.csv profiles/NAME.md "title" [steps=5] [instrumented steps to skip=3]

Optimizer steps are delimited by the AdamW launch; the steps of the timed region are the LAST `nsteps` steps before the
probes bench.py runs afterwards (every step before them is identical work). For each step: wall time (first kernel start
to AdamW end), the same split into forward (to the loss kernel), backward (to the last gradient kernel) and optimizer,
busy time per stream (union of launch intervals), and per launch family: stream, launches per step, average duration,
milliseconds per step. Launch durations on two overlapping streams add up to more than the wall time; the per-stream busy
union is what sums to the step."""
import collections
import csv
import re
import sys

src, dst, title = sys.argv[1:4]
nsteps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
skip_last = int(sys.argv[5]) if len(sys.argv) > 5 else 3  # bench.py runs 3 more steps with timing events around the gate|up launches
rows = []
for r in csv.DictReader(open(src)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Queue_Id"]),
                 int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["LDS_Block_Size"]), int(r["VGPR_Count"])))
rows.sort()


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*$", "", n)
    return n


# step boundaries: a step starts at the embedding gather, holds a loss kernel, and ends with the run of AdamW launches
# that follows its gradient-norm kernels (the probes bench.py runs afterwards launch AdamW without a forward)
steps, cur, state = [], None, 0   # state: 0 before loss, 1 loss seen, 2 norm seen, 3 inside the AdamW run
for r in rows:
    n = r[2]
    if "embed_fwd_kernel" in n:
        cur, state = [], 0
    if cur is None:
        continue
    if state == 3 and "adamw" not in n:
        steps.append(cur)
        cur = None
        continue
    cur.append(r)
    if short(n).startswith("ce_"):
        state = max(state, 1)
    elif "norm_finish" in n and state >= 1:
        state = 2
    elif "adamw" in n and state >= 2:
        state = 3
if cur and state == 3:
    steps.append(cur)
steps = steps[-(nsteps + skip_last):len(steps) - skip_last] if skip_last else steps[-nsteps:]
assert steps, "no optimizer steps found"


def union(iv):
    iv = sorted(iv)
    tot, ce = 0, None
    for s, e in iv:
        if ce is None or s > ce:
            tot += e - s
            ce = e
        elif e > ce:
            tot += e - ce
            ce = e
    return tot


fam = collections.OrderedDict()
walls, fw, bw, op, busy = [], [], [], [], collections.defaultdict(list)
gaps = collections.defaultdict(list)
for st in steps:
    t0 = st[0][0]
    t_ce = max(r[1] for r in st if short(r[2]).startswith(("ce_", "loss_finish")))
    t_opt = min(r[0] for r in st if "sumsq_chunks" in r[2] or "norm_finish" in r[2] or "adamw" in r[2])  # round 6: the norm is the finish kernel alone
    t1 = max(r[1] for r in st)
    walls.append(t1 - t0); fw.append(t_ce - t0); bw.append(t_opt - t_ce); op.append(t1 - t_opt)
    for q in set(r[3] for r in st):
        iv = [(r[0], r[1]) for r in st if r[3] == q]
        busy[q].append(union(iv))
    for phase, lo, hi in (("fwd", t0, t_ce), ("bwd", t_ce, t_opt), ("opt", t_opt, t1 + 1)):
        q0 = [r for r in st if r[3] == st[0][3] and lo <= r[1] - 1 < hi]
        g = sum(max(0, b[0] - a[1]) for a, b in zip(q0, q0[1:]))
        gaps[phase].append((g, len(q0)))
    for r in st:
        ph = "fwd" if r[1] <= t_ce else ("bwd" if r[0] < t_opt else "opt")
        key = (ph, short(r[2]), r[4], r[5], r[3])
        fam.setdefault(key, []).append(r[1] - r[0])

ns = len(steps)
mean = lambda v: sum(v) / len(v)
with open(dst, "w") as f:
    f.write(f"# {title}\n\n")
    f.write(f"Source: `{src}` (rocprofv3 --kernel-trace of `bench.py --steps 5 --warmup 2`); the {ns} optimizer steps of the timed region (the {skip_last} instrumented steps and the probes that follow are left out).\n\n")
    f.write(f"Step wall time (first kernel start → AdamW end): **{mean(walls)/1e6:.2f} ms** = forward {mean(fw)/1e6:.2f} + backward {mean(bw)/1e6:.2f} + optimizer {mean(op)/1e6:.2f} ms"
            " (rocprofv3 serialises nothing but adds ≈2-3 % to the un-profiled step).\n\n")
    f.write("| stream (HSA queue) | busy ms / step (union of its launch intervals) |\n|---|---|\n")
    for q, v in sorted(busy.items()):
        f.write(f"| {q}{' (caller)' if q == steps[0][0][3] else ' (weight-gradient side stream)'} | {mean(v)/1e6:.2f} |\n")
    f.write("\nIdle time between consecutive launches on the caller's stream (dependent-launch gaps + waits on the other stream):\n\n| phase | launches / step | gap ms / step | mean gap µs |\n|---|---|---|---|\n")
    for ph in ("fwd", "bwd", "opt"):
        g = mean([x[0] for x in gaps[ph]]); n = mean([x[1] for x in gaps[ph]])
        f.write(f"| {ph} | {n:.0f} | {g/1e6:.2f} | {g/1e3/max(1,n-1):.2f} |\n")
    f.write("\n## Launch families (per optimizer step)\n\n| phase | kernel | blocks | stream | launches / step | avg µs | ms / step |\n|---|---|---|---|---|---|---|\n")
    for ph in ("fwd", "bwd", "opt"):
        items = [(k, v) for k, v in fam.items() if k[0] == ph]
        items.sort(key=lambda kv: -sum(kv[1]))
        tot = 0
        for k, v in items:
            ms = sum(v) / 1e6 / ns
            tot += ms
            if ms < 0.01:
                continue
            blocks = f"{k[2]}" + (f"×{k[3]}" if k[3] > 1 else "")
            f.write(f"| {ph} | `{k[1][:90]}` | {blocks} | {'main' if k[4] == steps[0][0][3] else 'side'} | {len(v)/ns:.0f} | {mean(v)/1e3:.1f} | {ms:.3f} |\n")
        f.write(f"| {ph} | **sum of launch durations** | | | | | **{tot:.2f}** |\n")
print(open(dst).read())
