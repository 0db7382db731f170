"""GPU idle time inside the timed steps of a `rocprofv3 --kernel-trace` run of bench.py: union of the kernels' [start, end] intervals over all
streams against the wall span, the largest gaps and what ran on either side of them.  PROFILING TOOL.
    python tools/kernel_trace_gaps.py <run_kernel_trace.csv> [steps_at_the_end=3]"""
import csv, json, sys

rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
# the timed steps: from the nsteps-th last of_step_advance kernel's predecessor boundary -- use the optimizer's step counter kernel as the step marker
marks = [e[1] for e in ev if "of_step_advance" in e[2]]
if len(marks) < nsteps + 1:
    raise SystemExit(f"only {len(marks)} step markers")
t0, t1 = marks[-nsteps - 1], marks[-1]
win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
busy, cur_s, cur_e, gaps, last_name = 0, None, None, [], None
for s, e, name in win:
    if cur_e is None:
        cur_s, cur_e, last_name = s, e, name
        continue
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last_name[:70], name[:70]))
        cur_s, cur_e, last_name = s, e, name
    elif e > cur_e:
        cur_e, last_name = e, name
busy += cur_e - cur_s
span = t1 - t0
gaps.sort(reverse=True)
hist = {"<2us": 0, "2-5us": 0, "5-20us": 0, ">20us": 0}
tot = {"<2us": 0, "2-5us": 0, "5-20us": 0, ">20us": 0}
for g, _, _ in gaps:
    k = "<2us" if g < 2000 else "2-5us" if g < 5000 else "5-20us" if g < 20000 else ">20us"
    hist[k] += 1
    tot[k] += g
print(json.dumps({"steps": nsteps, "ms_per_step_span": round(span / nsteps / 1e6, 3), "busy_ms_per_step": round(busy / nsteps / 1e6, 3),
                  "idle_ms_per_step": round((span - busy) / nsteps / 1e6, 3), "kernels_per_step": round(len(win) / nsteps, 1),
                  "gaps_per_step": {k: round(v / nsteps, 1) for k, v in hist.items()},
                  "gap_ms_per_step": {k: round(v / nsteps / 1e6, 3) for k, v in tot.items()},
                  "largest_gaps_us": [(round(g / 1e3, 1), a, b) for g, a, b in gaps[:12]]}))
