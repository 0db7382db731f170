#!/usr/bin/env python
"""Group a rocprofv3 `*_kernel_stats.csv` of `bench.py` into kernel families and print ms per step for each (DESIGN.md section 8 is this
table).  usage: python tools/kernel_stats_families.py <kernel_stats.csv> <steps in the run, warm-up included> [--json]
PROFILING TOOL, not part of the library."""
import csv
import json
import sys

FAMILIES = [  # first match wins
    ("ours: of_gemm big tiles (w4h/w4m/pp)", ("of_gemm_w4h_kernel", "of_gemm_w4m_kernel", "of_gemm_pp_kernel")),
    ("ours: of_gemm mid tiles / batched dW / split-K", ("of_gemm_mid", "of_splitk", "of_dot_finish", "of_gemm")),
    ("ours: fused attention branch (forward)", ("of_xattn_fused", "of_pack_frag16")),
    ("ours: attention (fwd, dq, dkv)", ("of_attn_",)),
    ("ours: LayerNorm (fwd, bwd, colsum)", ("of_ln_",)),
    ("ours: AdamW + grad-norm", ("of_adamw", "of_sumsq", "of_clip")),
    ("ours: cross-entropy", ("of_ce_",)),
    ("ours: elementwise (gelu, casts, adds)", ("of_quick_gelu", "of_ew", "of_to_bf16", "of_text_time", "of_")),
    ("vendor GEMM (hipBLASLt: frozen towers)", ("Cijk_",)),
    ("torch elementwise / fill / copy / rng", ("at::native", "rocclr", "at::cuda")),
]


def main():
    path, steps = sys.argv[1], float(sys.argv[2])
    fam = {name: [0.0, 0] for name, _ in FAMILIES}
    fam["other"] = [0.0, 0]
    total = 0.0
    for r in csv.DictReader(open(path)):
        ns, calls = float(r["TotalDurationNs"]), int(r["Calls"])
        total += ns
        for name, keys in FAMILIES:
            if any(k in r["Name"] for k in keys):
                fam[name][0] += ns
                fam[name][1] += calls
                break
        else:
            fam["other"][0] += ns
            fam["other"][1] += calls
    out = [{"family": k, "ms_per_step": round(v[0] / 1e6 / steps, 2), "launches_per_step": round(v[1] / steps, 1),
            "share": round(v[0] / total, 4)} for k, v in fam.items() if v[1]]
    out.sort(key=lambda e: -e["ms_per_step"])
    if "--json" in sys.argv:
        print(json.dumps({"source": path, "steps": steps, "kernel_ms_per_step": round(total / 1e6 / steps, 2), "families": out}))
        return
    print(f"{'family':52s} {'ms/step':>8s} {'launches':>9s} {'share':>6s}")
    for e in out:
        print(f"{e['family']:52s} {e['ms_per_step']:8.2f} {e['launches_per_step']:9.1f} {100 * e['share']:5.1f}%")
    print(f"{'sum of kernel durations':52s} {total / 1e6 / steps:8.2f}")


if __name__ == "__main__":
    main()
