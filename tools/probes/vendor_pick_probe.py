"""Which vendor kernel: torch.mm on the up_proj / down_proj / Wqkv launches with the library's default heuristic, with the committed TunableOp
table (tuned on operands rotating through 1 GiB) and with round 4's table (tools/ab/tunableop_hot_tuned_r04.csv, tuned on cache-resident
operands), against of_gemm -- hot loop, rotating operand sets, and behind a 200-us streaming pass (one process per table: TunableOp state is
global).  PROFILING TOOL."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch
    from open_flamingo_amd.hip import abi
    from open_flamingo_amd.hip.ops import Ops
    from open_flamingo_amd.train import towers
    from bench_gemm_ab import make
    arm, table = sys.argv[1], sys.argv[2]
    n = towers.use_tuned_vendor_gemms(table) if table != "-" else 0
    ops = Ops.default()
    FILL = torch.randn(1 << 28, device="cuda").to(torch.bfloat16)
    FOUT = torch.empty_like(FILL)
    R = 6

    def med(ts):
        ts = sorted(ts)
        return round(ts[len(ts) // 2] * 1e3, 1)

    for name, M, N, K in (("up_proj", 8192, 8192, 2048), ("down_proj", 8192, 2048, 8192), ("Wqkv", 8192, 6144, 2048)):
        sets = [make(M, N, K, 0, 0, abi.EPI_STORE_BF16)[:3] for _ in range(R)]
        if arm == "ours":
            fn = lambda i: ops.gemm(sets[i][0], sets[i][1], sets[i][2])
        else:          # y = x W^T exactly as train/frozen_blocks.py calls it (no out=: the TunableOp route of at::mm)
            fn = lambda i: torch.mm(sets[i][0], sets[i][1].t())
        for i in range(8):
            fn(i % R)
        torch.cuda.synchronize()
        rec = dict(arm=arm, table_entries=n, case=name)
        for mode in ("hot loop", "rotating", "behind a 200 us pass"):
            ts = []
            for b in range(24):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if mode == "behind a 200 us pass":
                    ops.gelu_fwd(FILL, out=FOUT)
                s.record()
                fn(0 if mode == "hot loop" else b % R)
                e.record()
                ts.append((s, e))
            torch.cuda.synchronize()
            rec[mode + " us"] = med([a.elapsed_time(b) for a, b in ts[4:]])
        print(json.dumps(rec), flush=True)
        del sets
else:
    for arm, table in (("ours", "-"), ("vendor default", "-"), ("vendor, table tuned on rotating operands", os.path.join(ROOT, "open_flamingo_amd/train/tuned/tunableop_gfx950_of3b_cfg2.csv")),
                       ("vendor, round 4's table", os.path.join(ROOT, "tools/ab/tunableop_hot_tuned_r04.csv"))):
        out = subprocess.run([sys.executable, os.path.abspath(__file__), arm, table], capture_output=True, text=True)
        print("\n".join(l for l in out.stdout.splitlines() if l.startswith("{")) or ("FAILED " + out.stderr[-300:]), flush=True)
