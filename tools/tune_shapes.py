"""Add shapes to the committed TunableOp table without re-tuning the ones it already holds:
    python tools/tune_shapes.py <table.csv> <out.csv> M,N,K [M,N,K ...]      (y[M,N] = x[M,K] @ w[N,K]^T, bf16)
"""
import sys
import torch
import torch.cuda.tunable as tun

table, out = sys.argv[1], sys.argv[2]
tun.enable(True)
tun.tuning_enable(True)
tun.set_max_tuning_duration(30)
tun.set_max_tuning_iterations(100)
assert tun.read_file(table), "table rejected (validators)"
n0 = len(tun.get_results())
for spec in sys.argv[3:]:
    M, N, K = (int(v) for v in spec.split(","))
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    for _ in range(3):
        torch.mm(x, w.t())
torch.cuda.synchronize()
res = tun.get_results()
print("entries", n0, "->", len(res))
have = {(l.split(",")[0], l.split(",")[1]) for l in open(table) if not l.startswith("Validator")}
with open(table) as f, open(out, "w") as g:
    g.write(f.read())
    for op, params, sol, ms in res:
        if (op, params) not in have:
            g.write(f"{op},{params},{sol},{ms}\n")
