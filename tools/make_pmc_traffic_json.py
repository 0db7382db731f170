"""profiles/pmc_traffic.json from the raw record tools/gpu_pmc_traffic.sh prints (separate rocprofv3 --pmc passes of
tools/prof_gemm_shapes.py): bytes per launch of the dominant GEMM launches, keyed by (a_trans b_trans epilogue kernel M N K), plus
the sha of the library the passes ran -- bench.py quotes an entry only on a line measured with that very build.

    python tools/make_pmc_traffic_json.py profiles/r05_final_gemm_hbm_traffic_pmc.txt <libofhip_sha16> > profiles/pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 reports half of a wide coalesced streaming read)."""
import ast
import json
import re
import sys

# kernel symbol fragment -> (key, launch description, algorithmic bytes); launch index inside the symbol's rows of prof_gemm_shapes.py
LAUNCHES = {
    "of_gemm_w4m_kernel<true, true, 5, false>": ("1 1 5 w4m256 8192 2048 8192", 0,
        "dW = dY^T X, 8192x2048x8192, fp32 gradient written with beta = 0 (the step epilogue leaves weight gradients for the backward to overwrite)",
        2 * (8192 * 8192 + 8192 * 2048) + 4 * 8192 * 2048),
    "of_gemm_w4m_kernel<false, false, 2, false>": ("0 0 2 w4m256 8192 2048 8192", 0, "ffn down + gate + residual, 8192x2048x8192, fp32 stream",
        2 * (8192 * 8192 + 2048 * 8192) + 2 * 4 * 8192 * 2048),
    "of_gemm_w4m_kernel<false, false, 1, false>": ("0 0 1 w4m256 8192 8192 2048", 0, "ffn up + GELU (a and b), 8192x8192x2048",
        2 * (8192 * 2048 + 8192 * 2048) + 2 * 2 * 8192 * 8192),
    "of_gemm_w4h_kernel<true, 3, 0>": ("0 1 3 w4h256x128 8192 8192 2048", 0,
        "ffn dX * gate * gelu'(a) + gate-gradient dot, 8192x8192x2048 (two workgroups per CU, 256x128 tiles)",
        2 * (8192 * 2048 + 2048 * 8192) + 2 * 8192 * 8192 + 2 * 8192 * 8192),
    # the fused attention branch (not an of_gemm launch: bench.py does not quote it; here for the record): x once + once more for the residual is
    # NOT counted twice -- algorithmic = x (fp32) + LN(x), q, o (bf16, saved) + y (fp32) + LN_ff(y) (bf16) + the two packed weights + k | v
    "of_xattn_fused_fwd_kernel<2048>": ("xattn_fused_fwd 8192 2048", 0,
        "LN -> to_q -> windowed attention -> to_out + gate + residual -> LN_ff, 8192 rows x 2048, 8 heads x 64, 2 x 64 media tokens per sequence",
        4 * 8192 * 2048 + 2 * 8192 * 2048 + 2 * 2 * 8192 * 512 + 4 * 8192 * 2048 + 2 * 8192 * 2048 + 2 * 2 * 512 * 2048 + 2 * 4096 * 1024),
}


def main():
    raw, sha = sys.argv[1], sys.argv[2]
    rows = {}
    for line in open(raw):
        m = re.match(r"\((.*)\) (\[.*\])\s*$", line.strip())
        if not m:
            continue
        name, grid, counter = ast.literal_eval("(" + m.group(1) + ")")
        rows[(name, counter)] = ast.literal_eval(m.group(2))
    kernels = {}
    for frag, (key, idx, desc, alg) in LAUNCHES.items():
        got = {c: v for (n, c), v in rows.items() if frag in n}
        if not {"FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"} <= set(got):
            continue
        hit, miss = got["TCC_HIT_sum"][idx], got["TCC_MISS_sum"][idx]
        kernels[key] = {"symbol": frag, "launch": desc, "fetch_bytes": int(got["FETCH_SIZE"][idx] * 1024 * 2),
                        "write_bytes": int(got["WRITE_SIZE"][idx] * 1024), "algorithmic_bytes": alg,
                        "l2_hit_rate": round(hit / (hit + miss), 4)}
    json.dump({"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum in separate passes (tools/gpu_pmc_traffic.sh, "
                         f"tools/prof_gemm_shapes.py; raw: {raw}); FETCH_SIZE (KB) doubled per MI355X_MICROARCH.md (gfx950 under-reports wide "
                         f"coalesced reads by 2x) and it counts Infinity-Cache hits",
               "libofhip_sha16": sha, "unit": "bytes per launch", "kernels": kernels,
               "key": "a_trans b_trans epilogue kernel M N K -- a launch of another shape, or a line measured with another build of the "
                      "library than libofhip_sha16, gets traffic: null"}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
