"""Where does a big tile's time go?  tools/libofhip_tools.so (-DOF_TOOLS_BUILD) stamps the 100-MHz wall clock in wave 0 of every
workgroup of of_gemm_w4m_kernel: entry, prologue done (stage 0 landed), K loop done, last epilogue instruction issued, all stores
acknowledged -- plus the XCC / SE / CU it ran on.  Per launch this prints the median phase lengths and, per CU, the gap between one
workgroup's end and the next one's entry (launch + drain cost that a persistent kernel would not pay).  PROFILING TOOL."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open_flamingo_amd.hip import abi
from tools_lib import tools_ops
from bench_gemm_ab import make, timed

ops = tools_ops()
lib = ops.lib
lib.of_tools_set_stamp_buffer.argtypes = [ctypes.c_void_p]
E = abi
CASES = [("NT store_bf16", 8192, 8192, 2048, 0, 0, E.EPI_STORE_BF16), ("NT gelu 2 outputs", 8192, 8192, 2048, 0, 0, E.EPI_GELU),
         ("NN store_bf16", 8192, 8192, 2048, 0, 1, E.EPI_STORE_BF16), ("NN scale_dot", 8192, 8192, 2048, 0, 1, E.EPI_SCALE_DOT),
         ("NN dgelu_dot", 8192, 8192, 2048, 0, 1, E.EPI_DGELU_DOT), ("NT gate_resid fp32", 8192, 2048, 8192, 0, 0, E.EPI_GATE_RESID),
         ("TN acc_f32", 8192, 2048, 8192, 1, 1, E.EPI_ACC_F32), ("NT gate_resid K=512", 8192, 2048, 512, 0, 0, E.EPI_GATE_RESID),
         ("NN store K=8192", 8192, 2048, 8192, 0, 1, E.EPI_STORE_BF16)]
med = lambda t: float(t.double().median())
for name, M, N, K, ta, tb, epi in CASES:
    A, B, C, kw = make(M, N, K, ta, tb, epi)
    ntile = (M // 256) * (N // 256)
    fn = lambda: ops.gemm(A, B, C, ta=bool(ta), tb=bool(tb), epi=epi, safe=16, **kw)
    lib.of_tools_set_stamp_buffer(None)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    plain_us = min(timed(fn, 10) for _ in range(3)) * 1e3
    buf = torch.zeros(ntile, 8, dtype=torch.int64, device="cuda")
    lib.of_tools_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
    fn()
    torch.cuda.synchronize()
    lib.of_tools_set_stamp_buffer(None)
    s = buf.cpu()
    t = (s[:, :5] - s[:, 0].min()).double() / 100.0            # us since the first workgroup's entry
    hw = s[:, 7]
    cu_key = ((hw >> 32) & 0xf) * 4096 + ((hw >> 13) & 0x7) * 256 + ((hw >> 8) & 0xf)      # XCC, SE_ID [15:13], CU_ID [11:8]
    gaps_issue, gaps_ack, rounds = [], [], {}
    for key in cu_key.unique().tolist():
        idx = (cu_key == key).nonzero().flatten()
        order = idx[t[idx, 0].argsort()]
        rounds[len(order)] = rounds.get(len(order), 0) + 1
        for a, b in zip(order[:-1].tolist(), order[1:].tolist()):
            gaps_issue.append(float(t[b, 0] - t[a, 3]))
            gaps_ack.append(float(t[b, 0] - t[a, 4]))
    rec = dict(case=name, MNK=[M, N, K], tiles=ntile, launch_us_unstamped=round(plain_us, 1), cus_seen=int(cu_key.unique().numel()),
               tiles_per_cu=rounds, span_us=round(float(t[:, 4].max()), 1),
               prologue_us=round(med(t[:, 1] - t[:, 0]), 2), kloop_us=round(med(t[:, 2] - t[:, 1]), 2),
               epilogue_issue_us=round(med(t[:, 3] - t[:, 2]), 2), store_ack_us=round(med(t[:, 4] - t[:, 3]), 2),
               first_entry_spread_us=round(float(t[:, 0].kthvalue(min(256, ntile)).values), 2),
               gap_end_to_next_entry_us=(round(float(torch.tensor(gaps_issue).median()), 2) if gaps_issue else None),
               gap_ack_to_next_entry_us=(round(float(torch.tensor(gaps_ack).median()), 2) if gaps_ack else None))
    print(json.dumps(rec), flush=True)
