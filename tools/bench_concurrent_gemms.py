"""Do two independent big GEMMs of a gated block's backward finish sooner on two streams than back to back?  (DESIGN §8 lead 1:
the K = 2048 launches' epilogues are HBM bursts in lock-step; a second kernel's main loops could fill those windows.)
Pairs: (dgelu_dot NN 8192x8192x2048, dW2 TN 2048x8192x8192) and (du NN 8192x2048x8192, dW1 TN 8192x2048x8192).  PROFILING TOOL."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from open_flamingo_amd.hip import abi
from open_flamingo_amd.hip.ops import Ops
from tools.bench_gemm_ab import make


def main():
    ops = Ops.default()
    E = abi
    pairs = {"dgelu_dot || dW2": ((8192, 8192, 2048, 0, 1, E.EPI_DGELU_DOT), (2048, 8192, 8192, 1, 1, E.EPI_ACC_F32)),
             "du || dW1": ((8192, 2048, 8192, 0, 1, E.EPI_STORE_BF16), (8192, 2048, 8192, 1, 1, E.EPI_ACC_F32)),
             "ffn_up+gelu || to_q (fwd, independent inputs)": ((8192, 8192, 2048, 0, 0, E.EPI_GELU), (8192, 512, 2048, 0, 0, E.EPI_STORE_BF16))}
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for name, (ga, gb) in pairs.items():
        fa_args, fb_args = make(*ga), make(*gb)
        # separate Ops instances would share the grow-only workspaces: give the second its own
        ops2 = Ops(ops.lib, ops._stream_fn)

        def fa():
            A, B, C, kw = fa_args
            ops.gemm(A, B, C, ta=bool(ga[3]), tb=bool(ga[4]), epi=ga[5], **kw)

        def fb():
            A, B, C, kw = fb_args
            ops2.gemm(A, B, C, ta=bool(gb[3]), tb=bool(gb[4]), epi=gb[5], **kw)

        def seq():
            fa()
            fb()

        def conc():
            cur = torch.cuda.current_stream()
            s1.wait_stream(cur)
            s2.wait_stream(cur)
            with torch.cuda.stream(s1):
                fa()
            with torch.cuda.stream(s2):
                fb()
            cur.wait_stream(s1)
            cur.wait_stream(s2)

        res = {}
        for label, fn in (("a", fa), ("b", fb), ("sequential", seq), ("two_streams", conc)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10)
            res[label + "_us"] = round(best * 1e3, 1)
        print(json.dumps(dict(pair=name, **res)), flush=True)


if __name__ == "__main__":
    main()
