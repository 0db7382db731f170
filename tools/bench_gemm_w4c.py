import sys
sys.argv=[sys.argv[0]]
exec(open('tools/bench_gemm_w4b.py').read().split("ALL = (")[0])
ALL = ("pp", "dma0", "dma2")
for _ in range(2):
    run("NT", 8192, 2048, 8192, False, False, ALL)
    run("NT sq", 8192, 8192, 8192, False, False, ALL)
    run("NT up", 8192, 8192, 2048, False, False, ALL)
