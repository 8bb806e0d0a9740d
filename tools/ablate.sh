for a in 0 1 3; do echo "== ablate $a"; HMA_GEMM_ABLATE=$a timeout 120 python tools/gemm_bench.py 2>&1 | grep -E "fwd|bwd|ref"; done
