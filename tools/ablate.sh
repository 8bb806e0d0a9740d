for a in 0 1 2 4 3 5 6 7; do echo "== ablate $a"; HMA_GEMM_ABLATE=$a timeout 120 python tools/gemm_bench.py 2>&1 | grep -E "fwd qkvt|fwd fc1|fwd fc2|ref  plain bf16->f32|fwd proj bf16->resid  "; done
