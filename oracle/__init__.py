"""TEST INFRASTRUCTURE ONLY.

`oracle/` is a CPU restatement (plain PyTorch fp32) of the reference hot path
`hma/model/*` + the AdamW/clip step of `hma/train_multi.py`.  It exists so the
HIP path in `hma_amd/` can be checked; it is never imported by the product.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this package.  Parity pin: `tests/golden/*.safetensors` were produced by
importing the real reference in the build container
(`tests/golden/make_golden.py`); `tests/test_oracle_golden.py` checks this
restatement against every one of them.
"""
