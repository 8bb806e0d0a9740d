"""Reference points from the vendor libraries torch ships (hipBLASLt / rocBLAS GEMMs, the SDPA flash kernels) at this
path's shapes: how far the hand-written kernels are from a tuned plain kernel.  Measurement tooling only -- nothing in
hma_amd/ calls these."""
import torch, time, sys

def bench(f, n=20, w=5):
    for _ in range(w):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us

M = 163840
dev = "cuda"
print("NT GEMMs  y[M,N] = x[M,K] w[N,K]^T  (bf16 in / bf16 out)")
for K, N in ((256, 768), (256, 256), (256, 1024), (768, 256), (1024, 256)):
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    us = bench(lambda: torch.nn.functional.linear(x, w))
    print(f"  K={K:5d} N={N:5d}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.0f} TFLOP/s  {(M * K + M * N) * 2 / us / 1e6:6.2f} TB/s")
print("TN GEMMs  dW[N,K] = dy[M,N]^T x[M,K]")
for K, N in ((256, 768), (256, 256), (256, 1024), (1024, 256)):
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    us = bench(lambda: dy.t() @ x)
    print(f"  K={K:5d} N={N:5d}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.0f} TFLOP/s  {(M * K + M * N) * 2 / us / 1e6:6.2f} TB/s")
print("SDPA spatial: 512 frames x 8 heads x 320 tokens x 32")
q = torch.randn(512, 8, 320, 32, device=dev, dtype=torch.bfloat16, requires_grad=True)
k = torch.randn_like(q, requires_grad=True)
v = torch.randn_like(q, requires_grad=True)
f = lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v)
us = bench(f)
print(f"  fwd {us:8.1f} us")
o = f()
do = torch.randn_like(o)
def fb():
    o = torch.nn.functional.scaled_dot_product_attention(q, k, v)
    o.backward(do)
us2 = bench(fb)
print(f"  fwd+bwd {us2:8.1f} us (bwd ~ {us2 - us:.1f})")
x = torch.randn(M, 256, device=dev)
print(f"  copy fp32 [M,256] -> {bench(lambda: x.clone()):.1f} us ({M * 256 * 8 / bench(lambda: x.clone()) / 1e6:.2f} TB/s)")
