import torch, time
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
n = 1 << 28  # 1 GiB fp32
a = torch.empty(n, device="cuda"); b = torch.empty(n, device="cuda")
s = t(lambda: a.fill_(1.0)); print(f"fill  (write 1 GiB): {4*n/s/1e12:.2f} TB/s")
s = t(lambda: b.copy_(a)); print(f"copy  (r+w 2 GiB)  : {8*n/s/1e12:.2f} TB/s")
s = t(lambda: a.sum()); print(f"sum   (read 1 GiB) : {4*n/s/1e12:.2f} TB/s")
s = t(lambda: a.add_(1.0)); print(f"add_  (r+w in place): {8*n/s/1e12:.2f} TB/s")
c = torch.empty(n // 2, device="cuda", dtype=torch.bfloat16)
s = t(lambda: c.copy_(a[: n // 2])); print(f"cast f32->bf16 (r 1 GiB*0.5 + w 0.25): {(4*(n//2)+2*(n//2))/s/1e12:.2f} TB/s")
