"""HBM efficiency vs access granularity (measurement tooling): writing / reading a [M, 768] bf16 matrix in per-row pieces of
64 / 128 / 512 / 1536 bytes, one launch per piece offset -- the access shape of a kernel that produces a row's columns over
several steps (the fused blocks) against one that streams whole rows (the GEMM epilogues)."""
import torch

M = 163840
dev = "cuda"
dst = torch.zeros(M, 768, dtype=torch.bfloat16, device=dev)
x32 = torch.zeros(M, 256, dtype=torch.float32, device=dev)


def bench(f, n=5):
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for piece in (32, 64, 256, 768):  # columns (bf16): 64 B, 128 B, 512 B, 1536 B per row
    src = torch.randn(M, piece, device=dev).to(torch.bfloat16)
    npc = 768 // piece

    def wr():
        for i in range(npc):
            dst[:, i * piece:(i + 1) * piece].copy_(src)

    def rd():
        for i in range(npc):
            src.copy_(dst[:, i * piece:(i + 1) * piece])

    tw, tr = bench(wr), bench(rd)
    mb = M * 768 * 2 / 1e6
    print(f"piece {piece * 2:5d} B/row x {npc:2d} launches: write {tw:7.1f} us ({mb / tw:5.2f} TB/s of useful bytes; x2 with the source read)   "
          f"read {tr:7.1f} us ({mb / tr:5.2f} TB/s)")
for piece in (32, 64, 256):  # fp32 residual: 128 B, 256 B, 1024 B per row
    src = torch.randn(M, piece, device=dev)
    npc = 256 // piece

    def rd():
        for i in range(npc):
            src.copy_(x32[:, i * piece:(i + 1) * piece])

    tr = bench(rd)
    mb = M * 256 * 4 / 1e6
    print(f"fp32 piece {piece * 4:5d} B/row x {npc:2d}: read {tr:7.1f} us ({mb / tr:5.2f} TB/s of useful bytes)")
