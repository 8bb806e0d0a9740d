#!/usr/bin/env python3
"""Timing of a block's weight-gradient launch as the engine issues it at the bench size (hma_gemm_tn_multi: the seven problems of
engine.py's backward plan with their operand kinds -- fc2's gelu(u) and fc1's dU in the fused MLP's fragment order, the spatial
dqkv head-blocked, the two folded LayerNorm affines).  Operands rotate over NB copies so the 256 MB MALL does not serve them.

  python tools/tn_multi_bench.py                       shipped library
  HMA_LIB=hma_amd/libhma_hip_prof.so HMA_GEMM_TN_ABLATE=2 python tools/tn_multi_bench.py     debug build: ablation bits
      (1 no DMA, 2 no MFMA, 4 no LDS reads, 8 no partial stores, 16 no barrier); the debug build also prints the phase timers of
      the leading wave half (cycles per stage: wait for the stage's pieces | barrier | fragment reads | MFMA issue | DMA issue).
  TN_SET=seven|six|mlp|attn|nolin|...  which problems go into the launch (comma list of fc2,fc1,proj_t,qkv_t,lin,proj_s,qkv_s)
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hma_amd import _lib

if os.environ.get("HMA_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["HMA_LIB"])
from hma_amd import ops
from hma_amd._lib import A_BF16, A_BF16_AFFINE, A_BF16_FRAG32, A_BF16_HEADBLK

dev = "cuda"
M = int(os.environ.get("GEMM_M", 163840))
SA = int(os.environ.get("TN_SA", 320))
NB = int(os.environ.get("TN_NB", 2))
WS = torch.empty(256 * (65536 + 256), device=dev)


def prob(N, K, bias, affine=False, y_kind=A_BF16, a_kind=A_BF16, y_group=(0, 0)):
    ps = []
    dW = torch.zeros(N, K, device=dev)
    db = torch.zeros(N, device=dev) if bias else None
    gam = torch.randn(K, device=dev) if affine else None
    bet = torch.randn(K, device=dev) if affine else None
    wm = torch.randn(N, K, device=dev) if affine else None
    dg = torch.zeros(K, device=dev) if affine else None
    dbt = torch.zeros(K, device=dev) if affine else None
    for _ in range(NB):
        dy = torch.randn(M, N, device=dev).bfloat16()
        x = torch.randn(M, K, device=dev).bfloat16()
        g = ops.make_gemm_tn(dY=ops.ptr(dy), ldy=N, y_kind=y_kind, y_group=y_group, A=ops.ptr(x), lda=K,
                             a_kind=A_BF16_AFFINE if affine else a_kind, M=M, N=N, K=K, dW=ops.ptr(dW), lddw=K, dBias=ops.ptr(db),
                             gamma=ops.ptr(gam), beta=ops.ptr(bet), ws=ops.ptr(WS), ws_elems=WS.numel(), w_master=ops.ptr(wm),
                             dgamma=ops.ptr(dg), dbeta=ops.ptr(dbt))
        ps.append((g, dy, x))
    return dict(ps=ps, keep=(dW, db, gam, bet, wm, dg, dbt), mb=(M * N + M * K) * 2 / 1e6, fl=2.0 * M * N * K)


ALL = {
    "fc2": lambda: prob(256, 1024, True, a_kind=A_BF16_FRAG32),
    "fc1": lambda: prob(1024, 256, True, affine=True, y_kind=A_BF16_FRAG32),
    "proj_t": lambda: prob(256, 256, True),
    "qkv_t": lambda: prob(768, 256, False),
    "lin": lambda: prob(256, 256, True),
    "proj_s": lambda: prob(256, 256, True),
    "qkv_s": lambda: prob(768, 256, False, affine=True, y_kind=A_BF16_HEADBLK, y_group=(SA, 0)),
}
SETS = {"seven": list(ALL), "six": [k for k in ALL if k != "lin"], "mlp": ["fc2", "fc1"], "attn": ["proj_t", "qkv_t", "proj_s", "qkv_s"]}


def run(names):
    probs = [ALL[n]() for n in names]
    arrs = []
    for i in range(NB):
        gs = [p["ps"][i][0] for p in probs]
        arrs.append(((C.POINTER(type(gs[0])) * len(gs))(*[C.pointer(g) for g in gs]), gs))

    def fn(i):
        _lib.call("hma_gemm_tn_multi", ops.stream_ptr(), arrs[i % NB][0], len(names))

    for i in range(8):
        fn(i)
    torch.cuda.synchronize()
    lib = _lib.load()
    has_prof = hasattr(lib, "hma_debug_prof")
    if has_prof:
        try:
            lib.hma_debug_prof.argtypes = [C.POINTER(C.c_ulonglong)]
            lib.hma_debug_prof.restype = C.c_int
            out = (C.c_ulonglong * 16)()
            lib.hma_debug_prof(out)  # clear
        except AttributeError:
            has_prof = False
    reps = int(os.environ.get("TN_REPS", 40))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    mb = sum(p["mb"] for p in probs)
    fl = sum(p["fl"] for p in probs)
    print(f"multi[{','.join(names)}] abl={os.environ.get('HMA_GEMM_TN_ABLATE', '0')}  {us:8.1f} us (ring + reduction)  {mb:6.0f} MB operands  "
          f"{mb / us:6.2f} TB/s {fl / us / 1e6:7.1f} TFLOP/s", flush=True)
    if has_prof:
        out = (C.c_ulonglong * 16)()
        if lib.hma_debug_prof(out) == 0 and out[8]:
            n = out[8]
            names_p = ["wait pieces", "barrier", "frag reads", "mfma issue", "dma issue", "tail"]
            tot = sum(out[i] for i in range(6))
            print("   phase cycles per workgroup (leading half, wave 0): " +
                  "  ".join(f"{names_p[i]} {out[i] / n:9.0f} ({100.0 * out[i] / max(tot, 1):4.1f} %)" for i in range(6)) +
                  f"   total {tot / n:9.0f} over {n} workgroups", flush=True)


_w = torch.randn(8192, 8192, device=dev)
for _ in range(20):
    _w @ _w  # clocks up before the first shape
torch.cuda.synchronize()
for s in (os.environ.get("TN_SET") or "seven").split(";"):
    run(SETS.get(s, s.split(",")))
