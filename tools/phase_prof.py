#!/usr/bin/env python3
"""Phase breakdown of the persistent NT GEMM kernels (debug build of gemm.hip with -DHMA_PROF).

Build here:   python tools/phase_prof.py --build      (writes hma_amd/libhma_hip_prof.so)
Run on GPU:   python tools/phase_prof.py [case ...]
Buffers are rotated over several copies so that consecutive launches do not hit the 256 MB MALL.
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROF_LIB = os.path.join(ROOT, "hma_amd", "libhma_hip_prof.so")

if "--build" in sys.argv:
    from hma_amd import build as B
    B.build()
    obj = os.path.join(B.HERE, "build", "gemm_prof.o")
    subprocess.run([B._hipcc(), *B.FLAGS, "-DHMA_PROF", "-c", os.path.join(B.CSRC, "gemm.hip"), "-o", obj], check=True)
    others = [os.path.join(B.HERE, "build", s.replace(".hip", ".o")) for s in B.SOURCES if s != "gemm.hip"]
    subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", PROF_LIB, obj, *others], check=True)
    print(PROF_LIB)
    sys.exit(0)

import torch
from hma_amd import _lib
_lib.LIB_PATH = PROF_LIB
from hma_amd import ops
from hma_amd._lib import *

lib = _lib.load()
lib.hma_debug_prof.argtypes = [C.POINTER(C.c_ulonglong)]
lib.hma_debug_prof.restype = C.c_int
dev = "cuda"
M = int(os.environ.get("GEMM_M", 163840))
NSET = 6
def bf(*s): return [torch.randn(*s, device=dev).bfloat16() for _ in range(NSET)]
def f32(*s): return [torch.randn(*s, device=dev) for _ in range(NSET)]
def ebf(*s): return [torch.empty(*s, device=dev, dtype=torch.bfloat16) for _ in range(NSET)]

g = torch.randn(256, device=dev); b = torch.randn(256, device=dev)
cases = {}
def case(name, fn, flops): cases[name] = (fn, flops)

xh = bf(M, 256); x32 = f32(M, 256)
w768 = torch.randn(768, 256, device=dev).bfloat16(); w256 = torch.randn(256, 256, device=dev).bfloat16()
w1024 = torch.randn(1024, 256, device=dev).bfloat16(); wk1024 = torch.randn(256, 1024, device=dev).bfloat16()
wk768 = torch.randn(256, 768, device=dev).bfloat16()
b1024 = torch.randn(1024, device=dev); b256 = torch.randn(256, device=dev)
qkv = ebf(M, 768); u = ebf(M, 1024); h = ebf(M, 1024); o256 = ebf(M, 256)
case("qkv", lambda i: ops.linear(xh[i], w768, None, epi=EPI_BF16, out=qkv[i], gamma=g, beta=b), 2.0 * M * 768 * 256)
case("qkvt", lambda i: ops.linear(xh[i], w768, None, epi=EPI_BF16, out=qkv[i]), 2.0 * M * 768 * 256)
case("fc1", lambda i: ops.linear(xh[i], w1024, b1024, epi=EPI_GELU2, out=u[i], out2=h[i], gamma=g, beta=b), 2.0 * M * 1024 * 256)
case("fc2", lambda i: ops.linear(h[i], wk1024, b256, epi=EPI_RESID, out=x32[i]), 2.0 * M * 1024 * 256)
case("dfc2", lambda i: ops.linear(x32[i], w1024, None, epi=EPI_DGELU, out=h[i], aux=u[i]), 2.0 * M * 1024 * 256)
case("dfc1", lambda i: ops.linear(u[i], wk1024, None, epi=EPI_BF16, out=o256[i]), 2.0 * M * 1024 * 256)
case("dqkv", lambda i: ops.linear(qkv[i], wk768, None, epi=EPI_RESID, out=x32[i]), 2.0 * M * 768 * 256)
case("dprj", lambda i: ops.linear(x32[i], w256, None, epi=EPI_BF16, out=o256[i]), 2.0 * M * 256 * 256)
case("modlin", lambda i: ops.linear(xh[i], w256, b256, epi=EPI_RESID, out=x32[i], out2=o256[i]), 2.0 * M * 256 * 256)

WS = torch.empty(256 * (65536 + 256), device=dev)
def tn(dy, x, N, K, bias=True, gamma=None, beta=None):
    dW = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev) if bias else None
    return lambda i: ops.linear_wgrad(dy[i], x[i], dW, db, gamma=gamma, beta=beta, ws=WS)
case("w_proj", tn(x32, xh, 256, 256), 2.0 * M * 256 * 256)
case("w_fc2", tn(x32, h, 256, 1024), 2.0 * M * 256 * 1024)
case("w_fc1", tn(u, xh, 1024, 256, gamma=g, beta=b), 2.0 * M * 256 * 1024)
case("w_qkv", tn(qkv, xh, 768, 256, bias=False), 2.0 * M * 256 * 768)
# bf16 dY as the engine issues the layer wgrads (LDS-DMA ring kernel with HMA_GEMM_TN_DMA=tr: marks are
# prologue / issue / reads+mfma / epilogue / vmcnt wait / barrier)
case("wb_proj", tn(o256, xh, 256, 256), 2.0 * M * 256 * 256)
case("wb_qkv", tn(qkv, xh, 768, 256, bias=False), 2.0 * M * 256 * 768)
case("wb_fc1", tn(u, xh, 1024, 256, gamma=g, beta=b), 2.0 * M * 256 * 1024)
case("wb_fc2", tn(o256, h, 256, 1024), 2.0 * M * 256 * 1024)

names = [a for a in sys.argv[1:] if not a.startswith("-")] or list(cases)
PH = ["loop/prologue", "issue-loads", "lds+mfma", "epilogue", "wait+lds-store", "barrier", "acc-copy", "-"]
out = (C.c_ulonglong * 16)()
for n in names:
    fn, fl = cases[n]
    for i in range(NSET): fn(i)
    torch.cuda.synchronize()
    lib.hma_debug_prof(out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3 * NSET
    e0.record()
    for r in range(reps): fn(r % NSET)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    lib.hma_debug_prof(out)
    tot = sum(out[i] for i in range(8)) or 1
    waves = out[8] / reps
    parts = "  ".join(f"{PH[i]} {100.0 * out[i] / tot:4.1f}%" for i in range(7) if out[i])
    print(f"{n:7s} {us:7.1f} us {fl / us / 1e6:6.0f} TF | waves/launch {waves:.0f}, ticks/wave {tot / max(out[8], 1):.0f} | {parts}")
