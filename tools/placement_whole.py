"""Whole-GPU volumes (config 3: 128 GiB per vector): does it matter whether x and p are two allocations or the two
halves of one?   python tools/placement_whole.py arena|separate|arena_px [shape]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datafusion_amd import _cabi as cabi
from datafusion_amd.kernels import HipKernels
mode = sys.argv[1]
shape = tuple(int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "512,32,32,256,256").split(","))
small = torch.rand(64 << 20, device="cuda:0")          # the small tensors a solver allocates first (256 MiB)
k = HipKernels(*shape, 4, "cuda:0")
k.new_factored_residual(); k.ra.zero_(); k.rb.zero_()
if mode == "separate":
    x = torch.zeros(k.shape, device="cuda:0"); p = torch.zeros(k.shape, device="cuda:0")
else:
    arena = torch.zeros(2 * k.N, device="cuda:0")
    a, b = arena[:k.N].view(k.shape), arena[k.N:].view(k.shape)
    x, p = (a, b) if mode == "arena" else (b, a)
best = 1e9
for _ in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); k.sweep(cabi.SWEEP_XP, x, None, p, factored=True); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
print(f"{mode} {shape}: sweep {best:.3f} ms = {16 * k.N / best / 1e6:.0f} GB/s  x@{x.data_ptr():x} p@{p.data_ptr():x}", flush=True)
