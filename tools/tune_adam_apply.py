"""Time dfx_adam_apply (DfxAdam.step) at C2 size over block counts.  python tools/tune_adam_apply.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datafusion_amd import DfxAdam
n = 1 << 30
p = torch.nn.Parameter(torch.rand(n, device="cuda:0"))
p.grad = torch.rand(n, device="cuda:0") * 1e-3
for nb in (0, 256, 512, 1024, 2048):
    o = DfxAdam([p], lr=1e-8, n_blocks=nb)
    o.step(); torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); o.step(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f"n_blocks={nb}: med {ts[3]:.3f} ms min {ts[0]:.3f} -> {28 * n / ts[3] / 1e6:.0f} GB/s", flush=True)
    del o
