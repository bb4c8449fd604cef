"""dfx_adam_apply (x, grad, exp_avg, exp_avg_sq: four arrays read, three written) at C2 size with its four vectors in
chosen memory classes.  python tools/tune_adam_apply_placement.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from datafusion_amd import _cabi as cabi, placement  # noqa: E402

dev = "cuda:0"
n = 1 << 30
pool = placement.pool(dev)
per = 4 * n // placement.CHUNK_BYTES
ids = pool._create(110)
pool._classify(ids)
by = {}
for g in ids:
    by.setdefault(pool.cls[g], []).append(g)
print({c: len(v) for c, v in by.items()}, pool.log[-1]["class_sequence"], flush=True)
names = sorted(by, key=lambda c: -len(by[c]))[:3]
A, B, Cc = names
lib = cabi.lib()
scal = torch.zeros(cabi.NSCAL, dtype=torch.float64, device=dev)
part = torch.empty(cabi.EW_MAXBLOCKS * cabi.NPART, dtype=torch.float64, device=dev)


def vec(cls):
    ch = [by[cls].pop() for _ in range(per)]
    ptr = pool._map(ch)
    t = torch.as_tensor(placement._Raw(ptr, n), device=dev)
    return t, ptr, ch, cls


def run(label, classes, nb=0):
    vs = [vec(c) for c in classes]
    x, g, m1, m2 = (v[0] for v in vs)
    x.uniform_(0, 1), g.uniform_(0, 1e-3), m1.zero_(), m2.zero_()
    ts = []
    for it in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        cabi.check(lib.dfx_adam_apply(x.data_ptr(), g.data_ptr(), m1.data_ptr(), m2.data_ptr(), n, 1e-8, 0.9, 0.999, 1e-8,
                                      it + 1, 0, nb, scal.data_ptr(), part.data_ptr(),
                                      torch.cuda.current_stream().cuda_stream), "dfx_adam_apply")
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts = sorted(ts[1:])
    print(f"{label:28s} x,g,m1,m2 in {classes}: med {ts[3]:.3f} ms min {ts[0]:.3f} -> {28 * n / ts[3] / 1e6:.0f} GB/s "
          f"({28 * n / ts[3] / 8e9:.3f} of peak)", flush=True)
    del x, g, m1, m2
    for _, ptr, ch, cls in vs:
        pool._unmap(ptr)
        by[cls].extend(ch)


for label, classes in (("all in one class", (A, A, A, A)), ("x|g|m1|m2 = A B C A", (A, B, Cc, A)),
                       ("A B C B", (A, B, Cc, B)), ("A B C C", (A, B, Cc, Cc)), ("A A B B", (A, A, B, B)),
                       ("A B A B", (A, B, A, B)), ("A B B A", (A, B, B, A)), ("A B B C", (A, B, B, Cc)),
                       ("A A B C", (A, A, B, Cc)), ("A B A C", (A, B, A, Cc))):
    run(label, classes)
for nb in (512, 1024):
    run(f"A B C A, {nb} workgroups", (A, B, Cc, A), nb)
pool._destroy_arena()
