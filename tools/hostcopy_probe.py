import sys, time, torch, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from datafusion_amd import hostcopy
x = torch.rand(256, 16, 16, 128, 128, device="cuda")
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); a = x.cpu().numpy(); t1 = time.perf_counter()
    print(f"x.cpu().numpy(): {t1 - t0:.3f} s = {x.numel() * 4 / (t1 - t0) / 1e9:.1f} GB/s", flush=True)
    del a
for chunk in (16, 64, 256):
    hostcopy._CHUNK_BYTES = chunk << 20
    for rep in range(3):
        t0 = time.perf_counter(); b = hostcopy.to_numpy(x); t1 = time.perf_counter()
        print(f"hostcopy.to_numpy chunk {chunk} MiB: {t1 - t0:.3f} s = {x.numel() * 4 / (t1 - t0) / 1e9:.1f} GB/s", flush=True)
        if rep == 0:
            assert np.array_equal(b, x.cpu().numpy())
        del b
print("threads", torch.get_num_threads())
