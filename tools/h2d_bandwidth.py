"""Host -> device copy rates on this box (dev tool): pinned and pageable sources, piece sizes, one or two copies in flight."""
import time
import torch
torch.cuda.init()
dev = torch.device("cuda:0")
for mib in (1, 4, 16, 64, 128):
    n = mib << 20
    src_p = torch.empty(n, dtype=torch.uint8).pin_memory()
    src_g = torch.empty(n, dtype=torch.uint8); src_g.fill_(1)
    dst = torch.empty(n, dtype=torch.uint8, device=dev)
    for name, src in (("pinned", src_p), ("pageable", src_g)):
        dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
        reps = max(3, 256 // mib)
        t = time.perf_counter()
        for _ in range(reps):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps
        print("%4d MiB %-8s: %.3f ms  %.1f GB/s" % (mib, name, dt * 1e3, n / dt / 1e9), flush=True)
# two streams, pinned, 64 MiB each
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
a = torch.empty(64 << 20, dtype=torch.uint8).pin_memory(); b = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
da = torch.empty(64 << 20, dtype=torch.uint8, device=dev); db = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(8):
    with torch.cuda.stream(s1): da.copy_(a, non_blocking=True)
    with torch.cuda.stream(s2): db.copy_(b, non_blocking=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t
print("two streams, 2 x 64 MiB pinned x 8: %.1f GB/s" % (16 * (64 << 20) / dt / 1e9))
# host memcpy rate into pinned memory (one thread)
import numpy as np
x = np.ones(128 << 20, dtype=np.uint8); y = a.numpy()
t = time.perf_counter()
for _ in range(4): y[:] = x[: 64 << 20]
print("numpy copy into pinned, one thread: %.1f GB/s" % (4 * (64 << 20) / (time.perf_counter() - t) / 1e9))
