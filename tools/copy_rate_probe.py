import time, torch, numpy as np
n = 128 << 20
host_pageable = np.ones(n, np.uint8)
pinned = torch.empty(n, dtype=torch.uint8).pin_memory()
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
tp = torch.from_numpy(host_pageable)
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return n / best / 1e9
print("H2D pageable %.1f GB/s  pinned %.1f GB/s" % (t(lambda: dev.copy_(tp)), t(lambda: dev.copy_(pinned, non_blocking=True))))
print("D2H pageable %.1f GB/s  pinned %.1f GB/s" % (t(lambda: tp.copy_(dev)), t(lambda: pinned.copy_(dev, non_blocking=True))))
t0 = time.perf_counter(); pinned.numpy()[:] = host_pageable; dt = time.perf_counter() - t0
print("host memcpy 1 thread %.1f GB/s" % (n / dt / 1e9))
