#!/usr/bin/env python3
"""Host-to-device copy rate of a query batch (140 MB = 2x10^7 packed k-mers of k = 27) from pageable and from pinned host memory: what a
pinned staging path in front of bft_gpu_query_presence could gain at most (MI355X box of this pool: 52.0 / 51.0 GB/s -- nothing)."""
import torch, time
n = 140_000_000
dev = torch.device("cuda", 0)
d = torch.empty(n, dtype=torch.uint8, device=dev)
for name, h in (("pageable", torch.empty(n, dtype=torch.uint8)), ("pinned", torch.empty(n, dtype=torch.uint8, pin_memory=True))):
    h.fill_(1)
    for _ in range(2):
        d.copy_(h); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        d.copy_(h, non_blocking=True); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(name, round(n / dt / 1e9, 1), "GB/s H2D")
