#!/usr/bin/env python3
"""hipMemsetAsync recorded into a HIP graph: correct on the first replay, garbage on the second (ROCm 7.0.2 on MI355X) -- why every path of the
library that a caller may capture zeroes with a kernel of its own (bft_zero_async, csrc/bft_dev.h).  Prints the byte sums after capture and after two
replays of a graph that holds one memset of a buffer filled with ones in between."""
import ctypes as C, torch
hip = C.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
dev = torch.device("cuda", 0)
for nbytes in (64, 560, 4096, 1 << 20):
    x = torch.ones(nbytes, dtype=torch.uint8, device=dev)
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        rc = hip.hipMemsetAsync(x.data_ptr(), 0, nbytes, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    after_capture = int(x.sum())
    x.fill_(1); torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    r1 = int(x.sum())
    x.fill_(1); torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    print(nbytes, "rc", rc, "sum after capture (nbytes = not executed eagerly)", after_capture, "after replay 1", r1, "after replay 2", int(x.sum()))
