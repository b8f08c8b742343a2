import torch, time
dev = torch.device("cuda", 0)
n = 1_000_000_000
a = torch.zeros(n, dtype=torch.uint8, device=dev)
b = torch.ones(n, dtype=torch.uint8, device=dev)
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print("fill 1GB ms", t(lambda: a.fill_(3)))
print("copy 1GB ms", t(lambda: a.copy_(b)))
a4 = a.view(torch.int32); 
idx = torch.randint(0, 300000, (4_000_000,), device=dev)
table = torch.randint(0, 255, (300000, 252), dtype=torch.uint8, device=dev)
out = torch.empty((4_000_000, 252), dtype=torch.uint8, device=dev)
print("index_select 4M x 252B ms", t(lambda: torch.index_select(table, 0, idx, out=out)))
