import torch, time
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3
for mb in (268, 537, 2048):
    n = mb * 1024 * 1024 // 2
    x = torch.empty(n, dtype=torch.bfloat16, device="cuda"); y = torch.empty_like(x)
    tw = t(lambda: x.zero_()); tc = t(lambda: y.copy_(x)); tr = t(lambda: x.view(torch.int16).max()) if False else 0
    s = t(lambda: torch.sum(x.view(torch.int32)))
    print(f"{mb:5d} MB: fill {mb/1024/1024*1048576/tw/1e6:.2f} TB/s   copy (r+w) {2*mb*1048576/tc/1e12:.2f} TB/s   read(sum) {mb*1048576/s/1e12:.2f} TB/s")
