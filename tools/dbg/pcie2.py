import torch, time
n = 8 << 20
h = torch.empty(64 * n, dtype=torch.uint8).pin_memory()
d = torch.empty(64 * n, dtype=torch.uint8, device="cuda")
s = torch.cuda.Stream()
torch.cuda.synchronize()
for rep in range(2):
    ts = []
    t00 = time.perf_counter()
    with torch.cuda.stream(s):
        for i in range(64):
            t0 = time.perf_counter()
            d[i * n:(i + 1) * n].copy_(h[i * n:(i + 1) * n], non_blocking=True)
            ts.append(time.perf_counter() - t0)
    t_issue = time.perf_counter() - t00
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t00
    print("8 MB copies: call %.1f us median, issue of 64: %.2f ms, all done: %.2f ms (%.1f GB/s)" % (sorted(ts)[32] * 1e6, t_issue * 1e3, t_all * 1e3, 64 * n / t_all / 1e9), flush=True)
