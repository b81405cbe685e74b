import torch, time
n = 256 << 20
h = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(4)]
d = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(4)]
def run(ns, chunk=None):
    ss = [torch.cuda.Stream() for _ in range(ns)]
    torch.cuda.synchronize()
    t = time.time()
    for rep in range(4):
        for i in range(4):
            with torch.cuda.stream(ss[i % ns]):
                if chunk is None:
                    d[i].copy_(h[i], non_blocking=True)
                else:
                    for o in range(0, n, chunk):
                        d[i][o:o + chunk].copy_(h[i][o:o + chunk], non_blocking=True)
    torch.cuda.synchronize()
    dt = time.time() - t
    return 16 * n / dt / 1e9
for ns in (1, 2, 4):
    print("streams", ns, "whole: %.1f GB/s" % run(ns), " 8MB chunks: %.1f GB/s" % run(ns, 8 << 20), " 1MB chunks: %.1f" % run(ns, 1 << 20), flush=True)
