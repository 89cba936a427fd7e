import torch, time
n = 3840*2176*4
g = torch.zeros(n, dtype=torch.float32, device="cuda")
for pin in (False, True):
    h = torch.empty(n, dtype=torch.float32, pin_memory=pin)
    h.copy_(g); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): h.copy_(g); torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    print("pinned" if pin else "pageable", round(dt*1e3, 2), "ms", round(n*4/dt/1e9, 1), "GB/s")
