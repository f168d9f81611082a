import torch, time, numpy as np
x = torch.zeros(15*20000, dtype=torch.float64, device="cuda")
h = torch.empty(15*20000, dtype=torch.float64)
hp = torch.empty(15*20000, dtype=torch.float64).pin_memory()
for name, dst in (("pageable", h), ("pinned", hp)):
    for _ in range(3): dst.copy_(x); torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(50): dst.copy_(x); torch.cuda.synchronize()
    print(name, (time.perf_counter()-t)/50*1e3, "ms")
a = np.empty(15*20000); b = hp.numpy()
t=time.perf_counter()
for _ in range(50): a[:] = b
print("host memcpy", (time.perf_counter()-t)/50*1e3, "ms")
