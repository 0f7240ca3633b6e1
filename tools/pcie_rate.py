import torch, time
x = torch.empty(65536*56448, dtype=torch.uint8, device="cuda")
h = torch.empty_like(x, device="cpu").pin_memory()
for _ in range(2): h.copy_(x, non_blocking=True); torch.cuda.synchronize()
t=time.perf_counter()
for _ in range(5): h.copy_(x, non_blocking=True)
torch.cuda.synchronize()
dt=(time.perf_counter()-t)/5
print("D2H pinned %.1f GB/s, %.1f ms for the fused obs of 65536 envs -> %.2f M env-steps/s if every step's obs goes to the host" % (x.numel()/dt/1e9, dt*1e3, 65536/dt/1e6))
