import torch, time
dev = torch.device("cuda")
n = 65536 * 201600
x = torch.empty(n, dtype=torch.uint8, device=dev)
for fn, name in ((lambda: x.zero_(), "zero_ u8"), (lambda: x.view(torch.int32).fill_(7), "fill_ i32"), (lambda: x.view(torch.float32).fill_(1.5), "fill_ f32")):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(20): fn()
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 20
    print(name, f"{ms*1e3:.0f} us  {n/ms/1e9:.2f} TB/s")
y = torch.empty_like(x)
for _ in range(3): y.copy_(x)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(10): y.copy_(x)
ev[1].record(); torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / 10
print("copy", f"{ms*1e3:.0f} us  r+w {2*n/ms/1e9:.2f} TB/s")
