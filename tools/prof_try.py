import torch, time
from torch.profiler import profile, ProfilerActivity
x = torch.randn(4096, 4096, device="cuda")
g = torch.cuda.CUDAGraph()
y = x @ x
torch.cuda.synchronize()
with torch.cuda.graph(g):
    for _ in range(3):
        y = x @ x
        y = torch.relu(y)
g.replay(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(4):
        g.replay()
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
print(len(ev))
from collections import defaultdict
agg = defaultdict(lambda: [0, 0.0])
for e in ev:
    agg[e.name][0] += 1; agg[e.name][1] += e.device_time if hasattr(e, "device_time") else e.cuda_time
for k, v in agg.items():
    print(k[:80], v)
