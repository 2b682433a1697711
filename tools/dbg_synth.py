import sys, torch
sys.path.insert(0, '.')
from hypergrep_amd import benchspec, device
patterns, needles, hpm = benchspec.c3_spec()
n = 1 << 20
t = torch.empty(n + 32, dtype=torch.uint8, device='cuda:0')
device.synth_device(t.data_ptr(), n, 7, needles, hpm * 5)
d = bytes(t[:n].cpu().numpy()); h = device.synth_host(n, 7, needles, hpm * 5)
diffs = [i for i in range(n) if d[i] != h[i]]
print(len(diffs), diffs[:20])
for i in diffs[:3]:
    print(repr(d[i-80:i+40])); print(repr(h[i-80:i+40]))
