import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, hypergrep_amd
from hypergrep_amd import benchspec, device
_, needles, hpm = benchspec.c3_spec()
text = torch.empty((1 << 20) + 64, dtype=torch.uint8, device="cuda:0")
device.synth_device(text.data_ptr(), 1 << 20, benchspec.SEED_BASE + 9, needles, hpm)
host = text[: 1 << 20].cpu().numpy()
cut = 1 << 20
while host[cut - 1] != 10: cut -= 1
path = "/dev/shm/hg_small.log"
open(path, "wb").write(host[:cut].tobytes())
for pats in (["needle_in_haystack"], ["ERROR", "status=5[0-9]{2}", "timeout after [0-9]+ ms"]):
    for i in range(6):
        if i == 4: os.environ["HYPERGREP_TRACE"] = "1"
        t0 = time.perf_counter(); c, rc = hypergrep_amd.grep(path, pats, count_only=True); t = time.perf_counter() - t0
        print(pats[0], i, rc, c, f"{t*1e3:.3f} ms", flush=True)
    os.environ.pop("HYPERGREP_TRACE", None)
os.unlink(path)
