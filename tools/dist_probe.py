"""Which load order lets torch (bundled ROCm) and the engine (system ROCm) share
one process?  Usage: python tools/dist_probe.py torch_first|engine_first"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
order = sys.argv[1]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
import libh264_amd
def torch_up():
    import torch, torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl")
    t = torch.ones(4, device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    return dist
if order == "torch_first":
    d = torch_up()
    eng = libh264_amd.Engine(0)
elif order == "engine_first":
    libh264_amd.engine_lib()
    eng = libh264_amd.Engine(0)
    d = torch_up()
else:
    libh264_amd.engine_lib()
    d = torch_up()
    eng = libh264_amd.Engine(0)
import numpy as np
b = eng.upload(np.arange(64, dtype=np.uint8))
print(order, "OK", eng.download(b)[:4])
maps = open("/proc/self/maps").read()
print(sorted({l.split()[-1] for l in maps.splitlines() if "libamdhip64" in l or "libhsa-runtime" in l}))
d.destroy_process_group()
