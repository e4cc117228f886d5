import sys
sys.path.insert(0, ".")
import torch
x = torch.ones(4, device="cuda"); torch.cuda.synchronize()
print("torch hip:", torch.version.hip, "device", torch.cuda.get_device_name(0))
import __graft_entry__ as g
g.smoke()
y = (x * 2).sum().item(); torch.cuda.synchronize()
print("torch + libdiskrag_hip in one process ok", y)
import os
print([l.split()[-1] for l in open(f"/proc/{os.getpid()}/maps") if "amdhip64" in l][:4])
