"""How eager multiplies a bf16 tensor by a 0-dim fp32 tensor: on the device (both operand
orders) the 0-dim operand is cast to bf16 first; CPU eager does so only when it comes first."""
import torch

x = torch.randn(100000).bfloat16()
s = torch.tensor(0.207014262676239)
r_bf = (x.float() * s.bfloat16().float()).bfloat16()
r_32 = (x.float() * s).bfloat16()
for dev in ("cpu", "cuda"):
    xd, sd = x.to(dev), s.to(dev)
    for name, y in (("x*s", xd * sd), ("s*x", sd * xd)):
        y = y.cpu()
        print(dev, name, "== bf16-rounded factor:", bool((y == r_bf).all()),
              " == fp32 factor:", bool((y == r_32).all()))
    if dev == "cuda":
        y = (xd * s).cpu()   # CPU 0-dim operand with a device tensor
        print("cuda x*s_cpu == bf16-rounded:", bool((y == r_bf).all()), " == fp32:", bool((y == r_32).all()))
