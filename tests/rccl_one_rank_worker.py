"""Rank process of tests/test_zx_rccl_one_rank_gpu.py: ONE rank, backend "nccl" (= RCCL on ROCm).
A one-GPU box cannot run a real multi-rank RCCL job (RCCL refuses two ranks on one device), but a
world of one still goes through everything the multi-rank path depends on before the first byte
moves between GPUs: the library loads, the communicator initialises, device tensors (not host
ones) are what the collectives carry, object broadcasts are staged through the current device,
and the package's exchange helpers accept the backend.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "sglang-diffusion_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from sglang_diffusion_amd.runtime.denoise import Req  # noqa: E402
from sglang_diffusion_amd.runtime.parallel import (_wire, all_gather_tensors, init_distributed,  # noqa: E402
                                                   scatter_requests, ulysses_all_to_all)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29617")
    os.environ["RANK"], os.environ["LOCAL_RANK"], os.environ["WORLD_SIZE"] = "0", "0", "1"
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    rank, world, local = init_distributed("nccl")
    dev = torch.device("cuda", 0)
    res = {"backend": dist.get_backend(), "world": world}
    dist.barrier()
    x = torch.arange(1024, device=dev, dtype=torch.float32)
    dist.all_reduce(x)
    res["all_reduce_exact"] = bool(torch.equal(x.cpu(), torch.arange(1024, dtype=torch.float32)))
    box = [{"a": 1, "t": (3, 4)}]
    dist.broadcast_object_list(box, src=0)
    res["object_broadcast"] = box[0] == {"a": 1, "t": (3, 4)}
    y = torch.randn(2, 33, 8, 128, device=dev).bfloat16()
    parts = all_gather_tensors(y)
    res["all_gather_exact"] = len(parts) == 1 and bool(torch.equal(parts[0], y))
    z = ulysses_all_to_all(y, dist.group.WORLD, scatter_dim=2, gather_dim=1)
    res["all_to_all_exact"] = bool(torch.equal(z, y))
    recv = torch.empty_like(y)
    dist.all_to_all_single(recv, y.contiguous())
    res["all_to_all_single_exact"] = bool(torch.equal(recv, y))
    host = torch.randn(5, 7)
    res["host_tensors_are_staged_on_the_gpu"] = _wire(host).is_cuda
    reqs = [Req(latents=torch.randn(1, 16, 64, device=dev).bfloat16(),
                prompt_embeds=torch.randn(1, 8, 32, device=dev).bfloat16(),
                pooled_embeds=torch.randn(1, 16, device=dev).bfloat16(), num_inference_steps=2)
            for _ in range(3)]
    meta, mine = scatter_requests(reqs, lambda costs: [list(range(len(costs)))])
    res["scatter_keeps_everything_on_the_one_rank"] = sorted(mine) == [0, 1, 2] and all(m["rank"] == 0 for m in meta)
    g2 = dist.new_group([0])
    t = torch.ones(4, device=dev)
    dist.all_reduce(t, group=g2)
    res["subgroup_all_reduce_exact"] = bool(torch.equal(t.cpu(), torch.ones(4)))
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
