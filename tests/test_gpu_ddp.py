"""BASELINE config 5 is a DDP training step (configs/raft-train1-chairs.yaml: Lightning's ddp strategy).  The mirror's training
forward is a plain nn.Module built from custom autograd nodes on torch's current stream; this test wraps it in
torch.nn.parallel.DistributedDataParallel over RCCL — one rank, all a single-GPU box can hold; the reducer, its gradient-ready hooks
and bucket all-reduce run as they do for N ranks — and checks that one step gives the gradients of the unwrapped model bit for bit
(an all-reduce over one rank is the identity, and DDP must not change the graph)."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_ddp_training_step_matches_plain_model(gpu):
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from ptlflow_amd.raft import RAFT
    from ptlflow_amd.train import sequence_loss
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 2, 3, 128, 160, generator=g).to(gpu)
    gt = (torch.randn(2, 2, 128, 160, generator=g) * 3).to(gpu)
    valid = torch.ones(2, 1, 128, 160, device=gpu)

    def step(model):
        model.zero_grad(set_to_none=True)
        out = model({"images": x})
        loss = sequence_loss(out["flow_preds"], gt, valid)
        loss.backward()
        return loss.detach()

    plain = RAFT(iters=3).load_synthetic(17).to(gpu).train()
    loss_plain = step(plain)
    grads = {n: p.grad.clone() for n, p in plain.named_parameters()}

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(gpu))     # "nccl" is RCCL on ROCm
    try:
        twin = RAFT(iters=3).load_synthetic(17).to(gpu).train()
        ddp = DDP(twin, device_ids=[torch.device(gpu).index or 0])
        loss_ddp = step(ddp)
        torch.cuda.synchronize()
        assert torch.equal(loss_ddp, loss_plain)
        for n, p in twin.named_parameters():
            assert p.grad is not None, f"{n}: DDP left no gradient"
            assert torch.equal(p.grad, grads[n]), f"{n}: gradient differs under DDP"
        # BatchNorm buffers of cnet (running statistics) are updated by the forward and broadcast by DDP: still finite and equal
        for (n, b), (_, b0) in zip(twin.named_buffers(), plain.named_buffers()):
            assert torch.equal(b, b0), n
    finally:
        dist.destroy_process_group()
