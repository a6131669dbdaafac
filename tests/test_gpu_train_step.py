"""BASELINE config 5, one end-to-end training step of the RAFT mirror: train-mode forward (`flow_preds`), sequence loss
(raft/raft.py:20-45, gamma 0.8), backward — correlation volume / pyramid / lookup, update block and convex upsampling on
libpfk autograd nodes — against float64 autograd through the CPU oracle's training forward, for EVERY parameter
(fnet, cnet, update_block).  368x496 crops (46x62 grid), as raft-train1-chairs.yaml."""
import pytest
import torch

from oracle import raft_oracle as O

pytestmark = pytest.mark.gpu


def _run(gpu, small, B, H, W, iters, tol):
    from ptlflow_amd.raft import RAFT
    from ptlflow_amd.train import sequence_loss
    model = RAFT(small=small, iters=iters).load_synthetic(21)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x = O.smooth_pair(B, H, W, seed=4)
    g = torch.Generator().manual_seed(9)
    gt = torch.randn(B, 2, H, W, generator=g) * 4
    valid = (torch.rand(B, 1, H, W, generator=g) > 0.1).float()
    gt[0, :, :8, :8] = 500.0                       # beyond max_flow: excluded by the loss
    # float64 oracle
    names = [n for n, _ in model.named_parameters()]
    P = {k: (v.double().requires_grad_(True) if k in names else (v.double() if v.is_floating_point() else v)) for k, v in sd.items()}
    preds = O.raft_forward_train(P, x.double(), iters=iters, small=small)
    loss_ref = O.sequence_loss(preds, gt.double(), valid.double())
    grads = dict(zip(names, torch.autograd.grad(loss_ref, [P[n] for n in names], allow_unused=True)))
    # libpfk
    model = model.to(gpu).train()
    out = model({"images": x.to(gpu)})
    assert len(out["flow_preds"]) == iters and tuple(out["flows"].shape) == (B, 1, 2, H, W)
    loss = sequence_loss(out["flow_preds"], gt.to(gpu), valid.to(gpu))
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) <= 1e-4 * abs(loss_ref.item())
    scale_all = max(float(v.abs().max()) for v in grads.values() if v is not None)
    worst = []
    for n, p in model.named_parameters():
        ref = grads[n]
        if ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        assert p.grad is not None, f"{n}: no gradient"
        # scale: the tensor's own, floored for tensors whose true gradient is (numerically) zero, e.g. a conv bias in front
        # of an instance / batch norm
        scale = max(float(ref.abs().max()), 1e-4 * scale_all)
        err = float((p.grad.double().cpu() - ref).abs().max())
        worst.append((err / scale, n))
    worst.sort(reverse=True)
    assert worst[0][0] <= tol, "gradient mismatch (err/scale, name): " + ", ".join(f"{e:.2e} {n}" for e, n in worst[:6])


def test_train_step_raft(gpu):
    _run(gpu, False, 2, 368, 496, 3, 5e-4)


def test_train_step_raft_small(gpu):
    _run(gpu, True, 1, 184, 248, 3, 5e-4)
