"""N > 1 ranks without a second GPU: two processes share cuda:0 (RCCL refuses two ranks on one device, so the process group
is gloo — DDP's reducer, gradient-ready hooks and bucket all-reduce are the same code, only the transport differs).

* DDP gradient averaging through the libpfk autograd nodes (BASELINE config 5's exchange step): each rank runs one training
  step of the RAFT mirror on its OWN half of a batch under DistributedDataParallel; the averaged gradient every rank ends up
  with must equal the mean of the two plain (un-wrapped) per-rank gradients.
* bench.py --gpus 2 under torch.distributed.run with PFK_BENCH_SHARED_DEVICE=1: the N > 1 branch of the bench (rank env,
  per-rank seeds, barrier, max-over-ranks time, whole-job pairs) end to end.

Neither is a scaling measurement — the ranks time-share one chip (DESIGN §6 says so)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ddp_worker(rank, world, port, q):
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    sys.path.insert(0, ROOT)
    import ptlflow_amd
    from ptlflow_amd.raft import RAFT
    from ptlflow_amd.train import sequence_loss
    ptlflow_amd.load_native()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(100 + rank)              # every rank its own samples
        x = torch.rand(2, 2, 3, 128, 160, generator=g).to(dev)
        gt = (torch.randn(2, 2, 128, 160, generator=g) * 3).to(dev)
        valid = torch.ones(2, 1, 128, 160, device=dev)

        def step(model):
            model.zero_grad(set_to_none=True)
            loss = sequence_loss(model({"images": x})["flow_preds"], gt, valid)
            loss.backward()
            return float(loss)

        plain = RAFT(iters=3).load_synthetic(17).to(dev).train()
        loss_plain = step(plain)
        mine = [p.grad.detach().float().cpu() for p in plain.parameters()]
        twin = RAFT(iters=3).load_synthetic(17).to(dev).train()
        ddp = DDP(twin, device_ids=[0])
        loss_ddp = step(ddp)
        torch.cuda.synchronize()
        avg = [p.grad.detach().float().cpu() for p in twin.parameters()]
        # every rank's plain gradients, gathered: the expectation is their mean
        both = [None] * world
        dist.all_gather_object(both, mine)
        worst = 0.0
        for i, a in enumerate(avg):
            want = (both[0][i] + both[1][i]) / 2
            scale = max(float(want.abs().max()), 1e-12)
            worst = max(worst, float((a - want).abs().max()) / scale)
        q.put((rank, loss_plain, loss_ddp, worst, [float(a.abs().sum()) for a in avg[:4]]))
    finally:
        dist.destroy_process_group()


def test_two_rank_ddp_averages_gradients_through_pfk_nodes(gpu):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, lp0, ld0, w0, s0), (_, lp1, ld1, w1, s1) = got
    assert lp0 == ld0 and lp1 == ld1                 # DDP does not change a rank's own forward
    assert lp0 != lp1                                 # the two ranks really saw different samples
    assert s0 == s1                                   # both ranks hold the same averaged gradient
    assert max(w0, w1) <= 1e-6, f"averaged gradient differs from the mean of the per-rank gradients by {max(w0, w1):.2e} of scale"


def test_bench_two_ranks_shared_device(gpu):
    env = dict(os.environ, PFK_BENCH_SHARED_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--batch", "2", "--iters", "4", "--height", "184", "--width", "320", "--no-cpu-baseline", "--no-roofline",
           "--no-split-modes", "--no-extra-legs", "--no-batch1"]
    run = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:]                       # rank 0 prints ONE line
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["scaling"] == "weak"
    assert r["config"]["global_batch"] == 4
    # whole-job value: both ranks' pairs over the max-over-ranks time
    assert r["value"] == pytest.approx(2 * 2 * 2 / (r["ms_per_step"] * 2 / 1e3), rel=1e-6)
    assert r["streamk_faults"] == 0


_SMALL = ["--steps", "2", "--warmup", "1", "--batch", "2", "--iters", "4", "--height", "184", "--width", "320", "--no-cpu-baseline",
          "--no-roofline", "--no-split-modes", "--no-extra-legs", "--no-batch1"]


def test_bench_gpus_2_self_launches_two_ranks(gpu):
    """`python bench.py --gpus 2` with NO launcher (how a driver may invoke it): bench.py starts the two ranks itself and the
    one JSON line says n_gpus == 2, ranks_seen == 2 (an all-reduce of ones over the process group)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PFK_BENCH_SHARED_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", *_SMALL], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["ranks_seen"] == 2 and r["backend"].startswith("gloo") and r["config"]["global_batch"] == 4


def test_bench_refuses_more_ranks_than_devices(gpu):
    """Without the shared-device flag a one-GPU box must FAIL `--gpus 2` loudly, never print an `n_gpus: 1` line."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs: the refusal cannot be provoked")
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PFK_BENCH_SHARED_DEVICE")}
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", *_SMALL], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert run.returncode != 0
    assert not [l for l in run.stdout.splitlines() if l.startswith("{")], run.stdout[-1000:]
    assert "only 1 GPU" in run.stderr
