#!/usr/bin/env python3
"""bench.py — frame-pairs/s of the RAFT hot path on MI355X (BASELINE.json metric).

    python bench.py                              # 1 GPU, raft 436x1024, 32 iterations, fp32
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A *step* is one forward of `ptlflow_amd.RAFT` (both encoders, correlation volume + pyramid + 32 x {lookup, update block,
coordinate update, convex upsample} — every kernel libpfk's, no MIOpen / rocBLAS call in the forward) over one batch of
synthetic 436x1024 frame pairs already resident in HBM.  One process per GPU, frame pairs are independent so ranks share
nothing (no data-path collective; RCCL is used only for the barrier and the max-over-ranks of the elapsed time): weak
scaling, `value` = all pairs of all ranks / max time (ptlflow_amd.shard.timed_steps).

Besides the driver's contract fields the JSON line carries
  roofline      dominant kernel (largest summed time among the MFMA conv launches), HIP events around every
                launch of it during one separate instrumented forward; bound = fp32 MFMA peak 157.3 TFLOP/s
  cpu_baseline  the CPU oracle (`oracle/raft_oracle.raft_forward`, a port of the reference forward on torch
                CPU) timed on this host's cores on the same input: a reported baseline, not a target
  epe_vs_cpu    end-point error of the GPU `flows` vs that CPU forward on the same input (gate: mean <= 1e-3)
  batch1, model_benchmark_protocol   one pair per forward; the latter is the reference's own protocol
                (model_benchmark.py:421-466: torch.rand input, 1 warm-up + 10 synchronised forwards, median)
  split_bf16, skip_dead_upsample     the same forward with split-bf16 convolution products / without the reference's dead
                per-iteration mask + upsample work (bit-identical output) — beside the headline, never as it
  dropin        the path a ptlflow user gets: the reference's own `ptlflow.models.raft.raft.RAFT` (oracle/ref_loader.py) with plain
                `ptlflow_amd.patch.accelerate(model)` applied; batch 1 under the reference's protocol, driven by the reference's own
                `model_benchmark.estimate_inference_time`; `every_iteration`: the same object with the dead-work skip opted out
  config3       BASELINE config 3: gma (fp32), raft / gma with bf16 operands (bf16 convolutions, bf16 correlation volume), and
                SEA-RAFT's correlation path (per-level volumes against the halved fmap2 + 4 / 12 lookups, sea_raft/corr.py:71-117)
                in fp32 and bf16 — every leg with its end-point / lookup error against the CPU oracle
  config4       BASELINE config 4 on one GPU: raft on KITTI-sized 375x1242 pairs, batch 8 per GPU, fp32
  train         BASELINE config 5 on one GPU: a full RAFT training step (batch 10, 368x496, 12 iterations; forward, sequence
                loss, backward, clip, AdamW), every kernel libpfk's; samples/s
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

def host_cores() -> int:
    """Cores this process may really use: scheduler affinity, capped by the cgroup CPU quota.  os.cpu_count()
    reports the whole host inside a container; oversubscribing OpenMP beyond the quota makes torch CPU ops
    orders of magnitude slower."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, min(n, 64))


FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same table, "Peak BF16/FP16 MFMA" (dense)
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="frame pairs per GPU per step (throughput setting; the batch-1 "
                    "latency figure of the reference's model_benchmark.py protocol is reported alongside as `batch1`)")
    ap.add_argument("--height", type=int, default=436)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=32)
    ap.add_argument("--model", default="raft", choices=["raft", "raft_small", "gma"])
    ap.add_argument("--skip-dead-upsample", action="store_true",
                    help="skip mask head + upsampling on non-final iterations (output-identical dead work)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-batch1", action="store_true")
    ap.add_argument("--conv-precision", default="fp32", choices=["fp32", "bf16x6", "bf16x3", "bf16"],
                    help="arithmetic of the update block's convolutions for the timed run: fp32 MFMA (default, the headline) "
                         "or split-bf16 MFMA (include/pfk.h, pfk_conv2d_bf16s)")
    ap.add_argument("--no-split-modes", action="store_true", help="skip the extra split-bf16 legs (`split_bf16` in the output)")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the model_benchmark-protocol, gma, bf16 and training legs")
    ap.add_argument("--no-whole-models", action="store_true", help="skip the whole-model legs on the reference's SEARAFT / CCMR / MSRAFTPlus classes")
    ap.add_argument("--torch-baseline", action="store_true",
                    help="dropin leg: also time the un-patched torch / MIOpen forward of the same model (first call compiles MIOpen kernels)")
    ap.add_argument("--cpu-forwards", type=int, default=3)
    ap.add_argument("--cpu-budget-s", type=float, default=30.0, help="stop timing CPU forwards after this many seconds")
    return ap.parse_args()


def instrumented_forward(model, inputs):
    """One extra forward with HIP events around every MFMA conv launch -> per-kernel-key stats."""
    eng = model.engine(inputs["images"].device)
    eng.profile = {}
    try:
        model(inputs)
        torch.cuda.synchronize()
        stats = {}
        for key, evs in eng.profile.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            stats[key] = {"launches": len(ms), "avg_us": 1e3 * sum(ms) / len(ms), "total_ms": sum(ms),
                          "gflop_per_launch": eng.flops[key] / 1e9, "bytes_per_launch": eng.bytes.get(key)}
    finally:
        eng.profile = None
    return stats


def roofline_b16(model, inputs):
    """`roofline_bf16`: the dominant launch of the bf16-storage forward (K8b, `pfk_conv2d_b16`) against the LARGER of its two floors —
    algorithmic FLOPs / the dense bf16 matrix peak and algorithmic HBM bytes / 8 TB/s (these launches move 16-bit activations: some
    are bound by their epilogue bytes, not by the matrix pipe) — plus the per-call-site table and the bf16 lookup."""
    stats = instrumented_forward(model, inputs)
    lookup = stats.pop("lookup", None)

    def floors(v):
        t_mfma = v["gflop_per_launch"] * 1e9 / (BF16_MFMA_PEAK_TFLOPS * 1e12)
        t_hbm = (v["bytes_per_launch"] or 0.0) / (HBM_PEAK_GBS * 1e9)
        return t_mfma, t_hbm

    table = {}
    for k, v in sorted(stats.items(), key=lambda kv: -kv[1]["total_ms"]):
        t_mfma, t_hbm = floors(v)
        t = v["avg_us"] * 1e-6
        table[k] = {"avg_us": round(v["avg_us"], 2), "n": v["launches"], "tflops": round(v["gflop_per_launch"] / t / 1e3, 1),
                    "gbs": round((v["bytes_per_launch"] or 0.0) / t / 1e9, 1), "bound": "mfma" if t_mfma >= t_hbm else "hbm",
                    "frac_of_floor": round(max(t_mfma, t_hbm) / t, 3)}
    dom = max(stats, key=lambda k: stats[k]["total_ms"])
    v = stats[dom]
    t_mfma, t_hbm = floors(v)
    t = v["avg_us"] * 1e-6
    if t_mfma >= t_hbm:
        obj = {"bound": "mfma", "achieved": v["gflop_per_launch"] / t / 1e3, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s"}
    else:
        obj = {"bound": "hbm", "achieved": v["bytes_per_launch"] / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    obj["frac"] = obj["achieved"] / obj["peak"]
    traffic = None
    try:   # the committed rocprofv3 PMC passes of the K8b forward (profiles/pmc_traffic.json, entries `<site>_bf16@b<batch>`), if they cover this launch
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        B = inputs["images"].shape[0]
        rec = pmc["entries"].get(f"{dom.rstrip('12')}_bf16@b{B}")
        if rec and tuple(inputs["images"].shape[-2:]) == (436, 1024):
            from ptlflow_amd import _build
            traffic = {"bytes": (2 * rec["fetch_kb"] + rec["write_kb"]) * 1024, "algorithmic_bytes": int(v["bytes_per_launch"]),
                       "stale": _build.source_hash() != pmc.get("kernel_source_sha16"),
                       "source": "profiles/pmc_traffic.json (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes of the bf16-storage forward)"}
    except Exception:
        pass
    obj.update({"kernel": f"conv_gemm_b16_kernel[{dom}]", "traffic": traffic, "avg_us": v["avg_us"], "launches_per_forward": v["launches"],
                "gflop_per_launch": v["gflop_per_launch"], "algorithmic_bytes_per_launch": v["bytes_per_launch"],
                "floor_us": {"mfma": 1e6 * t_mfma, "hbm": 1e6 * t_hbm},
                "method": "HIP events around each launch, separate instrumented forward of the bf16-storage model (in situ, side stream live)",
                "kernels": table})
    if lookup and lookup.get("bytes_per_launch"):
        gbs = lookup["bytes_per_launch"] / (lookup["avg_us"] * 1e-6) / 1e9
        obj["lookup"] = {"kernel": "lookup_kernel (K3, bf16 maps in the blocked layout, paired fetch, bf16 rows out)", "bound": "hbm",
                         "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "avg_us": lookup["avg_us"],
                         "algorithmic_bytes_per_launch": lookup["bytes_per_launch"]}
    return obj


def timed(fn, warmup: int, steps: int) -> float:
    """Mean seconds per call of `fn` over `steps` calls after `warmup` untimed ones (device-synchronised on both sides)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def protocol_leg(model, dev, H, W, n: int = 10):
    """The reference's own benchmarking protocol (model_benchmark.py:421-466, utils/timer.py:81-96): batch 1, `torch.rand`
    images in [0, 1], one warm-up forward, then >= 10 forwards each bracketed by a device synchronisation; the MEDIAN wall
    time is what `model_benchmark-all.csv` publishes."""
    g = torch.Generator().manual_seed(1234)
    one = {"images": torch.rand(1, 2, 3, H, W, generator=g).to(dev)}
    model(one)
    torch.cuda.synchronize()
    times = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model(one)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    times.sort()
    med = 0.5 * (times[(n - 1) // 2] + times[n // 2])
    return {"value": 1.0 / med, "unit": "frame-pairs/s", "ms_median": 1e3 * med, "ms_min": 1e3 * times[0], "ms_max": 1e3 * times[-1],
            "protocol": "model_benchmark.py: batch 1, torch.rand input, 1 warm-up + 10 synchronised forwards, median"}


def _epe(flow_gpu, flow_ref):
    """(mean, max) end-point error of one [1, 2, H, W] flow against the CPU reference flow."""
    d = (flow_gpu.float().cpu() - flow_ref).pow(2).sum(1).sqrt()
    return float(d.mean()), float(d.max())


def reference_raft(cpu_state, iters, small=False):
    """The reference's own `ptlflow.models.raft.raft.RAFT` (oracle/ref_loader.py: /root/reference, or the archive staged for the
    GPU box by oracle/stage_ref.py) carrying the bench's weights — or None where no reference is importable."""
    try:
        from oracle import ref_loader          # the reference as the thing being accelerated / timed as the baseline
        if not ref_loader.reference_available():
            return None, None
        m = ref_loader.build_raft(small=small, iters=iters)
        missing, unexpected = m.load_state_dict(cpu_state, strict=False)
        # (the metric accumulators BaseModel registers — `train_metrics.*`, `val_metrics.*` — are not weights)
        missing = [k for k in missing if not k.startswith(("train_metrics.", "val_metrics."))]
        if missing or unexpected:
            raise RuntimeError(f"state_dict mismatch with the reference class: missing {missing[:3]}, unexpected {unexpected[:3]}")
        return m.eval(), ref_loader.REFERENCE_KIND
    except Exception as e:
        return None, repr(e)[:200]


def reference_function_protocol(model, H, W, n: int = 10):
    """The reference's OWN `model_benchmark.estimate_inference_time` (model_benchmark.py:422-466, its `ptlflow.utils.timer.Timer`:
    utils/timer.py:81-96) driving `model` — the function the published `model_benchmark-all.csv` numbers come from, imported
    unmodified (oracle/ref_loader.ref_script).  Returns the leg, or None where the script is not importable."""
    try:
        from oracle import ref_loader
        mb = ref_loader.ref_script("model_benchmark")
        from jsonargparse import Namespace
    except Exception as e:
        return {"error": repr(e)[:200]}
    times = sorted(mb.estimate_inference_time(Namespace(num_samples=n, batch_size=1), model, (H, W), "fp32"))
    med = times[len(times) // 2]                       # `final_speed_mode: median`, model_benchmark.py:335-341
    return {"value": 1.0 / med, "unit": "frame-pairs/s", "ms_median": 1e3 * med, "ms_min": 1e3 * times[0], "ms_max": 1e3 * times[-1],
            "driver": "reference function",
            "protocol": f"model_benchmark.estimate_inference_time (the reference's own function and Timer, {mb.__file__}): batch 1, "
                        f"torch.rand input, 1 warm-up + {n} synchronised forwards, median"}


def dropin_leg(cpu_state, dev, H, W, iters, pair_cpu, ref_flows, torch_baseline=False, batch8=None):
    """Throughput of the drop-in SEAM path — what `ptlflow.get_model("raft")` + `patch.accelerate(model)` runs: the REFERENCE'S
    OWN `RAFT` class (its `forward`, raft.py:125-194, its `preprocess_images` / `InputPadder` / `postprocess_predictions`) with
    seams B1 / B3 / B4 / B5 on libpfk.  The leg's own numbers are those of PLAIN `accelerate(model)` — the default a user gets: in
    eval the loop's dead mask-head / upsampling work is skipped and the last iteration's runs as the fused kernel (`flows`
    bit-identical); `every_iteration` = the same object with that opted out (`skip_dead_upsample=False, fuse_mask_upsample=False`:
    the reference loop's work in full).  Batch 1 under model_benchmark.py's protocol — driven by the reference's own
    `estimate_inference_time` where it is importable — plus the batch-8 throughput setting."""
    from ptlflow_amd import patch
    m, kind = reference_raft(cpu_state, iters)
    if m is None:
        return {"skipped": "no reference importable (neither /root/reference nor the staged archive oracle/_ref)" + (f": {kind}" if kind else "")}
    leg = {"what": (f"ptlflow.models.raft.raft.RAFT — the reference's own class ({kind}: "
                    f"{'/root/reference' if kind == 'tree' else 'oracle/_ref archive staged by oracle/stage_ref.py'}) "
                    "+ plain ptlflow_amd.patch.accelerate(model): seams B1/B3/B4/B5 on libpfk, dead work skipped (default)"),
           "model_class": f"{type(m).__module__}.{type(m).__name__}"}
    m = m.to(dev)
    if torch_baseline:      # stock PyTorch-ROCm ops (MIOpen convolutions, matmul / avg_pool2d / grid_sample): today's ptlflow on this GPU
        with torch.no_grad():
            t = protocol_leg(m, dev, H, W, n=5)
        leg["torch_rocm_unpatched"] = {"value": t["value"], "unit": "frame-pairs/s", "ms_median": t["ms_median"],
                                       "what": "the same model object before patch.accelerate: stock PyTorch-ROCm ops"}

    def measure(out_leg):
        with torch.no_grad():
            out = m({"images": pair_cpu.to(dev)})
            if ref_flows is not None:
                mean, mx = _epe(out["flows"][:1, 0], ref_flows)
                out_leg["epe_vs_cpu"] = {"mean": mean, "max": mx, "gate": 1e-3}
            t = reference_function_protocol(m, H, W)
            if t is None or "error" in t:                  # (the script could not be imported: this file's re-statement of the protocol)
                t2 = protocol_leg(m, dev, H, W)
                t2["driver"] = "bench.py re-statement" + (f" ({t['error']})" if t else "")
                t = t2
            out_leg.update(t)
            if batch8 is not None:
                sec = timed(lambda: m(batch8), 2, 5)
                out_leg["batch8"] = {"value": batch8["images"].shape[0] / sec, "unit": "frame-pairs/s", "ms_per_step": 1e3 * sec,
                                     "note": "the same accelerated object on the headline's batch (8 smooth pairs per forward)"}
        return out["flows"].clone()

    patch.accelerate(m)
    try:
        leg["dead_work_skip"] = getattr(m.update_block, "_skip", None) is not None
        flows_default = measure(leg)
    finally:
        patch.restore(m)
    try:
        every = {}
        patch.accelerate(m, skip_dead_upsample=False, fuse_mask_upsample=False)
        try:
            flows_every = measure(every)
            every["identical_flows"] = bool(torch.equal(flows_every, flows_default))
        finally:
            patch.restore(m)
        leg["every_iteration"] = every
        leg["identical_flows"] = every["identical_flows"]
    except Exception as e:
        leg["every_iteration"] = {"error": repr(e)[:300]}
    return leg


def whole_model_legs(dev, H, W, check: bool):
    """The other §8 families as WHOLE reference models on the GPU (BASELINE config 3's sea_raft side, SURVEY §8 f1's ccmr /
    ms_raft_p): the reference's own classes with the seams that apply —
      sea_raft_s_full   `SEARAFT` (ResNet-FPN + ConvNeXt block on torch / MIOpen; `get_corr_block` = seam B1 on K1-K3 with the
                        bilinear-1/2 pyramid), fp32 and under bf16 autocast, with and without the seam;
      ccmr / ms_raft_p  constructed with their defaults (`alternate_corr=True`): their `AlternateCorrBlock` calls this repo's
                        `alt_cuda_corr` plug-in (seam B2, K7) with zero patching; `accelerate` adds seam B3.
    pairs/s by model_benchmark.py's protocol (batch 1); EPE of sea_raft against the model's own CPU forward on the smooth pair
    (ccmr / ms_raft_p are checked against their CPU forwards in tests/test_gpu_reference_models.py at sizes whose materialised
    volume fits the CPU run)."""
    from oracle import ref_loader
    from ptlflow_amd import patch
    from ptlflow_amd.synth import smooth_pair
    if not ref_loader.reference_available():
        return {"skipped": "no reference importable (oracle/_ref not staged)"}
    legs = {}
    x1 = smooth_pair(1, H, W, seed=1234)

    def run(name, build, unpatched=True, autocast=False, epe=False, cpu_corr_module=None):
        leg = {}
        try:
            torch.manual_seed(1234)
            m = build().eval()
            ref = None
            if epe and check:
                # CPU side of an `alternate_corr=True` family: no CPU extension exists and the materialised volume of the 1/2-resolution
                # scale would be 50 GB, so the reference's own fallback runs — `IterativeCorrBlock`, which `get_corr_block` picks when its
                # module-global `alt_cuda_corr` is None ({family}/corr.py:104-118)
                import sys
                cm = sys.modules.get(cpu_corr_module) if cpu_corr_module else None
                saved = getattr(cm, "alt_cuda_corr", None) if cm is not None else None
                if cm is not None:
                    cm.alt_cuda_corr = None
                try:
                    t0 = time.perf_counter()
                    with torch.no_grad():
                        ref = m({"images": x1.clone()})["flows"][:, 0]
                    leg["cpu_forward_s"] = time.perf_counter() - t0
                finally:
                    if cm is not None:
                        cm.alt_cuda_corr = saved
                if cm is not None:
                    leg["cpu_side"] = "the same model object on the CPU with the reference's IterativeCorrBlock fallback (alt_cuda_corr unset)"
            m = m.to(dev)
            with torch.no_grad():
                stock_bf16 = None
                if unpatched:
                    t = protocol_leg(m, dev, H, W, n=5)
                    leg["unpatched_torch_rocm"] = {"value": t["value"], "ms_median": t["ms_median"]}
                    if autocast and ref is not None:     # what bf16 autocast costs the SAME model on stock PyTorch-ROCm ops
                        with torch.autocast("cuda", dtype=torch.bfloat16):
                            stock_bf16 = _epe(m({"images": x1.to(dev)})["flows"][:1, 0], ref)
                patch.accelerate(m)
                try:
                    t = protocol_leg(m, dev, H, W)
                    leg.update({"value": t["value"], "unit": "frame-pairs/s", "ms_median": t["ms_median"], "protocol": t["protocol"]})
                    if ref is not None:
                        mean, mx = _epe(m({"images": x1.to(dev)})["flows"][:1, 0], ref)
                        leg["epe_vs_cpu"] = {"mean": mean, "max": mx, "gate": 1e-3}
                    if autocast:
                        with torch.autocast("cuda", dtype=torch.bfloat16):
                            t = protocol_leg(m, dev, H, W)
                            leg["bf16_autocast"] = {"value": t["value"], "ms_median": t["ms_median"]}
                            if ref is not None:
                                mean, mx = _epe(m({"images": x1.to(dev)})["flows"][:1, 0], ref)
                                leg["bf16_autocast"]["epe_vs_cpu_fp32"] = {"mean": mean, "max": mx}
                                if stock_bf16 is not None:
                                    leg["bf16_autocast"]["epe_of_stock_torch_rocm_autocast_vs_cpu_fp32"] = {"mean": stock_bf16[0], "max": stock_bf16[1]}
                finally:
                    patch.restore(m)
            leg["model_class"] = f"{type(m).__module__}.{type(m).__name__}"
            del m
        except Exception as e:
            leg["error"] = repr(e)[:300]
        torch.cuda.empty_cache()
        legs[name] = leg

    S = ref_loader.ref_module("ptlflow.models.sea_raft.sea_raft")
    run("sea_raft_s_full", lambda: S.SEARAFT(block_dims=[64, 128, 256]), autocast=True, epe=True)
    C = ref_loader.ref_module("ptlflow.models.ccmr.ccmr")
    # (un-patched they cannot run on the GPU at all: no alt_cuda_corr without this repo)
    run("ccmr_full", lambda: C.CCMR(), unpatched=False, epe=True, cpu_corr_module="ptlflow.models.ccmr.corr")
    M = ref_loader.ref_module("ptlflow.models.ms_raft_plus.ms_raft_plus")
    run("ms_raft_p_full", lambda: M.MSRAFTPlus(), unpatched=False, epe=True, cpu_corr_module="ptlflow.models.ms_raft_plus.corr")
    return legs


def sea_raft_corr_leg(dev, bf16: bool, batch=8, h=55, w=128, D=256, check=True):
    """BASELINE config 3, SEA-RAFT side: its correlation path on the shared kernels (sea_raft/corr.py:71-117) at the 436x1024
    grid — four per-level volumes (fmap1 x fmap2 halved l times, `pfk_fmap_pool2x2_f32` + K1 / K1b) and the per-iteration
    radius-4 lookup (K3) — for SEA-RAFT's 4 (sea_raft_s / _m) and 12 (sea_raft_l) iterations.  The ConvNeXt update block and
    the ResNet encoders of that family are out of scope, so pairs/s here is of the correlation path alone.
    `err_vs_cpu_fp32`: the last lookup of pair 0 against the CPU oracle's fp32 pyramid + lookup on the same maps / coordinates."""
    from ptlflow_amd.corr import CorrBlock
    g = torch.Generator().manual_seed(77)
    f1 = torch.randn(batch, D, h, w, generator=g) * 0.5
    f2 = torch.randn(batch, D, h, w, generator=g) * 0.5
    base = torch.stack(torch.meshgrid(torch.arange(w, dtype=torch.float32), torch.arange(h, dtype=torch.float32), indexing="xy"), 0)
    coords = [(base[None] + torch.rand(batch, 2, h, w, generator=g) * 16 - 8).contiguous() for _ in range(12)]
    dt = torch.bfloat16 if bf16 else torch.float32
    d1, d2 = f1.to(dev).to(dt), f2.to(dev).to(dt)
    dc = [c.to(dev) for c in coords]
    blk = CorrBlock(d1, d2, num_levels=4, radius=4, pyramid="bilinear_f2")
    leg = {"batch": batch, "grid": f"{h}x{w}", "dim": D, "volume_dtype": str(blk.volume_dtype).replace("torch.", "")}
    for iters in (4, 12):
        def run():
            blk.update(d1, d2)
            for i in range(iters):
                blk.lookup_pm(dc[i])
        sec = timed(run, 2, 10)
        leg[f"iters{iters}"] = {"value": batch / sec, "unit": "frame-pairs/s (correlation path only)", "ms_per_step": 1e3 * sec}
    if check:
        from oracle import raft_oracle as O  # checker only
        got = blk(dc[11])[:1].float().cpu()
        want = O.lookup(O.sea_correlation_pyramid(f1[:1], f2[:1], 4), coords[11][:1], 4)
        err = (got - want).abs()
        leg["err_vs_cpu_fp32"] = {"max_abs": float(err.max()), "mean_abs": float(err.mean()), "ref_max_abs": float(want.abs().max())}
    return leg


def train_leg(dev, batch=10, H=368, W=496, iters=12, steps=4, warmup=2):
    """BASELINE config 5 on one GPU: a full training step of RAFT on a FlyingChairs-shaped batch (raft-train1-chairs.yaml:
    batch 10 per rank, 368x496 crops, 12 iterations, gamma 0.8, AdamW lr 4e-4 wd 1e-4, gradient clipping at 1.0):
    train-mode forward -> sequence loss -> backward -> clip -> optimizer step, all inside the timed region."""
    from ptlflow_amd.raft import RAFT
    from ptlflow_amd.train import sequence_loss
    model = RAFT(iters=iters).load_synthetic(1234).to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=4e-4, weight_decay=1e-4, eps=1e-8)
    g = torch.Generator().manual_seed(99)
    inputs = {"images": torch.rand(batch, 2, 3, H, W, generator=g).to(dev)}
    gt = (torch.rand(batch, 2, H, W, generator=g) * 20 - 10).to(dev)
    valid = torch.ones(batch, 1, H, W, device=dev)
    last = {}

    def step():
        out = model(inputs)
        loss = sequence_loss(out["flow_preds"], gt, valid, 0.8, 400.0)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        last["loss"] = loss

    sec = timed(step, warmup, steps)

    # share of the step spent in the two encoders (forward + backward of fnet on both frames and cnet), on the path the model
    # really uses: libpfk autograd nodes (ptlflow_amd/train_encoder.py) — or torch / MIOpen with native_encoders=False
    from ptlflow_amd.train_encoder import encoder_train

    def enc_only():
        x, _ = model.preprocess(inputs["images"])
        i1, i2 = x[:, 0].contiguous(), x[:, 1].contiguous()
        if model.native_encoders:
            f, c = encoder_train(model.fnet, torch.cat([i1, i2], 0)), encoder_train(model.cnet, i1)
            (f.square().mean() + c.square().mean()).backward()
        else:
            f1, f2 = model.fnet([i1, i2])
            c = model.cnet(i1)
            (f1.square().mean() + f2.square().mean() + c.square().mean()).backward()
        opt.zero_grad(set_to_none=True)

    enc = timed(enc_only, 1, 3)
    launches = None
    try:     # kernel launches of one whole step (forward, loss, backward, clip, optimizer), counted by the profiler on an extra step
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step()
            torch.cuda.synchronize()
        launches = sum(1 for e in prof.events() if str(getattr(e, "device_type", "")).endswith("CUDA"))
    except Exception:
        pass
    return {"value": batch / sec, "unit": "samples/s", "ms_per_step": 1e3 * sec, "loss": float(last["loss"]),
            "launches_per_step": launches,
            "encoders_fwd_bwd_ms": 1e3 * enc, "encoder_share": enc / sec,
            "encoders_on": "libpfk" if model.native_encoders else "torch/MIOpen",
            "config": f"raft train step, batch {batch}, {H}x{W}, {iters} iterations, fp32, sequence loss + backward + clip 1.0 + AdamW; "
                      "encoders, correlation / lookup, update block and upsampling forward+backward on libpfk autograd nodes"}


def self_launch(args) -> None:
    """`python bench.py --gpus N` (N > 1) started WITHOUT a launcher: start the N ranks ourselves — one process per GPU under
    `torch.distributed.run` on 127.0.0.1, the invocation the docstring shows — and exit with their status.  What this replaces
    in the reference is Lightning's own process launch (ptlflow/utils/lightning/ptlflow_trainer.py:71, :281).  Refuses loudly
    when the box has fewer than N devices (unless PFK_BENCH_SHARED_DEVICE=1 asks for the one-device code-path smoke): a line
    that says `n_gpus: 1` for a `--gpus 8` request would be a wrong SCALE record."""
    import subprocess
    shared = os.environ.get("PFK_BENCH_SHARED_DEVICE") == "1"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and not shared:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_dev} GPU(s) visible on this node; refusing to report fewer ranks "
                         "than requested (PFK_BENCH_SHARED_DEVICE=1 runs all ranks on cuda:0 as a code-path smoke)")
    # `--standalone`: torchrun binds a free port of 127.0.0.1 ITSELF (no probe socket of ours whose port another process could
    # take between our close and its bind) and hands MASTER_ADDR / MASTER_PORT to the ranks
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // args.gpus)))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # PFK_BENCH_SHARED_DEVICE=1: every rank uses cuda:0 and the process group runs on gloo — a smoke of the N > 1 branch (rank
    # env, barrier, max-over-ranks, whole-job throughput) on a box with ONE GPU, where RCCL refuses two ranks on one device.
    # Its numbers are not a scaling measurement: the ranks time-share one chip.
    shared = os.environ.get("PFK_BENCH_SHARED_DEVICE") == "1"
    dev_index = 0 if shared else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist, ranks_seen = None, 1
    # under torch.distributed.run (RANK set) the process group is created even for one rank, so that a single-GPU box can
    # exercise the RCCL branch (init, barrier, max-reduce) that the N > 1 runs rely on
    if world > 1 or "RANK" in os.environ:
        import torch.distributed as dist  # noqa: F811
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)   # "nccl" is RCCL on ROCm
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                                 # every rank contributes 1: the job size as the collective sees it
        ranks_seen = int(ones.item())
        if ranks_seen != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the all-reduce counted {ranks_seen} rank(s)")

    import ptlflow_amd
    from ptlflow_amd.raft import GMA, RAFT
    from ptlflow_amd.synth import smooth_pair

    ptlflow_amd.load_native()
    if os.environ.get("PFK_CUDNN_BENCHMARK") == "1":      # experiment knob: MIOpen find mode for the encoders
        torch.backends.cudnn.benchmark = True
    small = args.model == "raft_small"
    if args.model == "gma":
        def make(prec, every_iter=not args.skip_dead_upsample):
            return GMA(iters=args.iters, upsample_every_iter=every_iter, conv_precision=prec)
    else:
        def make(prec, every_iter=not args.skip_dead_upsample):
            return RAFT(small=small, iters=args.iters, upsample_every_iter=every_iter, conv_precision=prec)
    model = make(args.conv_precision)
    if os.environ.get("PFK_OVERLAP") == "0":          # experiment knob: mask head + upsampling on the main stream
        model.overlap_mask_head = False
    if os.environ.get("PFK_FUSE_MASK") == "1":        # experiment knob: K13 (fused mask conv2 + softmax + upsampling) instead of the pair
        model.fuse_mask_upsample = True
    model.load_synthetic(1234).eval()
    cpu_state = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev)
    if os.environ.get("PFK_CHANNELS_LAST") == "1":        # experiment knob: NHWC encoders
        model.fnet = model.fnet.to(memory_format=torch.channels_last)
        model.cnet = model.cnet.to(memory_format=torch.channels_last)
    images_cpu = smooth_pair(args.batch, args.height, args.width, seed=1234 + rank)
    inputs = {"images": images_cpu.to(dev)}

    # warm-up, barrier + device sync, EXACTLY `steps` timed steps, barrier + sync, max over ranks: ptlflow_amd/shard.py
    # (the same function the world_size-2 gloo test drives on CPU)
    from ptlflow_amd.shard import timed_steps
    state = {}

    def one_step():
        state["out"] = model(inputs)

    elapsed = timed_steps(one_step, args.steps, args.warmup, torch.cuda.synchronize)
    out = state["out"]
    # stream-K fix-ups that timed out would have poisoned tiles with NaN: a timed run with a non-zero count is not a result
    faults = model.engine(dev).faults()
    if faults:
        raise SystemExit(f"stream-K workspace reports {faults} timed-out fix-up(s): results invalid")

    pairs = args.batch * args.steps * world
    result = {
        "metric": "frame-pairs/sec, RAFT 32-iter 436x1024" if (args.model == "raft" and args.iters == 32 and (args.height, args.width) == (436, 1024))
                  else f"frame-pairs/sec, {args.model} {args.iters}-iter {args.height}x{args.width}",
        "value": pairs / elapsed,
        "unit": "frame-pairs/s",
        "n_gpus": world,
        # all-reduce of ones over the process group (None: no group, plain 1-GPU run) and the backend that carried it
        "ranks_seen": ranks_seen if dist is not None else None,
        "backend": None if dist is None else ("gloo (shared-device smoke)" if shared else "nccl (RCCL)"),
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32" if args.conv_precision == "fp32" else
                 f"f32 storage/accumulate, update-block conv products as split bf16 ({args.conv_precision})",
        "data": "synthetic",
        "streamk_faults": faults,
        "config": {"workload": f"{args.model} random-init (seeded), {args.height}x{args.width} frame pairs, {args.iters} iterations, "
                               f"{args.conv_precision}, batch {args.batch}/GPU, eval forward incl. encoders, "
                               + ("dead mask/upsample work skipped on non-final iterations" if args.skip_dead_upsample
                                  else "mask head + convex upsample on every iteration as the reference"),
                   "global_batch": args.batch * world, "parallelism": f"dp{world} (independent replicas, no collectives)"},
    }
    if shared:
        result["config"]["shared_device"] = "all ranks on cuda:0 over gloo (N > 1 code-path smoke on a one-GPU box, not a scaling number)"

    if rank == 0:
        if args.batch > 1 and not args.no_batch1:
            one = {"images": inputs["images"][:1].contiguous()}
            for _ in range(2):
                model(one)
            torch.cuda.synchronize()
            b0 = time.perf_counter()
            for _ in range(10):
                model(one)
            torch.cuda.synchronize()
            ms1 = 1e3 * (time.perf_counter() - b0) / 10
            result["batch1"] = {"value": 1e3 / ms1, "unit": "frame-pairs/s", "ms_per_forward": ms1,
                                "note": "same model, one pair per forward (model_benchmark.py protocol), 10 timed forwards"}
        if args.batch == 8 and not args.no_batch1 and args.conv_precision == "fp32":
            # the same forward at 16 pairs per GPU (memory is not the limit on 288 GB; the shorter launches quantise better on 256
            # CUs): reported beside `value`, which stays at the batch 8 of every earlier round
            x16 = {"images": smooth_pair(16, args.height, args.width, seed=4321 + rank).to(dev)}
            sec16 = timed(lambda: model(x16), 1, 3)
            result["batch16"] = {"value": 16 / sec16, "unit": "frame-pairs/s", "ms_per_step": 1e3 * sec16}
            del x16
            torch.cuda.empty_cache()
        if not args.no_roofline:
            stats = instrumented_forward(model, inputs)
            lookup_stats = stats.pop("lookup", None)        # the HBM-bound kernel of the path: reported beside the dominant one
            dom = max(stats, key=lambda k: stats[k]["total_ms"])
            s = stats[dom]
            achieved = s["gflop_per_launch"] / (s["avg_us"] * 1e-6) / 1e3  # TFLOP/s
            traffic = None
            try:   # PMC counters cannot be read from inside this process: take the committed rocprofv3 figures if they cover this launch
                pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
                rec = pmc["entries"].get(f"{dom.rstrip('12')}@b{args.batch}")
                if rec and (args.height, args.width, small) == (436, 1024, False):
                    from ptlflow_amd import _build      # the stamp both libraries carry: every file of csrc/ + include/pfk.h + flags
                    # MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half the bytes of the 128-byte requests these kernels
                    # cause (calibrated in profiles/r03_b) -> doubled; WRITE_SIZE taken as reported (KB)
                    traffic = {"bytes": (2 * rec["fetch_kb"] + rec["write_kb"]) * 1024,
                               "algorithmic_bytes": int(s["bytes_per_launch"]) if s.get("bytes_per_launch") else None,
                               "stale": _build.source_hash() != pmc.get("kernel_source_sha16"),
                               "source": "profiles/pmc_traffic.json (rocprofv3 TCC_EA0_RDREQ x 128 B / FETCH_SIZE x2 + WRITE_SIZE, separate "
                                         "--pmc passes; `stale`: the kernel sources changed since that measurement)"}
            except Exception:
                pass
            result["roofline"] = {"kernel": f"conv_gemm_kernel[{dom}]", "bound": "mfma", "achieved": achieved,
                                  "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP32_MFMA_PEAK_TFLOPS,
                                  "traffic": traffic, "avg_us": s["avg_us"], "launches_per_forward": s["launches"],
                                  "gflop_per_launch": s["gflop_per_launch"],
                                  "method": "HIP events around each launch, separate instrumented forward"}
            if lookup_stats and lookup_stats.get("bytes_per_launch"):
                gbs = lookup_stats["bytes_per_launch"] / (lookup_stats["avg_us"] * 1e-6) / 1e9
                tr = None
                try:
                    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
                    rec = pmc["entries"].get(f"lookup@b{args.batch}")
                    if rec and (args.height, args.width, small) == (436, 1024, False):
                        from ptlflow_amd import _build
                        tr = {"bytes": (2 * rec["fetch_kb"] + rec["write_kb"]) * 1024, "algorithmic_bytes": int(lookup_stats["bytes_per_launch"]),
                              "stale": _build.source_hash() != pmc.get("kernel_source_sha16")}
                except Exception:
                    pass
                result["roofline_lookup"] = {"kernel": "lookup_kernel (K3, blocked 4x8 volume layout)", "bound": "hbm", "achieved": gbs,
                                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": tr,
                                             "avg_us": lookup_stats["avg_us"], "launches_per_forward": lookup_stats["launches"],
                                             "algorithmic_bytes_per_launch": lookup_stats["bytes_per_launch"],
                                             "method": "HIP events around each lookup launch in the same instrumented forward (in situ: the "
                                                       "2.1 GB pyramid is cold in the caches between iterations)"}
            result["kernels"] = {k: {"avg_us": round(v["avg_us"], 2), "n": v["launches"],
                                     "tflops": round(v["gflop_per_launch"] / (v["avg_us"] * 1e-6) / 1e3, 1)}
                                 for k, v in sorted(stats.items(), key=lambda kv: -kv[1]["total_ms"])}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import raft_oracle as O  # checker + reported baseline only
            cores = host_cores()
            torch.set_num_threads(cores)

            def cpu_times(fn):
                times, res, t_start = [], None, time.perf_counter()
                for i in range(args.cpu_forwards + 1):
                    c0 = time.perf_counter()
                    res = fn()
                    dt = time.perf_counter() - c0
                    # the first forward is a warm-up unless the budget leaves room for nothing else
                    if i or dt > args.cpu_budget_s / 2:
                        times.append(dt)
                    if time.perf_counter() - t_start > args.cpu_budget_s:
                        break
                times.sort()
                return times, res

            # The baseline is the REFERENCE'S OWN forward wherever it is importable (/root/reference, or the archive
            # oracle/stage_ref.py staged for the GPU box): kind "reference".  Only without it the port is timed (kind "port").
            rm, ref_kind = (reference_raft(cpu_state, args.iters, small=small) if args.model in ("raft", "raft_small") else (None, None))
            if rm is not None:
                with torch.no_grad():
                    times, ref = cpu_times(lambda: rm({"images": images_cpu[:1]}))
                ref = {"flows": ref["flows"].float()}
                kind = "reference"
                sample = (f"{len(times)} forward(s) of the reference's own ptlflow.models.raft.raft.{'RAFTSmall' if small else 'RAFT'}.forward (oracle/ref_loader.py, "
                          f"source: {ref_kind}) on the first frame pair of the batch, median; torch {torch.__version__} CPU, {cores} threads")
                del rm
            else:
                if args.model == "gma":
                    times, ref = cpu_times(lambda: O.gma_forward(cpu_state, images_cpu[:1], iters=args.iters))
                else:
                    times, ref = cpu_times(lambda: O.raft_forward(cpu_state, images_cpu[:1], iters=args.iters, small=small))
                kind = "port"
                sample = (f"{len(times)} full forward(s) of oracle/raft_oracle.py (bit-identical restatement of the reference forward, ~12 % "
                          f"slower: explicit gather lookup vs grid_sample) on the first frame pair of the batch, median; "
                          f"torch {torch.__version__} CPU, {cores} threads" + (f"; reference not importable: {ref_kind}" if ref_kind else ""))
            med = times[len(times) // 2]
            result["cpu_baseline"] = {"value": 1.0 / med, "unit": "frame-pairs/s", "cores": cores, "kind": kind, "sample": sample}
            mean, mx = O.epe(out["flows"][:1, 0].float().cpu(), ref["flows"][:, 0])
            result["epe_vs_cpu"] = {"mean": mean, "max": mx, "gate": 1e-3}
        else:
            ref = None
        if world == 1 and args.conv_precision == "fp32" and not args.no_split_modes:
            # Same forward with the update block's convolutions on the bf16 matrix cores with split operands
            # (reported beside the fp32 headline, never as `value`): throughput at the same batch and EPE against the same
            # CPU forward (or, without it, against the fp32 GPU output).
            modes = {}
            base = out["flows"][:1, 0].float().cpu()
            for prec in ("bf16x6", "bf16x3"):
                m2 = make(prec).eval()
                m2.load_state_dict(cpu_state)
                m2 = m2.to(dev)
                for _ in range(2):
                    o2 = m2(inputs)
                torch.cuda.synchronize()
                s0 = time.perf_counter()
                for _ in range(5):
                    o2 = m2(inputs)
                torch.cuda.synchronize()
                ms = 1e3 * (time.perf_counter() - s0) / 5
                # end-point error against the CPU forward of the cpu_baseline leg when it ran, else against the fp32 GPU output
                target, against = (ref["flows"][:, 0], "cpu") if ref is not None else (base, "gpu_fp32")
                d = (o2["flows"][:1, 0].float().cpu() - target).pow(2).sum(1).sqrt()
                mean, mx = float(d.mean()), float(d.max())
                modes[prec] = {"value": args.batch * 1e3 / ms, "unit": "frame-pairs/s", "ms_per_step": ms,
                               "epe_mean": mean, "epe_max": mx, "epe_against": against}
                del m2, o2
                torch.cuda.empty_cache()
            result["split_bf16"] = modes
        if world == 1 and args.conv_precision == "fp32" and not args.no_split_modes and not args.skip_dead_upsample and not small:
            # The reference computes the mask head and the convex upsampling on every iteration although eval only returns the
            # last one (raft.py:180-187).  `value` above does the same work; this leg skips the dead work (bit-identical output,
            # tests/test_gpu_model.py) and is reported beside it.
            m3 = make("fp32", every_iter=False).eval()
            m3.load_state_dict(cpu_state)
            m3 = m3.to(dev)
            for _ in range(2):
                o3 = m3(inputs)
            torch.cuda.synchronize()
            s0 = time.perf_counter()
            for _ in range(5):
                o3 = m3(inputs)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - s0) / 5
            result["skip_dead_upsample"] = {"value": args.batch * 1e3 / ms, "unit": "frame-pairs/s", "ms_per_step": ms,
                                            "identical_output": bool(torch.equal(o3["flows"], out["flows"]))}
            del m3, o3
        if world == 1 and args.conv_precision == "fp32" and not args.no_split_modes and args.model == "raft":
            # A/B of the loop-invariant hoist (ptlflow_amd/update.py, UpdateEngine): the same forward with the GRU convolutions in
            # the reference's single-chain form — conv over cat([h, inp, motion]) every iteration — beside `value`, which computes
            # the context features' part of them once per forward
            m4 = RAFT(small=small, iters=args.iters, upsample_every_iter=not args.skip_dead_upsample, hoist_context=False).eval()
            m4.load_state_dict(cpu_state)
            m4 = m4.to(dev)
            sec = timed(lambda: m4(inputs), 2, 5)
            o4 = m4(inputs)
            d = (o4["flows"][:1, 0].float() - out["flows"][:1, 0].float()).pow(2).sum(1).sqrt()
            result["single_chain_gru"] = {"value": args.batch / sec, "unit": "frame-pairs/s", "ms_per_step": 1e3 * sec,
                                          "epe_vs_value_mean": float(d.mean()), "epe_vs_value_max": float(d.max()),
                                          "what": "hoist_context=False: every GRU convolution over all of cat([h, inp, motion]) in every "
                                                  "iteration, as round 3 ran it"}
            if ref is not None:
                mean, mx = _epe(o4["flows"][:1, 0], ref["flows"][:, 0])
                result["single_chain_gru"]["epe_vs_cpu"] = {"mean": mean, "max": mx}
            del m4, o4
            torch.cuda.empty_cache()
        default_cfg = args.model == "raft" and (args.height, args.width, args.iters) == (436, 1024, 32) and args.conv_precision == "fp32"
        if world == 1 and default_cfg and not args.no_extra_legs:
            del out
            torch.cuda.empty_cache()
            check = not args.no_cpu_baseline            # the CPU oracle as the CHECKER of every extra leg (never timed here)
            if check:
                from oracle import raft_oracle as O
            ref_raft = ref["flows"][:, 0] if ref is not None else None      # CPU forward of the cpu_baseline leg: same weights, pair 0
            # (a) like-for-like with the reference's published protocol (batch 1, rand, median of 10) on the mirror ...
            mbp = reference_function_protocol(model, args.height, args.width)     # the reference's own function drives the mirror
            if mbp is None or "error" in mbp:
                mbp2 = protocol_leg(model, dev, args.height, args.width)
                mbp2["driver"] = "bench.py re-statement" + (f" ({mbp['error']})" if mbp else "")
                mbp = mbp2
            result["model_benchmark_protocol"] = mbp
            del model
            torch.cuda.empty_cache()
            # ... and on the drop-in seam path (the reference's caller loop, seams B1 / B3 / B4 patched)
            try:
                result["dropin"] = dropin_leg(cpu_state, dev, args.height, args.width, args.iters, images_cpu[:1], ref_raft,
                                              args.torch_baseline, batch8=inputs)
            except Exception as e:  # a leg must never take the headline line down with it
                result["dropin"] = {"error": repr(e)[:300]}
            torch.cuda.empty_cache()
            # (b) BASELINE config 3: gma on the shared CorrBlock / GRU path (fp32), bf16 operands for raft and gma, and SEA-RAFT's
            # correlation path; EPE of pair 0 against the CPU oracle's fp32 forward with the same weights
            legs = {}
            ref_gma = None
            for name, ctor, b in (("gma_fp32", lambda: GMA(iters=32), 4), ("raft_bf16", lambda: RAFT(iters=32, conv_precision="bf16"), 8),
                                  ("gma_bf16", lambda: GMA(iters=32, conv_precision="bf16"), 4)):
                try:
                    m = ctor().load_synthetic(1234).eval()
                    if check and name == "gma_fp32":
                        ref_gma = O.gma_forward({k: v.clone() for k, v in m.state_dict().items()},
                                                smooth_pair(1, args.height, args.width, seed=1234), iters=32)["flows"][:, 0]
                    m = m.to(dev)
                    xin = {"images": smooth_pair(b, args.height, args.width, seed=1234).to(dev)}
                    sec = timed(lambda: m(xin), 2, 5)
                    legs[name] = {"value": b / sec, "unit": "frame-pairs/s", "ms_per_step": 1e3 * sec, "batch": b}
                    target = ref_gma if name.startswith("gma") else ref_raft
                    if target is not None:
                        mean, mx = _epe(m(xin)["flows"][:1, 0], target)
                        legs[name].update({"epe_mean": mean, "epe_max": mx, "epe_against": "cpu fp32 forward of the same model"})
                    if name.endswith("_bf16"):      # the same model at the reference's protocol size (one pair per forward, median of 10)
                        x1 = {"images": xin["images"][:1].contiguous()}
                        for _ in range(2):
                            m(x1)
                        torch.cuda.synchronize()
                        ts = []
                        for _ in range(10):
                            t0 = time.perf_counter(); m(x1); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
                        ts.sort()
                        legs[name]["batch1"] = {"value": 1.0 / ts[len(ts) // 2], "unit": "frame-pairs/s", "ms_median": 1e3 * ts[len(ts) // 2]}
                    if name == "raft_bf16" and not args.no_roofline:
                        try:
                            result["roofline_bf16"] = roofline_b16(m, xin)
                        except Exception as e:
                            result["roofline_bf16"] = {"error": repr(e)[:300]}
                    del m, xin
                    torch.cuda.empty_cache()
                except Exception as e:
                    legs[name] = {"error": repr(e)[:300]}
            for name, bf in (("sea_raft_corr_f32", False), ("sea_raft_corr_bf16", True)):
                try:
                    legs[name] = sea_raft_corr_leg(dev, bf, check=check)
                except Exception as e:
                    legs[name] = {"error": repr(e)[:300]}
                torch.cuda.empty_cache()
            if not args.no_whole_models:
                try:
                    legs.update(whole_model_legs(dev, args.height, args.width, check))
                except Exception as e:
                    legs["whole_models"] = {"error": repr(e)[:300]}
            result["config3"] = legs
            # (c) BASELINE config 4 on one GPU: KITTI-sized pairs (375x1242 -> padded 376x1248, 47x156 grid), batch 8 per GPU
            try:
                m = RAFT(iters=32).eval()
                m.load_state_dict(cpu_state)
                m = m.to(dev)
                xk = smooth_pair(8, 375, 1242, seed=4321)
                xin = {"images": xk.to(dev)}
                sec = timed(lambda: m(xin), 2, 5)
                leg = {"value": 8 / sec, "unit": "frame-pairs/s", "ms_per_step": 1e3 * sec, "batch": 8,
                       "config": "raft, 375x1242 (KITTI), 32 iterations, fp32, batch 8 per GPU (BASELINE config 4: 64 pairs over 8 GPUs, no collectives)"}
                if check:
                    rk = O.raft_forward(cpu_state, xk[:1], iters=32)["flows"][:, 0]
                    mean, mx = _epe(m(xin)["flows"][:1, 0], rk)
                    leg["epe_vs_cpu"] = {"mean": mean, "max": mx, "gate": 1e-3}
                result["config4"] = leg
                del m, xin
                torch.cuda.empty_cache()
            except Exception as e:
                result["config4"] = {"error": repr(e)[:300]}
            # (d) BASELINE config 5: the training step
            try:
                result["train"] = train_leg(dev)
            except Exception as e:
                result["train"] = {"error": repr(e)[:300]}
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
