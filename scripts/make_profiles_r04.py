#!/usr/bin/env python3
"""profiles/r04_d (kernel traces + the bench line of the same run) and profiles/pmc_traffic.json (per-launch HBM traffic,
hash-stamped) of the round-4 final tree from the raw rocprofv3 output of scripts/gpu_final_r04.sh in gpurun_out/ (CPU only).

    python scripts/make_profiles_r04.py
"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles", "r04_d_kerneltrace_final.md")


def bench_line():
    for line in reversed(open(os.path.join(O, "y_bench.log")).read().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit("no JSON line in y_bench.log")


def cut(text, n):
    return "\n".join(l[:n] for l in text.splitlines())


def tail(name, n=3):
    try:
        return "\n".join(open(os.path.join(O, name)).read().strip().splitlines()[-n:])
    except OSError:
        return "(missing)"


def suite():
    try:
        lines = [l for l in open(os.path.join(O, "y_pytest.log")).read().splitlines() if " passed" in l or " failed" in l]
        return lines[-1].strip() if lines else "(no summary line)"
    except OSError:
        return "(missing)"


def main():
    d = bench_line()
    r = d["roofline"]
    c3 = d.get("config3", {})
    head = f"""# r04_d — round 4, final tree: kernel traces and the bench line of the same run (MI355X, one GPU)

Commands (`scripts/gpu_final_r04.sh`, one gpurun call): the whole GPU suite, `__graft_entry__.smoke()`, the driver's command
`python3 bench.py --gpus 1 --steps 20 --warmup 5`, micro-benches, then from /tmp with TMPDIR=/tmp
`rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline --no-roofline --no-split-modes --no-extra-legs
--no-batch1 [--batch 1] --steps K --warmup W`, `... -- python scripts/seam_prof.py` (the drop-in seam path: the reference's own `ptlflow.models.raft.raft.RAFT` out of the
staged archive + patch.accelerate, batch 1, 8 forwards) and `... -- python scripts/train_prof.py`
(4 training steps, batch 10, 368x496, 12 iterations).  Tables by scripts/trace_stats.py (regs = VGPRs + AGPRs per dispatch; scratch
must read 0 everywhere — tests/test_no_scratch.py).  Last whole GPU suite of the round (gpurun_out/y_pytest.log; an earlier tree of this round — the tests added after it, 325 in all now, ran in their own calls, DESIGN.md section 4): `{suite()}`.

**Bench line of this run** (gpurun_out/y_bench.log): **{d['value']:.1f} frame-pairs/s** fp32 ({d['ms_per_step']:.1f} ms/step, batch 8), EPE vs
the CPU oracle {d['epe_vs_cpu']['mean']:.2e} mean / {d['epe_vs_cpu']['max']:.2e} max, stream-K faults {d['streamk_faults']}; roofline
{r['kernel']}: {r['avg_us']:.1f} us = {r['achieved']:.1f} TFLOP/s = **{r['frac']:.3f}** of {r['peak']}; batch 1 {d['batch1']['value']:.1f};
model_benchmark protocol {d['model_benchmark_protocol']['value']:.1f} pairs/s ({d['model_benchmark_protocol']['ms_median']:.2f} ms median) on the mirror,
**{d['dropin']['value']:.1f} on the drop-in seam path, model class {d['dropin'].get('model_class')}** ({d['dropin']['ms_median']:.2f} ms, EPE
{d['dropin']['epe_vs_cpu']['mean']:.2e}; batch 8: {d['dropin'].get('batch8', {}).get('value', float('nan')):.1f} pairs/s); the round-3 form of the GRU launches
(`single_chain_gru`, hoist_context=False) in the same run: {d.get('single_chain_gru', {}).get('value', float('nan')):.1f} pairs/s; bf16x6
{d['split_bf16']['bf16x6']['value']:.1f} (EPE {d['split_bf16']['bf16x6']['epe_mean']:.2e}), bf16x3 {d['split_bf16']['bf16x3']['value']:.1f} (EPE
{d['split_bf16']['bf16x3']['epe_mean']:.2e}); skip_dead_upsample {d['skip_dead_upsample']['value']:.1f} (identical output:
{d['skip_dead_upsample']['identical_output']}); gma fp32 {c3['gma_fp32']['value']:.1f} (EPE {c3['gma_fp32'].get('epe_mean', float('nan')):.2e}), raft bf16
{c3['raft_bf16']['value']:.1f}, gma bf16 {c3['gma_bf16']['value']:.1f}; SEA-RAFT correlation path fp32 {c3['sea_raft_corr_f32']['iters4']['value']:.0f} /
{c3['sea_raft_corr_f32']['iters12']['value']:.0f} pairs/s (4 / 12 lookups; max error {c3['sea_raft_corr_f32']['err_vs_cpu_fp32']['max_abs']:.1e}), bf16
{c3['sea_raft_corr_bf16']['iters4']['value']:.0f} / {c3['sea_raft_corr_bf16']['iters12']['value']:.0f}; config 4 (KITTI 375x1242, batch 8)
{d['config4']['value']:.1f} pairs/s (EPE {d['config4']['epe_vs_cpu']['mean']:.2e}); train {d['train']['value']:.1f} samples/s
({d['train']['ms_per_step']:.1f} ms/step, {d['train'].get('launches_per_step')} launches); cpu_baseline {d['cpu_baseline']['value']:.2f} pairs/s
({d['cpu_baseline']['cores']} cores, kind {d['cpu_baseline']['kind']}).

Per-launch table of the instrumented forward (HIP events around every update-block convolution; mk / c1 / c2 carry the side
stream's overlap at batch 8): {json.dumps(d.get('kernels'))}

Micro-benches of the same run — correlation path (y_corr.log):
```
{tail('y_corr.log', 40)}
```
lookup / on-demand correlation (y_lookup.log):
```
{cut(tail('y_lookup.log', 13), 420)}
```
update-block convolutions, batch 8, 3 rounds round-robin, heuristic vs 64x64 x3 everywhere (y_conv_b8.log):
```
{tail('y_conv_b8.log', 13)}
```
batch 1 (y_conv_b1.log):
```
{tail('y_conv_b1.log', 13)}
```
the same launches on the split-bf16 kernels: heuristic vs round 2's tiles, three / two / one plane (y_conv_bf_b8.log):
```
{cut(tail('y_conv_bf_b8.log', 13), 460)}
```
weight gradient (y_wgrad.log):
```
{tail('y_wgrad.log', 14)}
```

"""
    open(P, "w").write(head)
    T = [sys.executable, os.path.join(ROOT, "scripts", "trace_stats.py")]
    for name, fw, top, title in (
            ("y_tr_f32", 5, 24, "raft fp32 (default bench command), batch 8, 5 forwards"),
            ("y_tr_b1", 13, 18, "raft fp32, batch 1, 13 forwards"),
            ("y_tr_x6", 5, 14, "raft with conv_precision=bf16x6 (split-bf16 kernels K8, profiles/r03_d, unchanged kernels), batch 8, 5 forwards"),
            ("y_tr_seam", 8, 24, "drop-in seam path: SeamRAFT + patch.accelerate (B1/B3/B4/B5), batch 1, 8 forwards"),
            ("y_tr_seam_torch", 3, 24, "the same object un-patched: stock PyTorch-ROCm ops (MIOpen / rocBLAS / grid_sample), batch 1, the FIRST 3 forwards of the process — MIOpen still runs its naive fallback convolution for the 7x7 / 1x5 / 5x1 shapes while it searches; warmed up the same forward takes 41 ms (bench.py --torch-baseline: 24.3 pairs/s)"),
            ("y_tr_train", 4, 30, "training step (BASELINE config 5 shape: batch 10, 368x496, 12 iterations), 4 steps incl. backward + AdamW")):
        if os.path.isdir(os.path.join(O, name)):
            subprocess.run(T + [os.path.join(O, name), "--forwards", str(fw), "--top", str(top), "--title", title, "--out", P],
                           check=True, stdout=subprocess.DEVNULL)
    # per-launch PMC traffic of the update block, keyed by position after the lookup (scripts/pmc_extract.py), + source-hash stamp
    tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if all(os.path.isdir(os.path.join(O, n)) for n in ("y_pmc_fetch", "y_pmc_write", "y_pmc_sq")):
        subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pmc_extract.py"), "--fetch", os.path.join(O, "y_pmc_fetch"),
                        "--write", os.path.join(O, "y_pmc_write"), "--sq", os.path.join(O, "y_pmc_sq"), "--batch", "8", "--out", tj],
                       check=True, stdout=subprocess.DEVNULL)
    doc = json.load(open(tj))
    sys.path.insert(0, ROOT)
    from ptlflow_amd import _build
    doc["kernel_source_sha16"] = _build.source_hash()
    doc["note_r04"] = ("round 4: every @b8 entry re-measured on the final tree (scripts/gpu_final_r04.sh: gpurun_out/y_pmc_fetch, y_pmc_write, y_pmc_sq); "
                       "zr / q are the launches with the context slice hoisted out (K = 1280).  `kernel_source_sha16` is now the stamp both libraries "
                       "carry (ptlflow_amd/_build.py::source_hash: every file of csrc/ + include/pfk.h + the flags), so a change to ANY kernel source "
                       "marks roofline.traffic `stale` in bench.py.")
    json.dump(doc, open(tj, "w"), indent=1)
    print(P, os.path.getsize(P), "bytes;", tj, "stamped", doc["kernel_source_sha16"])


if __name__ == "__main__":
    main()
