#!/usr/bin/env python3
"""profiles/r05_a (kernel traces, per-call-site table, counters and the bench line of the same run) and profiles/pmc_traffic.json
(per-launch HBM traffic, hash-stamped) of the round-5 final tree from the raw rocprofv3 output of scripts/gpu_final_r05.sh in
gpurun_out/ (CPU only).

    python scripts/make_profiles_r05.py
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles", "r05_a_kerneltrace_final.md")


def bench_line():
    for line in reversed(open(os.path.join(O, "z_bench.log")).read().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit("no JSON line in z_bench.log")


def cut(text, n):
    return "\n".join(l[:n] for l in text.splitlines())


def tail(name, n=3):
    try:
        lines = [l for l in open(os.path.join(O, name)).read().strip().splitlines() if "amdgpu.ids" not in l and "Warning" not in l]
        return "\n".join(lines[-n:])
    except OSError:
        return "(missing)"


def suite():
    try:
        lines = [l for l in open(os.path.join(O, "z_pytest.log")).read().splitlines() if " passed" in l or " failed" in l]
        return lines[-1].strip() if lines else "(no summary line)"
    except OSError:
        return "(missing)"


def g(d, *path, fmt="{:.1f}", default="n/a"):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    try:
        return fmt.format(d)
    except (ValueError, TypeError):
        return str(d)


def main():
    d = bench_line()
    r, rl = d["roofline"], d.get("roofline_lookup", {})
    c3 = d.get("config3", {})
    dk = d.get("dropin", {})
    head = f"""# r05_a — round 5, final tree: kernel traces, per-call-site table, counters and the bench line of the same run (MI355X, one GPU)

Commands (`scripts/gpu_final_r05.sh`, one gpurun call): the whole GPU suite, `__graft_entry__.smoke()`, the driver's command
`python3 bench.py --gpus 1 --steps 20 --warmup 5`, micro-benches, then from /tmp with TMPDIR=/tmp
`rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline --no-roofline --no-split-modes --no-extra-legs
--no-batch1 [--batch 1] --steps K --warmup W`, `... -- python scripts/seam_prof.py` (the drop-in seam path: the reference's own
`ptlflow.models.raft.raft.RAFT` out of the staged archive + patch.accelerate, batch 1, 8 forwards), `... -- python scripts/train_prof.py`
(4 training steps, batch 10, 368x496, 12 iterations), and separate `--kernel-trace --pmc <counter>` passes of the batch-8 command
(FETCH_SIZE, WRITE_SIZE, the SQ busy set; FETCH_SIZE once more with `PFK_VOLUME_LAYOUT=rowmajor`).  Kernel tables by
scripts/trace_stats.py (regs = VGPRs + AGPRs per dispatch; scratch must read 0 everywhere — tests/test_no_scratch.py); **per-call-site
tables by scripts/callsite_stats.py** — the convolutions of an iteration share kernel instantiations, so a launch is identified by
its position after the iteration's `lookup_kernel` dispatch (per queue): `roofline.avg_us` of the bench line can be read off the
`fm` row.  Whole GPU suite of this run (gpurun_out/z_pytest.log): `{suite()}`.

**Bench line of this run** (gpurun_out/z_bench.log): **{d['value']:.1f} frame-pairs/s** fp32 ({d['ms_per_step']:.1f} ms/step, batch 8), EPE vs
the reference's CPU forward {g(d, 'epe_vs_cpu', 'mean', fmt='{:.2e}')} mean / {g(d, 'epe_vs_cpu', 'max', fmt='{:.2e}')} max, stream-K faults {d['streamk_faults']}; roofline
{r['kernel']}: {r['avg_us']:.1f} us = {r['achieved']:.1f} TFLOP/s = **{r['frac']:.3f}** of {r['peak']}; lookup (K3, HBM-bound) in situ
{g(rl, 'avg_us')} us = {g(rl, 'achieved', fmt='{:.0f}')} GB/s of algorithmic bytes = **{g(rl, 'frac', fmt='{:.3f}')}** of 8000; batch 1 {g(d, 'batch1', 'value')}, batch 16 {g(d, 'batch16', 'value')};
model_benchmark protocol {g(d, 'model_benchmark_protocol', 'value')} pairs/s ({g(d, 'model_benchmark_protocol', 'ms_median', fmt='{:.2f}')} ms median) on the mirror,
**{g(dk, 'value')} on the drop-in seam path, model class {dk.get('model_class')}** ({g(dk, 'ms_median', fmt='{:.2f}')} ms, EPE
{g(dk, 'epe_vs_cpu', 'mean', fmt='{:.2e}')}; batch 8: {g(dk, 'batch8', 'value')} pairs/s); **the same object with `skip_dead_upsample=True`: {g(dk, 'skip_dead', 'value')}
pairs/s ({g(dk, 'skip_dead', 'ms_median', fmt='{:.2f}')} ms; batch 8: {g(dk, 'skip_dead', 'batch8', 'value')}; identical flows: {dk.get('skip_dead', {}).get('identical_flows')})**; the single-chain form of the
GRU launches (`hoist_context=False`): {g(d, 'single_chain_gru', 'value')} pairs/s; bf16x6 {g(d, 'split_bf16', 'bf16x6', 'value')} (EPE
{g(d, 'split_bf16', 'bf16x6', 'epe_mean', fmt='{:.2e}')}), bf16x3 {g(d, 'split_bf16', 'bf16x3', 'value')} (EPE {g(d, 'split_bf16', 'bf16x3', 'epe_mean', fmt='{:.2e}')}); the mirror's skip_dead_upsample
{g(d, 'skip_dead_upsample', 'value')} (identical output: {d.get('skip_dead_upsample', {}).get('identical_output')}); gma fp32 {g(c3, 'gma_fp32', 'value')} (EPE
{g(c3, 'gma_fp32', 'epe_mean', fmt='{:.2e}')}), raft bf16 {g(c3, 'raft_bf16', 'value')}, gma bf16 {g(c3, 'gma_bf16', 'value')}; sea_raft_s whole model
{g(c3, 'sea_raft_s_full', 'value')} pairs/s (EPE {g(c3, 'sea_raft_s_full', 'epe_vs_cpu', 'mean', fmt='{:.2e}')}); **ccmr whole model (defaults, `alternate_corr=True`) {g(c3, 'ccmr_full', 'value')}
pairs/s, EPE vs its own CPU forward at 436x1024 {g(c3, 'ccmr_full', 'epe_vs_cpu', 'mean', fmt='{:.2e}')} mean / {g(c3, 'ccmr_full', 'epe_vs_cpu', 'max', fmt='{:.2e}')} max; ms_raft_p
{g(c3, 'ms_raft_p_full', 'value')} pairs/s, EPE {g(c3, 'ms_raft_p_full', 'epe_vs_cpu', 'mean', fmt='{:.2e}')} / {g(c3, 'ms_raft_p_full', 'epe_vs_cpu', 'max', fmt='{:.2e}')}**; config 4 (KITTI 375x1242, batch 8)
{g(d, 'config4', 'value')} pairs/s (EPE {g(d, 'config4', 'epe_vs_cpu', 'mean', fmt='{:.2e}')}); train {g(d, 'train', 'value')} samples/s
({g(d, 'train', 'ms_per_step')} ms/step, {d.get('train', {}).get('launches_per_step')} launches); cpu_baseline {g(d, 'cpu_baseline', 'value', fmt='{:.2f}')} pairs/s
({d.get('cpu_baseline', {}).get('cores')} cores, kind {d.get('cpu_baseline', {}).get('kind')}).

Per-launch table of the instrumented forward (HIP events around every update-block convolution; mk / c1 carry the side
stream's overlap at batch 8): {json.dumps(d.get('kernels'))}

Encoders, un-profiled (z_enc_time.log):
```
{tail('z_enc_time.log', 2)}
```
Micro-benches of the same run — correlation path, row-major and blocked (z_corr.log):
```
{tail('z_corr.log', 46)}
```
lookup on both volume layouts, 4 / 8 pixels per workgroup (z_lookup_blocked.log):
```
{cut(tail('z_lookup_blocked.log', 8), 400)}
```
fused mask conv2 + softmax + convex upsampling against the two launches it replaces (z_maskup.log):
```
{tail('z_maskup.log', 5)}
```
update-block convolutions, batch 8, 3 rounds round-robin, heuristic vs 64x64 x3 everywhere (z_conv_b8.log):
```
{tail('z_conv_b8.log', 18)}
```
batch 1 (z_conv_b1.log):
```
{tail('z_conv_b1.log', 18)}
```
the encoders' convolutions at fnet's batch-8 size (16 images): heuristic / 64x64 x3 / 128x96 on the 3-stage / on the 2-stage kernel (z_conv_enc.log):
```
{cut(tail('z_conv_enc.log', 9), 420)}
```

"""
    open(P, "w").write(head)
    C = [sys.executable, os.path.join(ROOT, "scripts", "callsite_stats.py")]
    for name, title, pmc in (
            ("z_tr_f32", "per call site, batch 8, blocked volume layout (the default): kernel trace", []),
            ("z_tr_f32_row", "per call site, batch 8, `PFK_VOLUME_LAYOUT=rowmajor`: kernel trace", []),
            ("z_tr_b1", "per call site, batch 1: kernel trace", []),
            ("z_pmc_fetch", "per call site, batch 8, blocked layout: counters (separate --pmc passes; FETCH_SIZE / WRITE_SIZE in KB as reported — "
             "FETCH_SIZE x2 per the gfx950 correction of MI355X_MICROARCH.md for these kernels' 128-byte requests; durations of the serialised PMC pass)",
             ["z_pmc_fetch", "z_pmc_write", "z_pmc_sq"]),
            ("z_pmc_fetch_row", "per call site, batch 8, row-major layout: FETCH_SIZE", ["z_pmc_fetch_row"])):
        if os.path.isdir(os.path.join(O, name)):
            subprocess.run(C + [os.path.join(O, name), "--title", title, "--out", P] + (["--pmc"] + [os.path.join(O, p) for p in pmc] if pmc else []),
                           check=True, stdout=subprocess.DEVNULL)
    T = [sys.executable, os.path.join(ROOT, "scripts", "trace_stats.py")]
    for name, fw, top, title in (
            ("z_tr_f32", 5, 24, "raft fp32 (default bench command), batch 8, 5 forwards"),
            ("z_tr_b1", 13, 18, "raft fp32, batch 1, 13 forwards"),
            ("z_tr_seam", 8, 24, "drop-in seam path: the reference's own ptlflow.models.raft.raft.RAFT + patch.accelerate (B1/B3/B4/B5), batch 1, 8 forwards"),
            ("z_tr_train", 4, 30, "training step (BASELINE config 5 shape: batch 10, 368x496, 12 iterations), 4 steps incl. backward + AdamW")):
        if os.path.isdir(os.path.join(O, name)):
            subprocess.run(T + [os.path.join(O, name), "--forwards", str(fw), "--top", str(top), "--title", title, "--out", P],
                           check=True, stdout=subprocess.DEVNULL)
    tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if all(os.path.isdir(os.path.join(O, n)) for n in ("z_pmc_fetch", "z_pmc_write", "z_pmc_sq")):
        subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pmc_extract.py"), "--fetch", os.path.join(O, "z_pmc_fetch"),
                        "--write", os.path.join(O, "z_pmc_write"), "--sq", os.path.join(O, "z_pmc_sq"), "--batch", "8", "--out", tj],
                       check=True, stdout=subprocess.DEVNULL)
    doc = json.load(open(tj))
    sys.path.insert(0, ROOT)
    from ptlflow_amd import _build
    doc["kernel_source_sha16"] = _build.source_hash()
    doc["note_r05"] = ("round 5: every @b8 entry re-measured on the final tree (scripts/gpu_final_r05.sh: gpurun_out/z_pmc_fetch, z_pmc_write, z_pmc_sq); "
                       "`lookup@b8` is K3 on the blocked 4x8 volume layout (row-major, same run: profiles/r05_a, PFK_VOLUME_LAYOUT=rowmajor pass).")
    json.dump(doc, open(tj, "w"), indent=1)
    print(P, os.path.getsize(P), "bytes;", tj, "stamped", doc["kernel_source_sha16"])


if __name__ == "__main__":
    main()
