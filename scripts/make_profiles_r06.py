#!/usr/bin/env python3
"""profiles/r06_a (kernel traces, per-call-site table, counters and the bench line of the same run) and profiles/pmc_traffic.json
(per-launch HBM traffic, hash-stamped) of the round-6 final tree from the raw rocprofv3 output of scripts/gpu_final_r06.sh in
gpurun_out/ (CPU only).

    python scripts/make_profiles_r06.py
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles", "r06_a_kerneltrace_final.md")


def bench_line():
    for line in reversed(open(os.path.join(O, "z6_bench.log")).read().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit("no JSON line in z6_bench.log")


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
        lines = [l for l in open(os.path.join(O, "z6_pytest.log")).read().splitlines() if " passed" in l or " failed" in l]
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
    rb = d.get("roofline_bf16", {})
    head = f"""# r06_a — round 6, final tree: kernel traces, per-call-site table, counters and the bench line of the same run (MI355X, one GPU)

Commands (`scripts/gpu_final_r06.sh`, one gpurun call): the whole GPU suite, `__graft_entry__.smoke()`, the driver's command
`python3 bench.py --gpus 1 --steps 20 --warmup 5`, micro-benches, then from /tmp with TMPDIR=/tmp
`rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline --no-roofline --no-split-modes --no-extra-legs
--no-batch1 [--batch 1] [--conv-precision bf16] --steps K --warmup W` (fp32 batch 8 / batch 1, **the K8b bf16-storage forward at batch 8 /
batch 1**), `... -- python scripts/seam_prof.py` (the drop-in seam path: the reference's own `ptlflow.models.raft.raft.RAFT` out of the
staged archive + patch.accelerate, batch 1, 8 forwards), `... -- python scripts/train_prof.py` (4 training steps, batch 10, 368x496, 12
iterations), and separate `--kernel-trace --pmc <counter>` passes of the batch-8 commands, fp32 and bf16 (FETCH_SIZE, WRITE_SIZE, the SQ
busy set).  Kernel tables by scripts/trace_stats.py (regs = VGPRs + AGPRs per dispatch; scratch must read 0 everywhere —
tests/test_no_scratch.py); **per-call-site tables by scripts/callsite_stats.py** — the convolutions of an iteration share kernel
instantiations, so a launch is identified by its position after the iteration's `lookup_kernel` dispatch (per queue; at batch 1 the
grouped launch `conv_gemm_v3_group_kernel` = convc1 | convf2 | the previous iteration's mask conv2 is a row of its own):
`roofline.avg_us` of the bench line can be read off the `fm` row, `roofline_bf16.avg_us` off the bf16 table's.  Whole GPU suite of this
run (gpurun_out/z6_pytest.log): `{suite()}`.

**Bench line of this run** (gpurun_out/z6_bench.log): **{d['value']:.1f} frame-pairs/s** fp32 ({d['ms_per_step']:.1f} ms/step, batch 8), EPE vs
the reference's CPU forward {g(d, 'epe_vs_cpu', 'mean', fmt='{:.2e}')} mean / {g(d, 'epe_vs_cpu', 'max', fmt='{:.2e}')} max, stream-K faults {d['streamk_faults']}; roofline
{r['kernel']}: {r['avg_us']:.1f} us = {r['achieved']:.1f} TFLOP/s = **{r['frac']:.3f}** of {r['peak']}; lookup (K3, HBM-bound) in situ
{g(rl, 'avg_us')} us = {g(rl, 'achieved', fmt='{:.0f}')} GB/s of algorithmic bytes = **{g(rl, 'frac', fmt='{:.3f}')}** of 8000; batch 1 {g(d, 'batch1', 'value')}, batch 16 {g(d, 'batch16', 'value')};
model_benchmark protocol {g(d, 'model_benchmark_protocol', 'value')} pairs/s ({g(d, 'model_benchmark_protocol', 'ms_median', fmt='{:.2f}')} ms median; driver: {d.get('model_benchmark_protocol', {}).get('driver')}) on the mirror,
**{g(dk, 'value')} on the drop-in seam path — plain `accelerate(model)` (dead work skipped by default, K13 behind seam B5), model class
{dk.get('model_class')}, driver: {dk.get('driver')}** ({g(dk, 'ms_median', fmt='{:.2f}')} ms, EPE {g(dk, 'epe_vs_cpu', 'mean', fmt='{:.2e}')}; batch 8: {g(dk, 'batch8', 'value')} pairs/s;
identical flows: {dk.get('identical_flows')}); the same class with `skip_dead_upsample=False` (every iteration keeps its mask head and upsampling, as the
reference): {g(dk, 'every_iteration', 'value')} pairs/s ({g(dk, 'every_iteration', 'ms_median', fmt='{:.2f}')} ms; batch 8: {g(dk, 'every_iteration', 'batch8', 'value')}); the single-chain form of the
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

**`roofline_bf16`** (the K8b forward, BASELINE config 3's precision; HIP events, in situ): {rb.get('kernel')} {g(rb, 'avg_us')} us = {g(rb, 'achieved')}
{rb.get('unit')} = **{g(rb, 'frac', fmt='{:.3f}')}** of {rb.get('peak')} (bound: {rb.get('bound')}; floors {json.dumps(rb.get('floor_us'))} us); its lookup (bf16 maps, paired
fetch, bf16 rows): {g(rb, 'lookup', 'avg_us')} us = {g(rb, 'lookup', 'achieved', fmt='{:.0f}')} GB/s of {g(rb, 'lookup', 'algorithmic_bytes_per_launch', fmt='{:.0f}')} algorithmic bytes; per call site:
{json.dumps(rb.get('kernels'))}

Encoders, un-profiled (z6_enc_time.log; fp32) and the stages of the bf16-storage forward (z6_stage_bf16.log — its "corr volume" figure is
the fp32 volume's: the script builds a default CorrBlock):
```
{tail('z6_enc_time.log', 2)}
{tail('z6_stage_bf16.log', 2)}
```
Batch 1, the grouped launch of the motion encoder off / on, two models each (z6_batch1.log; with the dead work skipped: z6_batch1_skip.log):
```
{tail('z6_batch1.log', 5)}
{tail('z6_batch1_skip.log', 5)}
```
Micro-benches of the same run — correlation path, row-major and blocked (z6_corr.log):
```
{tail('z6_corr.log', 46)}
```
lookup on both volume layouts, 4 / 8 pixels per workgroup, and on bf16 maps (paired fetch) with fp32 / bf16 rows out (z6_lookup_blocked.log):
```
{cut(tail('z6_lookup_blocked.log', 12), 400)}
```
fused mask conv2 + softmax + convex upsampling against the two launches it replaces — fp32 (K13, z6_maskup.log) and K8b (K13b, z6_maskup_b16.log):
```
{tail('z6_maskup.log', 5)}
{tail('z6_maskup_b16.log', 2)}
```
convf1 (7x7 on the flow): the tiled VALU kernel against the MFMA kernel (z6_cin2.log):
```
{tail('z6_cin2.log', 4)}
```
update-block convolutions (fp32), batch 8, 3 rounds round-robin, the library's choice (z6_conv_b8.log):
```
{tail('z6_conv_b8.log', 18)}
```

"""
    open(P, "w").write(head)
    C = [sys.executable, os.path.join(ROOT, "scripts", "callsite_stats.py")]
    for name, title, pmc in (
            ("z6_tr_f32", "per call site, fp32, batch 8 (blocked volume layout): kernel trace", []),
            ("z6_tr_b1", "per call site, fp32, batch 1 (grouped launch of the motion encoder): kernel trace", []),
            ("z6_tr_bf16", "per call site, K8b bf16-storage forward, batch 8: kernel trace", []),
            ("z6_tr_bf16_b1", "per call site, K8b bf16-storage forward, batch 1: kernel trace", []),
            ("z6_pmc_fetch", "per call site, fp32, batch 8: counters (separate --pmc passes; FETCH_SIZE / WRITE_SIZE in KB as reported — "
             "FETCH_SIZE x2 per the gfx950 correction of MI355X_MICROARCH.md for these kernels' 128-byte requests; durations of the serialised PMC pass)",
             ["z6_pmc_fetch", "z6_pmc_write", "z6_pmc_sq"]),
            ("z6_pmc_fetch_bf16", "per call site, K8b bf16-storage forward, batch 8: counters (same passes with --conv-precision bf16)",
             ["z6_pmc_fetch_bf16", "z6_pmc_write_bf16", "z6_pmc_sq_bf16"])):
        if os.path.isdir(os.path.join(O, name)):
            subprocess.run(C + [os.path.join(O, name), "--title", title, "--out", P] + (["--pmc"] + [os.path.join(O, p) for p in pmc] if pmc else []),
                           check=True, stdout=subprocess.DEVNULL)
    T = [sys.executable, os.path.join(ROOT, "scripts", "trace_stats.py")]
    for name, fw, top, title in (
            ("z6_tr_f32", 5, 24, "raft fp32 (default bench command), batch 8, 5 forwards"),
            ("z6_tr_b1", 13, 18, "raft fp32, batch 1, 13 forwards"),
            ("z6_tr_bf16", 8, 24, "raft, K8b bf16 activation storage (--conv-precision bf16), batch 8, 8 forwards"),
            ("z6_tr_bf16_b1", 13, 20, "raft, K8b bf16 activation storage, batch 1, 13 forwards"),
            ("z6_tr_seam", 8, 24, "drop-in seam path: the reference's own ptlflow.models.raft.raft.RAFT + patch.accelerate (B1/B3/B4/B5), batch 1, 8 forwards"),
            ("z6_tr_train", 4, 30, "training step (BASELINE config 5 shape: batch 10, 368x496, 12 iterations), 4 steps incl. backward + AdamW")):
        if os.path.isdir(os.path.join(O, name)):
            subprocess.run(T + [os.path.join(O, name), "--forwards", str(fw), "--top", str(top), "--title", title, "--out", P],
                           check=True, stdout=subprocess.DEVNULL)
    tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if all(os.path.isdir(os.path.join(O, n)) for n in ("z6_pmc_fetch", "z6_pmc_write", "z6_pmc_sq")):
        subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pmc_extract.py"), "--fetch", os.path.join(O, "z6_pmc_fetch"),
                        "--write", os.path.join(O, "z6_pmc_write"), "--sq", os.path.join(O, "z6_pmc_sq"), "--batch", "8", "--out", tj],
                       check=True, stdout=subprocess.DEVNULL)
    if all(os.path.isdir(os.path.join(O, n)) for n in ("z6_pmc_fetch_bf16", "z6_pmc_write_bf16", "z6_pmc_sq_bf16")):
        # the same passes of the K8b forward: entries `<site>_bf16@b8` (bench.py's roofline_bf16.traffic)
        subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pmc_extract.py"), "--fetch", os.path.join(O, "z6_pmc_fetch_bf16"),
                        "--write", os.path.join(O, "z6_pmc_write_bf16"), "--sq", os.path.join(O, "z6_pmc_sq_bf16"), "--batch", "8", "--tag", "_bf16",
                        "--out", tj], check=True, stdout=subprocess.DEVNULL)
    doc = json.load(open(tj))
    sys.path.insert(0, ROOT)
    from ptlflow_amd import _build
    doc["kernel_source_sha16"] = _build.source_hash()
    doc["note_r06"] = ("round 6: every @b8 entry re-measured on the final tree (scripts/gpu_final_r06.sh: gpurun_out/z6_pmc_fetch, z6_pmc_write, z6_pmc_sq); "
                       "`lookup@b8` is K3 on the blocked 4x8 volume layout.")
    json.dump(doc, open(tj, "w"), indent=1)
    print(P, os.path.getsize(P), "bytes;", tj, "stamped", doc["kernel_source_sha16"])


if __name__ == "__main__":
    main()
