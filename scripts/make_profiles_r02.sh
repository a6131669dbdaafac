#!/bin/bash
# profiles/r02_c (kernel traces) and profiles/pmc_traffic.json of the round-2 final tree from the raw rocprofv3 output of
# scripts/gpu_final3.sh in gpurun_out/ (run from the repo root, CPU only)
O=gpurun_out
P=profiles/r02_c_kerneltrace_final.md
{
echo "# r02_c — round 2, final tree: kernel traces (MI355X, one GPU)"
echo
echo "Commands (scripts/gpu_final3.sh): \`rocprofv3 --kernel-trace --output-format csv -- python bench.py --no-cpu-baseline --no-roofline"
echo "--no-split-modes --no-extra-legs --no-batch1 [--batch 1 | --model gma --batch 4 | --model raft_small | --conv-precision bf16x3 |"
echo "--conv-precision bf16] --steps K --warmup W\` and \`... -- python scripts/train_prof.py\` (4 training steps of RAFT, batch 10, 368x496,"
echo "12 iterations).  Summaries by scripts/trace_stats.py (regs = VGPR count per dispatch; scratch must read 0 everywhere — tests/test_no_scratch.py)."
echo "Same box, same tree, \`python bench.py\` (gpurun_out/g_bench.log): **69.4 frame-pairs/s fp32 (115.4 ms/step, batch 8), roofline fm 534.8 us ="
echo "124.2 TFLOP/s = 0.790 of 157.3; batch-1 53.2, model_benchmark protocol 52.6 (19.02 ms median); bf16x6 88.4 (EPE 1.07e-5), bf16x3 138.3"
echo "(EPE 6.8e-5); skip_dead_upsample 72.7 (bit-identical); gma fp32 46.3 (batch 4); raft bf16 190.1, gma bf16 105.2; train 86.3 samples/s"
echo "(115.9 ms/step, encoders 37.9 ms); cpu_baseline 0.64 pairs/s (16 cores)**; EPE vs the CPU oracle 1.02e-5 mean / 5.7e-5 max.  The whole GPU"
echo "suite on the final tree: 236 passed, 3 skipped (scripts/gpu_confirm.sh, gpurun_out/c_pytest.log; earlier confirm runs read up to 70.3 pairs/s / 0.806 on"
echo "another box of the pool).  Micro-benches of the same run: g_corr.log (K1 fp32 1908 us = 106 TF, K1 bf16 315 us = 2.52 TB/s, K2 488 us ="
echo "5.3 TB/s, K3 59.1 us = 2.77 TB/s), g_lookup.log (K7 level 0: 348.3 us at batch 8 = 16.6 TB/s of L2 gathers), g_conv_b1.log, g_conv_b8.log,"
echo "g_wgrad.log.  In the fp32 tables the mask head's second convolution and the convex upsampling run on a second stream (batch 8, gma)."
echo
} > $P
T="python scripts/trace_stats.py"
$T $O/g_tr_f32 --forwards 5 --top 24 --title "raft fp32 (default bench command), batch 8, 5 forwards" --out $P > /dev/null
$T $O/g_tr_b1 --forwards 13 --top 18 --title "raft fp32, batch 1 (13 forwards)" --out $P > /dev/null
$T $O/g_tr_gma --forwards 5 --top 18 --title "gma fp32, batch 4, 5 forwards (attention map, aggregation GEMM and the 512-input GRU on libpfk)" --out $P > /dev/null
$T $O/g_tr_small --forwards 5 --top 18 --title "raft_small fp32, batch 8, 5 forwards (SmallEncoder, ConvGRU, upflow8 on libpfk)" --out $P > /dev/null
$T $O/g_tr_x3 --forwards 5 --top 12 --title "bf16x3 split convolutions, batch 8, 5 forwards" --out $P > /dev/null
$T $O/g_tr_bf16 --forwards 5 --top 16 --title "bf16 operands + bf16 correlation volume (BASELINE config 3 precision), batch 8, 5 forwards" --out $P > /dev/null
$T $O/g_tr_train --forwards 4 --top 30 --title "training step (BASELINE config 5 shape: batch 10, 368x496, 12 iterations), 4 steps incl. backward + AdamW" --out $P > /dev/null
python scripts/pmc_extract.py --fetch $O/g_pmc_fetch --write $O/g_pmc_write --sq $O/g_pmc_sq --batch 8 --out profiles/pmc_traffic.json > /dev/null
wc -l $P
python - <<'PY'
import json, subprocess
cur = json.load(open("profiles/pmc_traffic.json"))["entries"]
old = json.loads(subprocess.check_output("python scripts/pmc_extract.py --fetch gpurun_out/f3_pmc_fetch --write gpurun_out/f4_pmc_write --batch 8", shell=True))
M = 8 * 55 * 128
alg = {"c1": (324, 1, 256, 256, "convc1 1x1 324->256"), "c2": (256, 9, 192, 192, "convc2 3x3 256->192"),
       "f2": (128, 9, 64, 64, "convf2 3x3 128->64 (stream-K at batch 8: + 8 MB of partial tiles written and read back)"),
       "cv": (256, 9, 126, 128, "conv 3x3 256->126 (+2 flow channels written by the same launch)"),
       "zr1": (384, 5, 256, 256, "convz1+convr1 1x5, hx 384 -> z, r*h"), "q1": (384, 5, 128, 128, "convq1 1x5 + GRU update"),
       "zr2": (384, 5, 256, 256, "convz2+convr2 5x1"), "q2": (384, 5, 128, 128, "convq2 5x1 + GRU update"),
       "fm": (128, 9, 512, 512, "flow-head conv1 + mask conv1 fused, 3x3 128->512"), "mk": (256, 1, 576, 576, "mask conv2 1x1 256->576")}
rows = []
for k, (cin, taps, cout, cw, note) in alg.items():
    x, o = cur[f"{k}@b8"], old[f"{k}@b8"]
    inb, outb, wb = M * cin * 4 / 1024, M * cw * 4 / 1024, cin * taps * cout * 4 / 1024
    fl = 2 * M * cin * taps * cout
    f2x = 2 * x["fetch_kb"]
    rows.append(f"| {k} | {note} | {x['avg_us']:.1f} | {fl / x['avg_us'] / 1e6:.1f} | {inb + wb:.0f} | {x['fetch_kb']} | {f2x} | {f2x / (inb + wb):.2f} | {outb:.0f} | "
                f"{o['write_kb']} ({o['write_kb'] / outb:.2f}x) | {x['write_kb']} ({x['write_kb'] / outb:.2f}x) | {(f2x + x['write_kb']) * 1024 / x['avg_us'] / 1e6:.2f} | "
                f"{x.get('sq_wait_any_frac', '')} | {x.get('mfma_busy_per_sq_busy', '')} |")
lk = cur["lookup@b8"]
out = ["# r02_b — round 2: PMC traffic per launch of the update-block kernels (batch 8, MI355X, final tree)", "",
"Commands (scripts/gpu_final3.sh; earlier passes gpu_final2.sh, gpu_r2r.sh, gpu_r2u.sh): separate passes, each `rocprofv3 --kernel-trace --pmc",
"<counters> --output-format csv -- python bench.py --no-cpu-baseline --no-roofline --no-split-modes --no-extra-legs --no-batch1 --steps 1",
"--warmup 1` from /tmp with TMPDIR=/tmp: `FETCH_SIZE` (gpurun_out/g_pmc_fetch), `WRITE_SIZE` (g_pmc_write; *before* = gpurun_out/f4_pmc_write,",
"commit d1de125, ahead of the scratch fix below), `TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum` (u_pmc_wrreq), `SQ_VALU_MFMA_BUSY_CYCLES",
"SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE` (g_pmc_sq).  Extraction: scripts/pmc_extract.py keys",
"every launch by its POSITION after the iteration's lookup kernel (c1 c2 f1* f2 cv zr1 q1 zr2 q2 fm mk; f1 is the 2-channel direct kernel) and",
"averages over the launches of the pass.  FETCH_SIZE is shown raw and with the guide's gfx950 correction (x2: the counter tallies 128-byte",
"requests at 64 B for 16-byte-per-lane loads, which is what these kernels issue).  KB = 1024 B; M = 8 x 55 x 128 = 56 320 pixels; algorithmic",
"input = activations + packed weights, read once.  `profiles/pmc_traffic.json` (what bench.py's `roofline.traffic` reads) holds the same numbers.", "",
"| launch | what | avg us (fetch pass) | TFLOP/s | algorithmic input KB | FETCH KB raw | FETCH KB x2 | x2 / input | algorithmic output KB | WRITE KB before | WRITE KB now | HBM-side TB/s | SQ_WAIT_ANY frac | MFMA busy / SQ busy |",
"|---|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|"] + rows + ["",
f"lookup_kernel<4,4,float>: {lk['avg_us']} us, FETCH {lk['fetch_kb']} KB raw, WRITE {lk['write_kb']} KB (algorithmic output 56 320 x 324 x 4 B = 71 280 KB: 1.00x; its",
"loads are 4-byte gathers from a 2.1 GB level-0 volume + three pooled levels, a width the x2 correction is not calibrated for).", "",
"Correlation volume and pyramid of the same passes (once per forward, batch 8; identified by grid size): K1 `pfk_corr_volume_f32` (96 800 blocks of",
"64x64, 16x16 supertile walk) FETCH 416.7 MB raw = **0.83 GB x2 against 0.115 GB of operands (7.2x; round 1: 6.4 GB = 53x)**, WRITE 1 548 800 KB =",
"1.586 GB = 8 x 7040^2 x 4 B exactly (1.00x), 1.9-2.2 ms under the profiler (scripts/corr_bench.py: 2.03 ms = 100 TFLOP/s): the kernel's traffic is",
"its own output; the operand re-reads left (each 1024-row supertile reads both 1 MB panels once) are half of that and not the limiter.",
"K2 `pool2x2_kernel<float>` level 0 -> 1: FETCH x2 = 1.557 GB (the volume, once), WRITE 389 MB, 403-420 us = 4.7 TB/s.", "",
"Reading the table:", "",
"* **Write: calibrated, and a bug found with it.**  WRITE_SIZE = 64 B x TCC_EA0_WRREQ on this chip (the raw pass shows WRREQ == WRREQ_64B on every",
"  kernel: the L2 only ever issues full 64-byte write requests here), and the GRU epilogues and the lookup reproduce their algorithmic bytes to",
"  0.1 % — the counter can be trusted for this access pattern.  The LINEAR-epilogue launches read **exactly 1.25x** after the LDS-transposed",
"  float4 epilogue went in.  Cause: `v *= a.scale` (float4 x kernel-argument scalar) made hipcc park a 16-byte slice of the kernel arguments",
"  in SCRATCH at kernel entry (`scratch_store_dwordx4` in the prologue, `scratch_load_dwordx4` in the epilogue; `.private_segment_fixed_size 32`):",
"  16 B per thread x 256 threads = 4 KB per 16 KB output tile = +25 % HBM writes on six launches per GRU iteration.  Written element-wise the",
"  scratch use is gone (column *now*: 1.00x everywhere; f2 carries the stream-K partial tiles it now uses), every kernel of libpfk reports",
"  `.private_segment_fixed_size 0`, and tests/test_no_scratch.py reads the kernel metadata out of libpfk.so on CPU and fails on any scratch or",
"  VGPR spill (it also caught 3 spilled VGPRs + a dynamically indexed vector in `corr_bf16_kernel`: 400 -> 320 us with both removed).",
"* **Fetch.**  The 1x1 convolutions read their (cold) input once: 1.04-1.07x — which also supports the x2 correction.  The multi-tap",
"  convolutions read 1.4-2.3x their one-pass input.  The tap re-reads of the implicit GEMM are mostly absorbed by LDS / L2 (else 5-9x), and it is",
"  not the halo (a 1x5 tile needs 68 pixels for 64 outputs and still reads 2.05x).  Tested and REJECTED: the tap re-use distance.  With the K",
"  order changed from (source, tap, chunk) to (source, chunk, tap) — every line's kh*kw uses in consecutive K-steps — FETCH moved by -5..-20 %",
"  only (c2 106 -> 91 MB, zr1 199 -> 161, zr2 174 -> 196, fm 59 -> 55; gpurun_out/y_pmc_fetch) while the kernels lost 12 % (fm 536 -> 611 us:",
"  the tap predicate becomes per-K-step vector work inside the hand-placed MFMA stream), so the change was reverted.  What remains is long-K",
"  launches whose column tiles / co-resident blocks drift apart over 36-72 K-steps and re-fetch lines the 4 MB L2 has dropped in between (the",
"  8-step 1x1 launches, which stay in lockstep, read 1.0x even with 9 column tiles).  It does not cost time: the multi-tap launches move",
"  0.33-0.90 TB/s (the 1x1 ones 1.2-1.3 TB/s), 4-16 % of HBM bandwidth, next to a matrix pipe that is ~80 % busy.",
"* **Issue.**  SQ_WAIT_ANY 11-12 % and MFMA-busy / SQ-busy = 26 on all big launches (busy cycles summed over the 4 SIMDs of 256 CUs against 32",
"  shader-engine SQ counters: 26 / 32 = 0.81 of the SIMD cycles have the matrix pipe busy); GRBM_GUI_ACTIVE / 8 XCDs / duration = 2.36 GHz, so no",
"  clock throttling hides in the fraction: fm at 123-126 TFLOP/s is 78-80 % of the 157.3 TFLOP/s fp32-MFMA peak, the rest is the per-tile",
"  prologue / epilogue and the per-K-step barrier.", ""]
open("profiles/r02_b_pmc_b8.md", "w").write("\n".join(out))
print("\n".join(rows))
PY
