"""scripts/callsite_stats.py — the per-call-site table of profiles/r05_a — on a synthetic kernel trace (CPU): the implicit-GEMM launches
of an iteration are told apart by their POSITION after the iteration's `lookup_kernel` dispatch; the mask branch's launches sit on a
side queue; with the fused kernel K13 an iteration has nine GEMM launches and no `mk`; encoder convolutions are never mis-filed."""
import csv
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    spec = importlib.util.spec_from_file_location("callsite_stats", os.path.join(ROOT, "scripts", "callsite_stats.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _trace(tmp_path, fused):
    rows, t, did = [], 0, 0

    def add(name, queue, dur):
        nonlocal t, did
        did += 1
        rows.append({"Kernel_Name": name, "Dispatch_Id": str(did), "Queue_Id": str(queue), "Start_Timestamp": str(t), "End_Timestamp": str(t + dur)})
        t += dur + 1

    conv = "void (anonymous namespace)::conv_gemm_v3_kernel<64, 64, 32, 32, 0, 1, 0, 32>(pfkg::GemmArgs)"
    for fwd in range(2):
        for _ in range(5):
            add(conv, 1, 1_000_000)                       # encoder convolutions: in front of the first lookup of the forward
        for it in range(3):
            add("void (anonymous namespace)::lookup_kernel<8, 4, float, false, true>(LookupArgs)", 1, 55_000)
            for k, d in enumerate((100, 400, 80, 270, 310, 165, 310, 165, 520)):
                add(conv, 1, d * 1000)
            add("flow_delta_kernel<8>(...)", 1, 35_000)
            if fused:
                add("(anonymous namespace)::mask_upsample_kernel(pfkg::GemmArgs)", 2, 196_000)
            else:
                add(conv, 2, 150_000)                     # mask conv2 on the side queue
                add("convex_upsample4_kernel(...)", 2, 62_000)
    d = tmp_path / ("fused" if fused else "pair")
    d.mkdir()
    with open(d / "r_kernel_trace.csv", "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    return str(d)


def _table(mod, path, capsys):
    import sys
    argv = sys.argv
    sys.argv = ["callsite_stats.py", path]
    try:
        mod.main()
    finally:
        sys.argv = argv
    out = capsys.readouterr().out
    tab = {}
    for line in out.splitlines():
        c = [x.strip() for x in line.strip().strip("|").split("|")]
        if len(c) >= 3 and c[1].isdigit():
            tab[c[0]] = (int(c[1]), float(c[2]))
    return tab


def test_positions_and_side_queue(tmp_path, capsys):
    mod = _load()
    tab = _table(mod, _trace(tmp_path, fused=False), capsys)
    assert tab["lookup"] == (6, 55.0) and tab["fm"] == (6, 520.0) and tab["c1"] == (6, 100.0) and tab["q2"] == (6, 165.0)
    assert tab["mk"] == (6, 150.0) and tab["upsample"] == (6, 62.0) and tab["flow_delta"] == (6, 35.0)
    assert "mask_upsample (fused)" not in tab


def test_fused_kernel_has_no_mk_row_and_encoders_are_not_misfiled(tmp_path, capsys):
    mod = _load()
    tab = _table(mod, _trace(tmp_path, fused=True), capsys)
    assert tab["mask_upsample (fused)"] == (6, 196.0) and "mk" not in tab and "upsample" not in tab
    assert tab["fm"] == (6, 520.0) and tab["zr1"] == (6, 310.0)        # the 1 ms encoder launches of the next forward stay out


def test_grouped_launch_and_the_b16_kernels(tmp_path, capsys):
    """Round 6: at small batches convc1 | convf2 | the previous iteration's mask conv2 are ONE grid (`conv_gemm_v3_group_kernel`), so
    the positional sequence behind it starts at c2 and ends with the last iteration's own mask conv2; the K8b kernels
    (`conv_gemm_b16_kernel`, `mask_upsample_b16_kernel`) follow the same rules as their fp32 forms."""
    mod = _load()
    rows, t, did = [], 0, 0

    def add(name, dur, queue=1):
        nonlocal t, did
        did += 1
        rows.append({"Kernel_Name": name, "Dispatch_Id": str(did), "Queue_Id": str(queue), "Start_Timestamp": str(t), "End_Timestamp": str(t + dur)})
        t += dur + 1

    conv = "void (anonymous namespace)::conv_gemm_v3_kernel<64, 64, 32, 32, 0, 1, 0, 36>(pfkg::GemmArgs)"
    group = "void (anonymous namespace)::conv_gemm_v3_group_kernel<32>((anonymous namespace)::GemmGroupArgs)"
    for it in range(3):
        add("lookup_kernel<4, 4, float, false, true>(LookupArgs)", 10_000)
        add("conv_cin2_tiled_kernel<7, float>(...)", 8_000)
        add(group, 40_000)
        if it:
            add("convex_upsample4_kernel<float>(...)", 8_000)
        for d in (63, 45, 48, 30, 48, 31, 77):
            add(conv, d * 1000)
        add("flow_delta_kernel<4, float>(...)", 7_000)
    add(conv, 25_000)                                       # the last iteration's mask conv2
    add("convex_upsample4_kernel<float>(...)", 8_000)
    d = tmp_path / "grouped"
    d.mkdir()
    with open(d / "r_kernel_trace.csv", "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    tab = _table(mod, str(d), capsys)
    assert tab["c1+f2[+mk] (grouped)"] == (3, 40.0)
    assert "c1" not in tab and "f2" not in tab
    assert tab["c2"] == (3, 63.0) and tab["fm"] == (3, 77.0) and tab["mk"] == (1, 25.0)
    assert tab["upsample"][0] == 3

    rows.clear()
    b16 = "void (anonymous namespace)::conv_gemm_b16_kernel<0, 256, 256, 2, 4, 2, 0>((anonymous namespace)::B16Args)"
    for it in range(2):
        add("lookup_kernel<8, 4, __bf16, false, true>(LookupArgs)", 52_000)
        for dd in (25, 65, 19, 44, 57, 41, 57, 41, 85):
            add(b16, dd * 1000)
        add("flow_delta_kernel<8, __bf16>(...)", 27_000)
        add("(anonymous namespace)::mask_upsample_b16_kernel((anonymous namespace)::MuB16Args)", 55_000)
    d2 = tmp_path / "b16"
    d2.mkdir()
    with open(d2 / "r_kernel_trace.csv", "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    tab = _table(mod, str(d2), capsys)
    assert tab["c1"] == (2, 25.0) and tab["fm"] == (2, 85.0) and "mk" not in tab
    assert tab["mask_upsample (fused)"] == (2, 55.0)
