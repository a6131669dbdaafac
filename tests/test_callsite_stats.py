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
