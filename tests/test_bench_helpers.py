"""CPU checks of bench.py's host-side bookkeeping (no GPU): the per-launch work table the by-symbol roofline is priced with, the parser of
the library's per-kernel report, the symbol spelling shared with tools/pmc_traffic.sh, and the synthetic raw signals."""
import json
import os

import numpy as np

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_per_launch_work_adds_up_to_the_role_totals():
    for filt, t_len, batch, task, layers in (("laplacian", 60, 256, "detection", 2), ("dual_random_walk", 60, 256, "detection", 2),
                                            ("dual_random_walk", 60, 512, "ssl", 2), ("laplacian", 12, 4, "detection", 3)):
        w = bench.algorithmic_work(filt, t_len, batch, task, layers)
        lw = bench.per_launch_work(filt, t_len, batch, layers)
        for role, per_launch in lw.items():
            assert abs(sum(per_launch) - w[role]) <= 1e-9 * max(w[role], 1.0), (filt, role)
        # backward roles are listed top layer first (the order their launches are recorded in)
        assert lw["gemm_tn_x"][-1] == 2.0 * t_len * batch * bench.N_NODES * ((2 if filt == "dual_random_walk" else 1) * 2 + 1) * bench.D_IN * 3 * bench.H_UNITS
    raw = bench.algorithmic_work("dual_random_walk", 60, 256, "detection", 2, raw=True)
    assert raw["fft_features"] == 4.0 * 60 * 256 * 19 * (200 + 2 * 100)             # windows read once, two feature tensors written


def test_prof_report_parser_and_symbol_spelling():
    txt = ("seq_fwd 40 10.0 seq_fwd2_kernel<64, 3, 5, false>\n"
           "gemm_tn_x 20 3.5 gemm_tnq_kernel<6, 6, 16, false, true, false>\n"
           "gemm_tn_x 20 5.9 gemm_tnq_kernel<5, 6, 16, true, false, false>\n"
           "zero 40 0.2 zero_kernel\n")
    p = bench.parse_prof_report(txt)
    assert p["seq_fwd"] == {"count": 40, "ms": 10.0, "symbols": {"seq_fwd2_kernel<64,3,5>": (40, 10.0)}}
    assert list(p["gemm_tn_x"]["symbols"]) == ["gemm_tnq_kernel<6,6,16,false,true>", "gemm_tnq_kernel<5,6,16,true>"]      # first-launch order
    assert p["gemm_tn_x"]["count"] == 40 and abs(p["gemm_tn_x"]["ms"] - 9.4) < 1e-12
    assert bench.short_symbol("diffuse_adj_rows_kernel<19, 5>") == "diffuse_adj_rows_kernel<19,5>"
    assert bench.short_symbol("gemm_nnr_kernel<4, 2>") == "gemm_nnr_kernel<4,2>"


def test_committed_pmc_traffic_files_carry_per_symbol_traffic_of_the_current_sources():
    """profiles/pmc_traffic_<workload>.json: stamped with the hash of the kernel sources they were collected on; bench.py uses them only
    when the stamp matches, and the by-symbol table must name the kernels the bench line ranks"""
    sha = bench.kernel_sources_sha256()
    for w, must in (("cfg2", ("gemm_nnf_kernel<13>", "gemm_nnf_kernel<8,true>", "seq_bwd2_kernel<64,3,5,false,true>",
                               "seq_fwd2_kernel<64,3,5,false,true>", "gemm_tnf_kernel<2>", "gemm_tnf_kernel<4>")),
                    ("cfg3", ("seq_fwd_kernel<64,5,5>", "corr_gram_kernel<7,true>")), ("cfg5", ("dec_fwd_persist_kernel<64,5>",)),
                    ("raw", ("fft200_features_kernel",))):
        d = json.load(open(os.path.join(ROOT, "profiles", f"pmc_traffic_{w}.json")))
        by_sym = d["traffic_bytes_per_launch_by_symbol"]
        for k in must:
            assert k in by_sym and by_sym[k] > 0, (w, k)
        if d["kernel_sources_sha256"] != sha:
            import warnings
            warnings.warn(f"profiles/pmc_traffic_{w}.json was collected on other kernel sources: bench.py will report traffic: null")


def test_synthetic_raw_signals_have_structure():
    x = bench.synthetic_raw_signals(3, 4, seed=5)
    assert tuple(x.shape) == (3, bench.N_NODES, 4 * bench.RAW_WINDOW) and x.dtype.is_floating_point
    assert np.array_equal(x.numpy(), bench.synthetic_raw_signals(3, 4, seed=5).numpy())          # seeded
    # channels of a clip are correlated through the shared sources (white noise alone would give |r| ~ 0.04 at this length)
    r = np.corrcoef(x[0].numpy())
    assert np.abs(r[np.triu_indices(bench.N_NODES, 1)]).max() > 0.5


def test_paired_h_part_role_is_priced_with_the_sum_of_its_two_problems():
    """the library reports ONE role (`gemm_tn_h`) where it launches the two h-part weight-gradient GEMMs of a cell as a pair:
    the work tables follow the roles of the report, totals unchanged, nothing counted twice; a report with the two separate
    roles (dev knob 19, shapes outside the whole-block kernel) leaves the tables alone"""
    work = bench.algorithmic_work("dual_random_walk", 60, 512, task="ssl")
    lw = bench.per_launch_work("dual_random_walk", 60, 512)
    total = sum(work.values())
    hg, hc, dhg, dhc = work["gemm_tn_hg"], work["gemm_tn_hc"], work["dec_gemm_tn_hg"], work["dec_gemm_tn_hc"]
    w2, l2 = bench.merge_paired_roles(dict(work), {k: list(v) for k, v in lw.items()}, {"gemm_tn_h": {}, "dec_gemm_tn_h": {}, "gemm_tn_x": {}})
    assert "gemm_tn_hg" not in w2 and "gemm_tn_hc" not in w2 and w2["gemm_tn_h"] == hg + hc and w2["dec_gemm_tn_h"] == dhg + dhc
    assert abs(sum(w2.values()) - total) < 1e-3 * total
    assert l2["gemm_tn_h"] == [a + b for a, b in zip(lw["gemm_tn_hg"], lw["gemm_tn_hc"])] and "gemm_tn_hg" not in l2
    w3, l3 = bench.merge_paired_roles(dict(work), {k: list(v) for k, v in lw.items()}, {"gemm_tn_hg": {}, "gemm_tn_hc": {}})
    assert w3 == work and l3 == lw
