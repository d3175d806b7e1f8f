"""CPU: the product library builds for gfx950, loads without a GPU and exports every symbol that
include/eeg_dcrnn.h declares (no compute calls here)."""
import ctypes
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
LIB = os.path.join(ROOT, "eeg_gnn_ssl_amd", "libeeg_dcrnn_hip.so")
DEV_LIB = os.path.join(ROOT, "eeg_gnn_ssl_amd", "libeeg_dcrnn_hip_dev.so")
PRODUCT_HEADERS = ("eeg_dcrnn.h", "eeg_dcrnn_prof.h")     # what libeeg_dcrnn_hip.so exports
DEV_HEADERS = ("eeg_dcrnn_dev.h",)                         # extra entry points of the dev build / test emulator only


def declared_symbols(headers=PRODUCT_HEADERS):
    out = set()
    for h in headers:
        text = re.sub(r"/\*.*?\*/", "", open(os.path.join(INC, h)).read(), flags=re.S)
        out |= set(re.findall(r"\b(eeg_dcrnn_\w+)\s*\(", text))
    return sorted(out)


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ("eeg_dcrnn_layer_fwd", "eeg_dcrnn_layer_bwd", "eeg_dcrnn_diffuse_fwd", "eeg_dcrnn_hop_polys",
                 "eeg_dcrnn_pack_cell", "eeg_dcrnn_cls_head_fwd", "eeg_dcrnn_last_error"):
        assert must in syms


def test_library_builds_loads_and_exports_every_declared_symbol():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "eeg_gnn_ssl_amd", "csrc"), "-j", "8", "all", "dev"],
                          stdout=subprocess.DEVNULL)
    dll = ctypes.CDLL(LIB)
    for s in declared_symbols():
        assert hasattr(dll, s), f"{s} declared in include/ but not exported by {LIB}"
    for s in declared_symbols(DEV_HEADERS):      # the product carries no tuning knobs / probes (no global mutable state)
        assert not hasattr(dll, s), f"development entry point {s} leaked into the product library"
        assert hasattr(ctypes.CDLL(DEV_LIB), s)
    dll.eeg_dcrnn_is_device_build.restype = ctypes.c_int
    assert dll.eeg_dcrnn_is_device_build() == 1
    assert dll.eeg_dcrnn_abi_version() == 5
    assert dll.eeg_dcrnn_supported(19, 64, 100, 3) == 1
    assert dll.eeg_dcrnn_supported(19, 48, 100, 3) == 0
    dll.eeg_dcrnn_last_error.restype = ctypes.c_char_p
    assert b"rnn_units" in dll.eeg_dcrnn_last_error()


def test_python_binding_matches_header():
    from eeg_gnn_ssl_amd import _lib
    assert sorted(_lib._SIGNATURES) == declared_symbols()
    assert sorted(_lib._SIGNATURES_DEV) == declared_symbols(DEV_HEADERS)


def test_operators_are_registered_with_the_dispatcher():
    """north_star: 'exposed as a torch.ops extension' — every operator of the path has a schema in the
    `eeg_dcrnn` namespace (implementations: eeg_gnn_ssl_amd/ops.py over the C ABI)."""
    import torch
    import eeg_gnn_ssl_amd  # noqa: F401  (registers the library)
    for name in ("hop_polys", "pack_cell", "diffusion_hops", "dconv", "dconv_bwd", "dcgru_layer", "dcgru_layer_bwd",
                 "dcgru_decoder", "dcgru_decoder_bwd", "spectral_basis", "pack_cell_spectral", "cls_head", "cls_head_bwd", "rng_take_", "dropout_mask", "gather_last", "corr_graph", "fft_features",
                 "bce_logits", "ce_logits", "masked_loss", "clip_adam_", "clip_adam_dev_", "teacher_flags_", "augment_draw_"):
        op = getattr(torch.ops.eeg_dcrnn, name)
        assert op.default._schema.name == f"eeg_dcrnn::{name}"
    assert torch.ops.eeg_dcrnn.clip_adam_.default._schema.is_mutable


def test_a_library_of_another_abi_version_is_refused_in_both_modes(tmp_path, monkeypatch):
    """strict=False (development A/B loads of older builds) tolerates MISSING entry points only: another ABI version means
    other signatures behind the same names, i.e. shifted arguments"""
    import pytest
    from eeg_gnn_ssl_amd import _lib
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1)
    for strict in (True, False):
        with pytest.raises(ImportError, match="ABI version"):
            _lib.EegDcrnnLib(LIB, strict=strict)


def test_dropout_generator_state_is_created_with_the_module_and_leaves_the_global_generator_alone():
    import torch
    from eeg_gnn_ssl_amd import DCRNNModel_classification, ops
    import bench
    torch.manual_seed(99)
    before = torch.get_rng_state()
    a, b = ops.make_rng_state("cpu", 0), ops.make_rng_state("cpu", 1)
    assert torch.equal(before, torch.get_rng_state())                       # a dedicated generator: no global draw
    assert a[0] != b[0] and a[1] == 0
    torch.manual_seed(99)
    assert torch.equal(ops.make_rng_state("cpu", 0), a)                     # torch.manual_seed governs the seed
    args = bench.make_args("laplacian", dropout=0.5)
    m = DCRNNModel_classification(args, 4)
    assert "_dropout_rng" in dict(m.named_buffers()) and "_dropout_rng" not in m.state_dict()
    m.set_dropout_seed(5, 9)
    assert m.dropout_rng_state() == (5, 9)


def test_operators_switch_to_the_device_of_their_tensors(monkeypatch):
    """Python-registered operators get no DeviceGuard from the dispatcher; ops._device_guarded supplies it: same device -> the
    implementation runs as it is (one comparison), another device -> inside torch.cuda.device(that device)."""
    import contextlib
    import torch
    from eeg_gnn_ssl_amd import ops
    entered = []

    @contextlib.contextmanager
    def fake_device(d):
        entered.append(d)
        yield

    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    monkeypatch.setattr(torch.cuda, "device", fake_device)
    calls = []
    run = ops._device_guarded(lambda *a, **k: calls.append((a, k)) or "out")
    monkeypatch.setattr(ops, "_tensor_device", lambda args: torch.device("cuda", 1) if args else None)
    assert run(1, 2, x=3) == "out" and entered == [torch.device("cuda", 1)] and calls == [((1, 2), {"x": 3})]
    monkeypatch.setattr(ops, "_tensor_device", lambda args: torch.device("cuda", 0) if args else None)
    assert run(5) == "out" and len(entered) == 1          # same device: no switch
    monkeypatch.undo()
    # the device is found in nested tensor lists too; CPU tensors name none
    assert ops._tensor_device((1, [torch.zeros(1)], "s")) is None


def test_size_queries_never_fault_on_degenerate_dims():
    """Every host-only size / capability query of the C ABI with zero batch, zero steps, zero widths: must return (0 = "nothing
    to allocate" / "not covered"), never fault -- round 5 found `eeg_dcrnn_corr_graph_ws_floats(0, T)` and five more dividing by
    the batch size (SIGFPE in the host process).  Runs in a subprocess: a fault would otherwise take the test session down."""
    import sys
    import textwrap
    code = textwrap.dedent("""
        import ctypes, itertools, sys
        sys.path.insert(0, %r)
        from eeg_gnn_ssl_amd import _lib
        lib = _lib.get_lib()
        n = 0
        for b, t in itertools.product((0, 1, 256), (0, 1, 60)):
            v = lib.query("eeg_dcrnn_corr_graph_ws_floats", b, t); n += 1
            assert (v == 0) == (b == 0 or t == 0), (b, t, v)
        layer = [(0, 0, 0, 0, 0, 0), (60, 0, 19, 64, 100, 3), (0, 256, 19, 64, 100, 3), (60, 256, 0, 64, 100, 3), (60, 256, 19, 0, 100, 3),
                 (60, 256, 19, 64, 0, 3), (60, 256, 19, 64, 100, 0)]
        for dims in layer:
            d = _lib.LayerDims()
            for f, v in zip(("T", "B", "N", "H", "Fin", "M"), dims): setattr(d, f, v)
            assert lib.query("eeg_dcrnn_layer_fwd_ws_floats", ctypes.byref(d)) == 0
            assert lib.query("eeg_dcrnn_layer_bwd_ws_floats", ctypes.byref(d), 1) == 0
            lib.query("eeg_dcrnn_batch_major_ok", ctypes.byref(d)); n += 3
        dec = [(0, 0, 0, 0, 0, 0, 0), (12, 0, 19, 64, 100, 5, 2), (0, 512, 19, 64, 100, 5, 2), (12, 512, 0, 64, 100, 5, 2), (12, 512, 19, 0, 100, 5, 2),
               (12, 512, 19, 64, 0, 5, 2), (12, 512, 19, 64, 100, 0, 2), (12, 512, 19, 64, 100, 5, 0)]
        for dims in dec:
            d = _lib.DecoderDims()
            for f, v in zip(("T", "B", "N", "H", "Dout", "M", "L"), dims): setattr(d, f, v)
            for fn in ("eeg_dcrnn_decoder_saved_floats", "eeg_dcrnn_decoder_fwd_ws_floats", "eeg_dcrnn_decoder_bwd_ws_floats", "eeg_dcrnn_decoder_is_persistent"):
                assert lib.query(fn, ctypes.byref(d)) == 0, (fn, dims); n += 1
        for dims in [(0, 19, 100, 3, 64), (4, 0, 100, 3, 64), (4, 19, 0, 3, 64), (4, 19, 100, 0, 64), (4, 19, 100, 3, 0)]:
            assert lib.query("eeg_dcrnn_dconv_fwd_ws_floats", *dims) == 0 and lib.query("eeg_dcrnn_dconv_bwd_ws_floats", *dims) == 0; n += 2
        for dims in [(0, 0, 0), (100, 64, 0), (0, 64, 3), (100, 0, 3)]:
            lib.query("eeg_dcrnn_pack_floats", *dims); lib.query("eeg_dcrnn_pack3_halves", *dims); n += 2
        for dims in [(0, 0, 0, 0), (19, 64, 100, 0), (19, 0, 100, 3), (0, 64, 100, 3), (19, 64, 0, 3)]:
            assert lib.query("eeg_dcrnn_supported", *dims) == 0; n += 1
        # and the sizes of a real shape are not zero
        d = _lib.LayerDims()
        for f, v in zip(("T", "B", "N", "H", "Fin", "M"), (60, 256, 19, 64, 100, 3)): setattr(d, f, v)
        assert lib.query("eeg_dcrnn_layer_bwd_ws_floats", ctypes.byref(d), 1) > 0
        print("queries:", n)
    """ % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, f"rc={r.returncode} (negative = signal)\n{r.stdout[-500:]}\n{r.stderr[-1500:]}"
    assert "queries:" in r.stdout


def test_product_has_no_cpu_path():
    """Without a GPU the product ops must refuse CPU tensors loudly (no silent fallback)."""
    import pytest
    import torch
    from eeg_gnn_ssl_amd import DCGRUCell, _lib
    _lib._LIB = None
    cell = DCGRUCell(100, 64, 2, 19)
    # the operators exist for the CUDA (= HIP) key only: the dispatcher refuses ("... from the 'CPU' backend"); when another
    # test of this process has installed the emulator's CPU-key registrations, the C-ABI stub of the product library refuses
    with pytest.raises(RuntimeError, match="no CPU path|'CPU' backend"):
        cell([torch.eye(19)], torch.zeros(2, 1900), torch.zeros(2, 19 * 64))


def test_split_bf16_report_parser():
    """bench.py --split-bf16-experiment: the lab binary's report (a committed run: profiles/r03_bf16x3_lab.txt) becomes the
    `experimental_split_bf16` object; the 6-product split is no less accurate than the fp32 matrix pipe."""
    import bench
    e = bench.parse_split_bf16_report(open(os.path.join(ROOT, "profiles", "archive", "r03_bf16x3_lab.txt")).read())
    assert e["six_products"]["ms"] > 0 and e["three_products"]["ms"] > 0 and e["fp32_mfma_ms"] > 0
    assert e["six_products"]["ms"] > e["three_products"]["ms"]
    assert e["six_products"]["max_abs_err_vs_fp64"] <= 1.5 * e["fp32_mfma_max_abs_err_vs_fp64"] < 2e-5
    assert e["three_products"]["max_abs_err_vs_fp64"] < 2e-5
    assert "NOT in the product path" in e["scope"]


def test_product_sources_do_not_know_the_emulator():
    """The SIMT emulator is test scaffolding: the product tree reaches its platform layer through one #include of
    EEG_PLATFORM_HEADER (csrc/platform.h by default) and carries no emulator switch, oracle import or CPU stand-in."""
    import os
    import re
    pkg = os.path.join(ROOT, "eeg_gnn_ssl_amd")
    hits = []
    for base, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith((".h", ".cpp", ".hip", ".py")) and f != "Makefile":
                continue
            text = open(os.path.join(base, f), errors="replace").read()
            for pat in (r"EEG_SIMT_EMU", r"simt_emu", r"platform_emu", r"^\s*(from|import)\s+oracle"):
                if re.search(pat, text, flags=re.M):
                    hits.append((os.path.relpath(os.path.join(base, f), ROOT), pat))
    assert not hits, hits
    common = open(os.path.join(pkg, "csrc", "common.h")).read()
    assert '#define EEG_PLATFORM_HEADER "platform.h"' in common and "#include EEG_PLATFORM_HEADER" in common


def test_strict_lengths_raise_like_the_reference():
    """`utils.last_relevant_pytorch` (utils.py:346-357) gathers at len-1 on host-side lengths: out-of-range lengths raise.  The
    opt-in host check reproduces that (the default path clamps without a host synchronisation)."""
    import pytest
    import torch
    from eeg_gnn_ssl_amd import utils
    utils.check_seq_lengths(torch.tensor([1, 12, 7]), 12)
    for bad in ([0, 3], [13, 1], [-2, 5]):
        with pytest.raises(RuntimeError, match="out of bounds"):
            utils.check_seq_lengths(torch.tensor(bad), 12)


def test_cosine_schedule_matches_torch():
    """utils.cosine_annealing_lr == CosineAnnealingLR(T_max=num_epochs) stepped per epoch (train.py:224,329)."""
    import torch
    from eeg_gnn_ssl_amd import utils
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=3e-4)
    sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=60)
    for e in range(1, 61):
        opt.step()
        sch.step()
        assert abs(sch.get_last_lr()[0] - utils.cosine_annealing_lr(3e-4, e, 60)) < 1e-12


def test_eval_and_checkpoint_helpers(tmp_path):
    """utils.thresh_max_f1 / eval_dict / CheckpointSaver (reference utils.py:84-150,285-343): the threshold
    maximises F1 over all candidate thresholds; best/last checkpoint protocol."""
    import numpy as np
    import torch
    from sklearn.metrics import f1_score
    from eeg_gnn_ssl_amd import utils
    rng = np.random.RandomState(0)
    y = (rng.rand(300) > 0.7).astype(int)
    prob = np.clip(0.35 * y + 0.65 * rng.rand(300), 0, 1)
    th = utils.thresh_max_f1(y, prob)
    best = max(f1_score(y, (prob >= t).astype(int)) for t in np.unique(prob))
    assert abs(f1_score(y, (prob >= th).astype(int)) - best) < 1e-12
    scores, pred, true = utils.eval_dict((prob > th).astype(int), y, y_prob=prob, file_names=[f"f{i}" for i in range(300)])
    assert set(scores) == {"acc", "F1", "precision", "recall", "auroc"} and len(pred) == 300 and true["f3"] == y[3]

    class Opt:
        def state_dict(self):
            return {"k": 1}
    model = torch.nn.Linear(3, 2)
    saver = utils.CheckpointSaver(str(tmp_path), "auroc", maximize_metric=True)
    saver.save(1, model, Opt(), 0.6)
    with torch.no_grad():
        model.weight.add_(1.0)
    saver.save(2, model, Opt(), 0.5)                       # worse: only last.pth.tar moves
    best_ck = torch.load(tmp_path / "best.pth.tar", weights_only=False)
    last_ck = torch.load(tmp_path / "last.pth.tar", weights_only=False)
    assert best_ck["epoch"] == 1 and last_ck["epoch"] == 2 and saver.best_val == 0.6
    fresh = torch.nn.Linear(3, 2)
    utils.load_model_checkpoint(str(tmp_path / "last.pth.tar"), fresh)
    assert torch.equal(fresh.weight, model.weight)
