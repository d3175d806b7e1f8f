"""CPU leg: the kernel SOURCES compiled against the SIMT emulator (tests/emu), driven through the
same C ABI + Python host layer as the product, checked against the reference's golden vectors.
This validates kernel logic / index maps / orchestration without a GPU; the `-m gpu` tests in
test_gpu_parity.py run the identical checks on the real MI355X library."""
import pytest

import cases
import parity_suite as ps


@pytest.fixture(scope="module", autouse=True)
def emulator():
    import emu_support
    lib = emu_support.install_emulator()
    yield lib
    emu_support.uninstall()


@pytest.mark.parametrize("tag", list(cases.CELL_CASES) + list(cases.CELL_K_CASES))
def test_cell(tag, golden, adj3d):
    ps.check_cell_case(tag, golden, adj3d, "cpu")


@pytest.mark.parametrize("tag", list(cases.DCONV_CASES))
def test_dconv(tag, golden, adj3d):
    ps.check_dconv_case(tag, golden, adj3d, "cpu")


@pytest.mark.parametrize("tag", list(cases.CLS_CASES))
def test_classification_model(tag, golden, adj3d):
    ps.check_cls_case(tag, golden, adj3d, "cpu")


@pytest.mark.parametrize("tag", list(cases.SSL_CASES))
def test_ssl_model(tag, golden, adj3d):
    ps.check_ssl_case(tag, golden, adj3d, "cpu")


def test_dropout_generator():
    ps.check_dropout_generator("cpu")


@pytest.mark.parametrize("tag", list(cases.DROPOUT_CLS_TAGS))
def test_classification_model_training_dropout(tag, golden_dropout, adj3d):
    ps.check_dropout_cls_case(tag, golden_dropout, adj3d, "cpu")


@pytest.mark.parametrize("tag", list(cases.DROPOUT_SSL_TAGS))      # dual_default (64 units) runs the persistent decoder kernels
def test_ssl_model_training_dropout(tag, adj3d):
    ps.check_dropout_ssl_case(tag, adj3d, "cpu")


@pytest.mark.parametrize("filt,din,layers", [("laplacian", 100, 2), ("dual_random_walk", 36, 2)])
def test_opt_in_split_bf16_gemms(filt, din, layers, adj3d):
    ps.check_split_bf16("cpu", adj3d, filt=filt, din=din, layers=layers, t_len=2, b=2)


SPECTRAL_CASES = [dict(din=100, layers=2, t_len=3, b=4, classes=1),
                  dict(din=8, layers=2, t_len=5, b=7, classes=4, k=3, seed=3, lengths=[5, 4, 3, 2, 1, 5, 5]),
                  dict(din=36, layers=3, t_len=2, b=3, classes=1, k=1, seed=5, n=20, act="relu"),
                  dict(din=12, layers=2, t_len=2, b=40, classes=1, seed=6, n=7),
                  # fused weight-gradient GEMM (kernels_gemm_f.h): three Xh tiles (the odd one rides on a junk tile), the widest input,
                  # and a ring that wraps (80 rows per frequency = 10 chunks through 5 stages)
                  dict(din=68, layers=2, t_len=3, b=3, classes=1, seed=8, n=5),
                  dict(din=128, layers=1, t_len=4, b=20, classes=1, seed=9, n=3)]


@pytest.mark.parametrize("case", SPECTRAL_CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items() if k in ("din", "layers", "n", "b", "k")))
def test_spectral_form_of_the_hoisted_x_part(case, adj3d):
    ps.check_spectral_form("cpu", adj3d, **case)


def test_spectral_basis_and_shared_support_detection(adj3d):
    ps.check_spectral_basis("cpu", adj3d)


def test_empty_inputs_raise_like_the_reference():
    ps.check_empty_inputs("cpu")


def test_operands_that_do_not_fit_each_other_are_refused_before_launch():
    ps.check_malformed_inputs("cpu")


def test_teacher_flags_known_answer():
    ps.check_teacher_flags("cpu")


def test_ssl_model_device_curriculum(adj3d):
    # (with dropout: the flags AND the masks come from the one device generator; the GPU suite also runs p = 0 and six draws)
    ps.check_ssl_device_curriculum("dual_default", adj3d, "cpu", 0.5, reps=4)


def test_device_step_adam():
    ps.check_device_step_adam("cpu")


def test_device_flags_refused_outside_the_persistent_kernels(adj3d):
    with pytest.raises(Exception, match="persistent decoder kernel"):
        ps.check_decoder_vs_oracle("cpu", "laplacian", 8, 16, 2, 4, 2, adj3d, seed=1, ratio="device")


def test_random_vs_oracle_h32_relu_varlen(adj3d):
    ps.check_vs_oracle_random("cpu", "dual_random_walk", 12, 32, 2, 4, 3, 4, adj3d, seed=3, lengths=[4, 2, 1], act="relu")


def test_widest_cell_uses_the_20_row_backward_layout(adj3d):
    """rnn_units=64 with 7 hop matrices (dual random walk, K=3): the BPTT kernel's LDS tiles only fit with 20
    node rows (SeqGeom::bwd_rows); logits and all gradients vs the oracle."""
    ps.check_vs_oracle_random("cpu", "dual_random_walk", 8, 64, 2, 3, 2, 1, adj3d, seed=5, k=3)


@pytest.mark.parametrize("n,h,filt,k", [(3, 16, "laplacian", 1), (21, 16, "dual_random_walk", 2), (32, 32, "random_walk", 3)])
def test_shape_sweep_vs_oracle(n, h, filt, k):
    ps.check_shape_sweep("cpu", n, h, filt, k)


def test_wide_rows_diffuse_in_column_chunks():
    """24 nodes x 200 input features x 5 hop matrices exceed one LDS tile: the diffusion kernels walk the rows
    in column chunks"""
    ps.check_shape_sweep("cpu", 24, 16, "dual_random_walk", 2, din=200, t_len=2, b=2, layers=2)


def test_single_wave_forward_kernel_at_default_width(emulator, adj3d, golden):
    """64 units with M <= 3 run the two-wave-per-SIMD forward kernel by default; the single-wave kernel (still used
    by the cycle probe and for every other width / hop count) must give the same results at that shape."""
    emulator.call("eeg_dcrnn_set_tuning", 12, 1)
    try:
        ps.check_cell_case("lap_default", golden, adj3d, "cpu")
        ps.check_cell_case("lap_l1_default", golden, adj3d, "cpu")
    finally:
        emulator.call("eeg_dcrnn_set_tuning", 12, 0)


@pytest.mark.parametrize("filt,k,lengths,act", [("dual_random_walk", 2, [3, 1, 2], "tanh"), ("laplacian", 2, None, "relu"),
                                                ("dual_random_walk", 1, None, "tanh"), ("laplacian", 0, [2, 3, 3], "tanh")])
def test_streamed_weight_bptt_kernel(emulator, adj3d, filt, k, lengths, act):
    """kernels_seq_stream.h (BPTT with two workgroups per CU, weights streamed from L2; by default only for batches
    beyond 1.5 clips per CU at M >= 4) forced on: whole model vs the oracle."""
    emulator.call("eeg_dcrnn_set_tuning", 3, 1)
    try:
        ps.check_vs_oracle_random("cpu", filt, 8, 64, 2, 3, 3, 4, adj3d, seed=5, lengths=lengths, act=act, k=k)
    finally:
        emulator.call("eeg_dcrnn_set_tuning", 3, 0)


def test_spectral_weight_gradients_fused_and_as_three_grouped_launches(emulator, adj3d):
    """Round 6: gemm_tnf_kernel computes the x-part and both h-part weight gradients of a cell in one pass over dYh; dev knob 23 = 1
    keeps the three grouped launches it replaces (still the path of input widths beyond 128).  Both against the oracle."""
    for separate in (0, 1):
        emulator.call("eeg_dcrnn_set_tuning", 23, separate)
        emulator.call("eeg_dcrnn_set_tuning", 20, separate)      # and gemm_nnf_kernel (weights in registers) / the round-5 grouped NN kernel
        emulator.call("eeg_dcrnn_set_tuning", 17, separate)      # and gemm_dxf_kernel (input gradient in one kernel) / grouped GEMM + node mix
        try:
            ps.check_spectral_form("cpu", adj3d, din=100, layers=2, t_len=3, b=6, classes=1, seed=11)
        finally:
            emulator.call("eeg_dcrnn_set_tuning", 23, 0)
            emulator.call("eeg_dcrnn_set_tuning", 20, 0)
            emulator.call("eeg_dcrnn_set_tuning", 17, 0)
    # gemm_nnf_kernel: a chunk range that crosses into the next frequency (weights reloaded), a ragged last chunk (80 rows = 64 + 16)
    ps.check_spectral_form("cpu", adj3d, din=100, layers=2, t_len=2, b=40, classes=1, seed=13, n=6)
    # beyond 128 input features the fused kernel has no instantiation: the grouped launches take the layer
    ps.check_spectral_form("cpu", adj3d, din=132, layers=1, t_len=2, b=3, classes=1, seed=12, n=4)


@pytest.mark.parametrize("filt,k", [("laplacian", 2), ("dual_random_walk", 2)])
def test_round3_hoisted_gemms_and_the_paired_h_part_launch(emulator, adj3d, filt, k):
    """The whole-block GEMMs of kernels_gemm_q.h take over from 256 rows per CU on (the GPU suite's full-size cases); dev knob 2
    lowers that to one row per CU so that the emulator runs them too: gemm_nnr_kernel, gemm_tnq_kernel (planar at M = 3, per-lane
    pointers at M = 5) and -- round 5 -- gemm_tnq_pair_kernel, the two h-part weight-gradient GEMMs of a cell in one launch
    (knob 19 = 1: one by one; both must match the oracle, and each other bit for bit: same partial sums, same reduction)."""
    grads = {}
    for one_by_one in (0, 1):
        emulator.call("eeg_dcrnn_set_tuning", 2, 4)
        emulator.call("eeg_dcrnn_set_tuning", 19, one_by_one)
        try:
            grads[one_by_one] = ps.check_vs_oracle_random("cpu", filt, 8, 64, 2, 4, 4, 4, adj3d, seed=7, k=k)
        finally:
            emulator.call("eeg_dcrnn_set_tuning", 2, 0)
            emulator.call("eeg_dcrnn_set_tuning", 19, 0)
    if grads[0] is not None:
        for name in grads[0]:
            assert (grads[0][name] == grads[1][name]).all(), name


def test_randomized_shapes_through_the_whole_block_gemms(emulator, adj3d):
    """Dev knob 2 hands every hoisted GEMM to the round-3 kernels (kernels_gemm_q.h: gemm_nnr, gemm_tnq, the paired h-part launch) where
    they cover the shape -- on the GPU they only start at 256 rows per CU, i.e. at the few full-size shapes of the GPU suite.  A
    seeded draw over filter types, hop counts, widths (planar 64-wide planes and per-lane pointers), layer counts, row counts that are
    and are not multiples of 16 (the TN kernel's condition: the others fall back), ragged lengths: logits and every gradient vs the
    oracle.  (600 s of the same generator: 427 cases, 0 mismatches.)"""
    import random
    import fuzz_gpu
    rng = random.Random(3)
    keep = ps.assert_close_scaled
    ps.assert_close_scaled = fuzz_gpu._close_scaled            # (floor for <= 8-element tensors: the one-class bias gradient cancels)
    emulator.call("eeg_dcrnn_set_tuning", 2, 4)
    try:
        done = 0
        while done < 14:
            filt = rng.choice(["laplacian", "random_walk", "dual_random_walk"])
            k, h = rng.choice([1, 2, 2, 3]), rng.choice([64, 64, 32, 16])
            din, layers = rng.choice([4, 8, 20, 36, 64, 100]), rng.choice([1, 2, 3])
            t_len, b = rng.choice([(4, 4), (2, 8), (8, 2), (1, 16), (16, 1), (4, 8), (3, 5), (2, 3)])
            lengths = None if rng.random() < 0.5 else [rng.randint(1, t_len) for _ in range(b)]
            try:
                ps.check_vs_oracle_random("cpu", filt, din, h, layers, t_len, b, rng.choice([1, 4]), adj3d, seed=rng.randrange(1 << 16),
                                          lengths=lengths, act="tanh", k=k)
                done += 1
            except RuntimeError as e:
                if "unsupported" not in str(e) and "needs" not in str(e):
                    raise
    finally:
        emulator.call("eeg_dcrnn_set_tuning", 2, 0)
        ps.assert_close_scaled = keep


def test_training_tail_kernels():
    ps.check_training_tail("cpu")


@pytest.mark.parametrize("filt,dout,h,layers,t_out,b,ratio,act", [
    ("laplacian", 8, 16, 2, 5, 3, 0.5, "tanh"),            # teacher forcing on some steps
    ("dual_random_walk", 12, 32, 3, 4, 2, 0.6, "relu"),    # shared cell used by two layers + teacher forcing
    ("laplacian", 20, 16, 1, 3, 2, None, "tanh"),          # single layer, fully autoregressive
    ("laplacian", 8, 64, 2, 2, 2, None, "tanh"),           # 64 units, M = 3: single-step launches of the two-wave kernel
    ("dual_random_walk", 20, 64, 2, 3, 2, 0.5, "tanh"),    # 64 units, Dout % 20 == 0: the persistent decoder kernel (M = 5), teacher forcing
    ("laplacian", 16, 64, 3, 3, 2, None, "relu"),          # persistent kernel, 3 layers (shared cell), Dout % 16 == 0
    ("laplacian", 100, 64, 2, 3, 2, 0.5, "tanh"),          # persistent kernels at M = 3, Dout = 100 (two input-gradient tiles per wave)
])
def test_decoder_vs_oracle(filt, dout, h, layers, t_out, b, ratio, act, adj3d):
    ps.check_decoder_vs_oracle("cpu", filt, dout, h, layers, t_out, b, adj3d, seed=1, ratio=ratio, act=act)


@pytest.mark.parametrize("filt,dout,layers,t_out,b,ratio,n,order", [
    ("laplacian", 16, 1, 1, 2, None, 12, 1),            # one layer, one step, 12 nodes (no remainder tile), M = 2
    ("dual_random_walk", 20, 2, 2, 2, None, 20, 1),     # 20 nodes: the 4x4 remainder tile full, M = 3
    ("laplacian", 16, 2, 3, 2, 0.5, 19, 0),             # max_diffusion_step = 0: M = 1, no hop slots
    ("dual_random_walk", 20, 4, 2, 1, None, 5, 1),      # four layers (three uses of the shared cell), 5 nodes
    ("dual_random_walk", 60, 2, 2, 2, None, 19, 2),     # Dout = 60, M = 5: three leftover 16-byte pieces per hop slot -> 4 tail chunks of the quad pack
    ("laplacian", 20, 3, 6, 2, "device", 19, 1),        # teacher-forcing flags read from DEVICE memory by the persistent kernels
])
def test_persistent_decoder_edge_shapes(filt, dout, layers, t_out, b, ratio, n, order, adj3d):
    """the persistent decoder kernels (forward and BPTT, kernels_decoder.h) at the edges of their range"""
    ps.check_decoder_vs_oracle("cpu", filt, dout, 64, layers, t_out, b, adj3d, seed=3, ratio=ratio, n=n, order=order)


def test_correlation_graph_supports(golden):
    ps.check_correlation_supports("cpu", golden)


def test_grad_sink_equals_autograd_accumulation(adj3d):
    ps.check_grad_sink("cpu", adj3d)


def test_hop_plane_handover_between_layers(adj3d):
    ps.check_plane_handover("cpu", adj3d)


def test_ssl_evaluation_driver(adj3d):
    ps.check_ssl_eval_driver("cpu", adj3d)


def test_evaluation_driver(adj3d):
    ps.check_eval_driver("cpu", adj3d)


def test_raw_signals_to_step_chain_vs_oracle():
    ps.check_raw_input_chain("cpu", b=2, t_len=2)


def test_fused_head_and_criterion_operator():
    ps.check_cls_head_loss("cpu", shapes=((1, 19, 64, 1), (5, 19, 64, 4), (37, 21, 32, 3)))


@pytest.mark.parametrize("task", ["detection", "classification"])
def test_fused_head_step_equals_public_path(task, adj3d):
    ps.check_fused_head_step_equals_public_path("cpu", adj3d, task)


def test_augmentation_draws_known_answer(adj3d):
    ps.check_augmentation_draws("cpu", adj3d)


@pytest.mark.parametrize("graph,raw", [("distance", True), ("correlation", True), ("distance", False)])
def test_augmented_step_vs_oracle(graph, raw, adj3d):
    ps.check_augmented_step("cpu", adj3d, graph=graph, raw=raw, b=4, t_len=2)


def test_fft_features(golden_fft):
    ps.check_fft_features("cpu", golden_fft)


def test_torch_ops_direct_and_opcheck(adj3d):
    ps.check_torch_ops("cpu", adj3d)


def test_batch_major_input_without_copy(adj3d):
    ps.check_batch_major_input("cpu", adj3d)


def test_zero_diffusion_steps_model_and_dconv(adj3d):
    """max_diffusion_step = 0 (the reference's `pass` branch, cell.py:80-81): one hop matrix = the identity.  Whole
    classification model vs the oracle, and the stand-alone DiffusionGraphConv forward + backward."""
    import torch
    from eeg_gnn_ssl_amd import DiffusionGraphConv
    from oracle import dcrnn_oracle as orc
    import cases
    ps.check_vs_oracle_random("cpu", "dual_random_walk", 8, 16, 2, 3, 2, 4, adj3d, seed=2, lengths=[3, 1], k=0)
    g = torch.Generator().manual_seed(1)
    mod = DiffusionGraphConv(1, 8, 16, 19, 0, 32, filter_type="laplacian")
    x = torch.randn(2, 19 * 8, generator=g, requires_grad=True)
    s = torch.randn(2, 19 * 16, generator=g, requires_grad=True)
    sup = cases.supports_for("laplacian", adj3d, 2)
    out = mod(sup, x, s, 32)
    ref = orc.diffusion_conv(sup, x.detach(), s.detach(), mod.weight.detach(), mod.biases.detach(), 19, 0)
    ps.assert_close(out.detach().numpy(), ref.numpy(), "dconv K=0")
    out.sum().backward()
    assert x.grad is not None and mod.weight.grad is not None and tuple(mod.weight.shape) == (24, 32)


def test_randomized_small_shapes_vs_oracle(emulator):
    """the randomized parity generator of the GPU suite (tests/fuzz_gpu.py) on the emulator build of the kernel sources, small shapes"""
    import fuzz_gpu
    done, refused, kinks = fuzz_gpu.run(cases=40, seed=11, dev="cpu", small=True)
    assert done["model"] + done["decoder"] == 40 and kinks <= 1


def test_randomized_small_shapes_through_the_spectral_form(emulator):
    """the spectral-form generator of tests/fuzz_gpu.py (--spectral on the GPU box) on the emulator, small shapes: every case against the oracle
    and against the general path of the same sources"""
    import fuzz_gpu
    assert fuzz_gpu.run_spectral(cases=8, seed=4, dev="cpu", small=True) == 8
