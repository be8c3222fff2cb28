"""GPU parity of the GRANULAR entry points: one part of calibration_step2
per C-ABI call (p4v_amax_init_*, p4v_*_search_*, p4v_score_argmax_gather), driven the way the reference calls its
_initialize_intervals / _search_best_*_interval methods one by one.  They launch the same kernels as the fused
p4v_*_calibrate calls, whose tables are checked against the reference's golden vectors in test_hip_parity.py -- so the
bar here is BIT-IDENTITY with the fused call: every score table, every selected index, every interval."""
import numpy as np
import pytest
import torch

from tests.helpers import golden_names, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from ptq4vit_amd import engine
    return engine


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _cands(eng, p, iv):
    """Reference linear.py:544-545: fp32 multipliers x the INITIAL interval, computed once before the round loop."""
    mult = eng.candidate_multipliers(p["eq_alpha"], p["eq_beta"], p["eq_n"], iv.device)
    return mult.view(-1, 1) * iv.reshape(1, -1)


def _same(a, b, what):
    np.testing.assert_array_equal(a.detach().cpu().numpy(), b.detach().cpu().numpy(), err_msg=what)


@pytest.mark.parametrize("name", golden_names("linear_") + golden_names("postgelu_"))
def test_linear_granular_sequence_is_bit_identical_to_the_fused_call(eng, name):
    g = load_golden(name)
    p = g["params"]
    nV, nH, nA, R = p["n_V"], p.get("n_H", 1), p.get("n_a", 1), p["search_round"]
    common = dict(weight=_t(g["weight"]), bias=_t(g["bias"]) if "bias" in g else None, x=_t(g["x"]), out=_t(g["out"]),
                  grad=_t(g["grad"]), w_bit=p["w_bit"], a_bit=p["a_bit"], metric=p["metric"], eq_n=p["eq_n"],
                  n_V=nV, n_H=nH, n_a=nA, postgelu=p["postgelu"])
    w_f, a_f, sc_f, be_f = eng.linear_calibrate(eq_alpha=p["eq_alpha"], eq_beta=p["eq_beta"], search_round=R,
                                                want_scores=True, **common)
    st = eng.LinearStepper(**common)
    w, a = st.init_intervals()
    wc, ac = _cands(eng, p, w), _cands(eng, p, a)
    for r in range(R):
        w, sw, bw = st.search_w(wc, w, a, want_scores=True)
        _same(sw, sc_f[r, 0], f"{name} round {r} w scores"); _same(bw, be_f[r, 0], f"{name} round {r} w argmax")
        a, sa, ba = st.search_a(ac, w, a, want_scores=True)
        # (the a-search table of the fused call uses column 0 only; with n_a > 1 it is the table of group 0)
        _same(sa[:, :1], sc_f[r, 1][:, :1], f"{name} round {r} a scores"); _same(ba[:1], be_f[r, 1][:1], f"{name} round {r} a argmax")
    _same(w, w_f, f"{name} w_interval"); _same(a, a_f, f"{name} a_interval")


@pytest.mark.parametrize("name", golden_names("matmul_"))
def test_matmul_granular_sequence_is_bit_identical_to_the_fused_call(eng, name):
    g = load_golden(name)
    p = g["params"]
    sos, R = p["sos"], p["search_round"]
    common = dict(A=_t(g["A"]), B=_t(g["B"]), out=_t(g["out"]), grad=_t(g["grad"]), A_bit=p["A_bit"], B_bit=p["B_bit"],
                  metric=p["metric"], eq_n=p["eq_n"], sos=sos)
    A_f, B_f, split_f, sc_f, be_f = eng.matmul_calibrate(eq_alpha=p["eq_alpha"], eq_beta=p["eq_beta"], search_round=R,
                                                         want_scores=True, **common)
    st = eng.MatMulStepper(**common)
    A_iv, B_iv = st.init_intervals()
    assert (A_iv is None) == sos
    Bc = _cands(eng, p, B_iv)
    Ac = None if sos else _cands(eng, p, A_iv)
    split = None
    for r in range(R):
        if sos:
            split, A_iv, s1, b1 = st.search_split(want_scores=True)
            _same(s1, sc_f[r, 0][:20, :1], f"{name} round {r} split scores"); _same(b1, be_f[r, 0][:1], f"{name} round {r} split argmax")
        else:
            A_iv, s1, b1 = st.search_A(Ac, A_iv, B_iv, want_scores=True)
            _same(s1, sc_f[r, 0], f"{name} round {r} A scores"); _same(b1, be_f[r, 0], f"{name} round {r} A argmax")
        B_iv, s2, b2 = st.search_B(Bc, A_iv, B_iv, split=split, want_scores=True)
        _same(s2, sc_f[r, 1], f"{name} round {r} B scores"); _same(b2, be_f[r, 1], f"{name} round {r} B argmax")
    _same(A_iv, A_f, f"{name} A_interval"); _same(B_iv, B_f, f"{name} B_interval")
    if sos:
        _same(split, split_f, f"{name} split")
        assert float(split.cpu()) == float(g["split"])


@pytest.mark.parametrize("name", golden_names("conv_"))
def test_conv_granular_sequence_is_bit_identical_to_the_fused_call(eng, name):
    g = load_golden(name)
    p = g["params"]
    R, st_ = p["search_round"], p["stride"]
    common = dict(weight=_t(g["weight"]), bias=_t(g["bias"]), x=_t(g["x"]), out=_t(g["out"]), grad=_t(g["grad"]),
                  stride=(st_, st_), padding=(0, 0), dilation=(1, 1), w_bit=p["w_bit"], a_bit=p["a_bit"],
                  metric=p["metric"], eq_n=p["eq_n"], channelwise=p["channelwise"])
    w_f, a_f, sc_f, be_f = eng.conv_calibrate(eq_alpha=p["eq_alpha"], eq_beta=p["eq_beta"], search_round=R,
                                              want_scores=True, **common)
    st = eng.ConvStepper(**common)
    w, a = st.init_intervals()
    wc, ac = _cands(eng, p, w), _cands(eng, p, a)
    aq = p["a_bit"] < 32
    for r in range(R):
        w, sw, bw = st.search_w(wc, w, a, want_scores=True)
        _same(sw, sc_f[r, 0], f"{name} round {r} w scores"); _same(bw, be_f[r, 0], f"{name} round {r} w argmax")
        if aq:
            a, sa, ba = st.search_a(ac, w, a, want_scores=True)
            _same(sa[:, :1], sc_f[r, 1][:, :1], f"{name} round {r} a scores"); _same(ba[:1], be_f[r, 1][:1], f"{name} round {r} a argmax")
        else:
            with pytest.raises(RuntimeError, match="a_bit >= 32"):
                st.search_a(ac, w, a)
    _same(w, w_f, f"{name} w_interval"); _same(a, a_f, f"{name} a_interval")


# ---- the module classes' per-pass methods (the names SURVEY.md s8 rows a4-a13 list) ----------------------------------
def _drive(m, first, second, cands1, cands2, rounds):
    for _ in range(rounds):
        getattr(m, first)(cands1)
        getattr(m, second)(cands2)


def _mult5(m, dev):
    from ptq4vit_amd import engine
    return engine.candidate_multipliers(m.eq_alpha, m.eq_beta, m.eq_n, dev)


@pytest.mark.parametrize("name", ["linear_qkv_hessian_w8a8", "postgelu_hessian_w8a8", "linear_blocks_nH2_na2"])
def test_linear_module_per_pass_methods_reproduce_calibration_step2(name):
    from ptq4vit_amd.quant_layers.linear import PostGeluPTQSLBatchingQuantLinear, PTQSLBatchingQuantLinear
    g = load_golden(name)
    p = g["params"]
    cls = PostGeluPTQSLBatchingQuantLinear if p["postgelu"] else PTQSLBatchingQuantLinear

    def make():
        m = cls(g["weight"].shape[1], g["weight"].shape[0], bias="bias" in g, w_bit=p["w_bit"], a_bit=p["a_bit"],
                metric=p["metric"], search_round=p["search_round"], eq_alpha=p["eq_alpha"], eq_beta=p["eq_beta"],
                eq_n=p["eq_n"], n_V=p["n_V"], n_H=p.get("n_H", 1), n_a=p.get("n_a", 1)).cuda()
        m.weight.data.copy_(_t(g["weight"]))
        if "bias" in g:
            m.bias.data.copy_(_t(g["bias"]))
        m.raw_input, m.raw_out, m.raw_grad = _t(g["x"]), _t(g["out"]), _t(g["grad"])
        return m
    fused = make()
    fused.calibration_step2()
    m = make()
    m._initialize_calib_parameters()
    m._initialize_intervals()
    mult = _mult5(m, m.weight.device)
    wc = mult.view(-1, 1, 1, 1, 1) * m.w_interval.unsqueeze(0)                    # reference linear.py:544
    ac = mult.view(1, 1, -1) * m._positive_a_interval().unsqueeze(-1)             # reference linear.py:545: (n_a, 1, eq_n+1)
    _drive(m, "_search_best_w_interval", "_search_best_a_interval", wc, ac, m.search_round)
    _same(m.w_interval, fused.w_interval, name + " w_interval")
    _same(m._positive_a_interval(), fused._positive_a_interval(), name + " a_interval")
    assert m.w_interval.shape == fused.w_interval.shape and m._positive_a_interval().shape == fused._positive_a_interval().shape


@pytest.mark.parametrize("name", ["matmul_qk_hessian_w8a8", "matmul_sos_hessian_w8a8"])
def test_matmul_module_per_pass_methods_reproduce_calibration_step2(name):
    from ptq4vit_amd.quant_layers.matmul import PTQSLBatchingQuantMatMul, SoSPTQSLBatchingQuantMatMul
    g = load_golden(name)
    p = g["params"]
    cls = SoSPTQSLBatchingQuantMatMul if p["sos"] else PTQSLBatchingQuantMatMul

    def make():
        m = cls(A_bit=p["A_bit"], B_bit=p["B_bit"], metric=p["metric"], search_round=p["search_round"],
                eq_alpha=p["eq_alpha"], eq_beta=p["eq_beta"], eq_n=p["eq_n"])
        m.raw_input, m.raw_out, m.raw_grad = [_t(g["A"]), _t(g["B"])], _t(g["out"]), _t(g["grad"])
        return m
    fused = make()
    fused.calibration_step2()
    m = make()
    m._initialize_calib_parameters()
    m._initialize_intervals()
    mult = _mult5(m, m.B_interval.device).view(-1, 1, 1, 1, 1, 1, 1, 1)
    Bc = mult * m.B_interval.unsqueeze(0)
    if p["sos"]:
        Ac = torch.tensor([2.0 ** (-i) for i in range(20)])                        # reference matmul.py:636
    else:
        Ac = mult * m.A_interval.unsqueeze(0)
    _drive(m, "_search_best_A_interval", "_search_best_B_interval", Ac, Bc, m.search_round)
    _same(m.A_interval, fused.A_interval, name + " A_interval"); _same(m.B_interval, fused.B_interval, name + " B_interval")
    assert m.A_interval.shape == fused.A_interval.shape and m.B_interval.shape == fused.B_interval.shape
    if p["sos"]:
        _same(m.split, fused.split, name + " split")
        with pytest.raises(NotImplementedError):
            m._search_best_A_interval(torch.tensor([0.5, 0.25]))


@pytest.mark.parametrize("name", ["conv_channelwise_hessian", "conv_layerwise_cosine", "conv_channelwise_hessian_a8_overlap"])
def test_conv_module_per_pass_methods_reproduce_calibration_step2(name):
    from ptq4vit_amd.quant_layers.conv import BatchingEasyQuantConv2d, ChannelwiseBatchingQuantConv2d
    g = load_golden(name)
    p = g["params"]
    cls = ChannelwiseBatchingQuantConv2d if p["channelwise"] else BatchingEasyQuantConv2d
    oc, ic, k, _ = g["weight"].shape

    def make():
        m = cls(ic, oc, k, stride=p["stride"], w_bit=p["w_bit"], a_bit=p["a_bit"], metric=p["metric"],
                search_round=p["search_round"], eq_alpha=p["eq_alpha"], eq_beta=p["eq_beta"], eq_n=p["eq_n"]).cuda()
        m.weight.data.copy_(_t(g["weight"])); m.bias.data.copy_(_t(g["bias"]))
        m.raw_input, m.raw_out, m.raw_grad = _t(g["x"]), _t(g["out"]), _t(g["grad"])
        return m
    fused = make()
    fused.calibration_step2()
    m = make()
    m._initialize_calib_parameters()
    m._initialize_intervals()
    mult = _mult5(m, m.weight.device)
    wc = mult.view(-1, 1, 1, 1, 1) * m.w_interval.unsqueeze(0)                     # reference conv.py:594
    ac = mult * m.a_interval.reshape(-1)[0]
    for _ in range(m.search_round):
        m._search_best_w_interval(wc)
        if m.a_bit < 32:
            m._search_best_a_interval(ac)                                           # reference conv.py:600
    _same(m.w_interval, fused.w_interval, name + " w_interval")
    _same(m.a_interval.reshape(-1), fused.a_interval.reshape(-1), name + " a_interval")
    assert m.w_interval.shape == fused.w_interval.shape


def test_score_argmax_gather_first_maximum_and_nan_rule(eng):
    """argmax(dim=0) takes the first index on ties and treats NaN as the maximum."""
    rng = np.random.default_rng(0)
    sc = rng.standard_normal((37, 9)).astype(np.float32)
    sc[5, 1] = sc[:, 1].max() + 1; sc[20, 1] = sc[5, 1]          # tie: first wins
    sc[11, 2] = np.nan; sc[30, 2] = np.nan                       # NaN: first NaN wins
    sc[0, 3] = np.inf
    sc[:, 4] = -np.inf                                           # all equal: index 0
    cands = rng.standard_normal((38, 9)).astype(np.float32)
    iv, best = eng.score_argmax_gather(_t(sc), _t(cands))
    want = np.array([int(np.flatnonzero(np.isnan(c))[0]) if np.isnan(c).any() else int(np.argmax(c)) for c in sc.T])
    np.testing.assert_array_equal(best.cpu().numpy(), want)
    np.testing.assert_array_equal(best.cpu().numpy(), torch.argmax(torch.from_numpy(sc), dim=0).numpy())
    np.testing.assert_array_equal(iv.cpu().numpy(), cands[want, np.arange(9)])


@pytest.mark.parametrize("name", ["linear_qkv_hessian_w8a8", "postgelu_hessian_w8a8", "linear_cosine_w8a8"])
def test_linear_passes_honour_a_custom_candidate_table(eng, name):
    """The granular calls take the candidate table as an INPUT (the reference passes `weight_interval_candidates` /
    `input_interval_candidates` as arguments, linear.py:455,497): a non-uniform, per-block table must give the
    oracle's scores and selections for exactly that table."""
    from oracle import ptq4vit_oracle as orc
    from tests.helpers import assert_argmax_tie_aware, assert_scores_close
    g = load_golden(name)
    p = g["params"]
    nV, eq_n = p["n_V"], p["eq_n"]
    o = orc.LinearOracle(g["weight"], g.get("bias"), w_bit=p["w_bit"], a_bit=p["a_bit"], metric=p["metric"],
                         search_round=1, eq_alpha=p["eq_alpha"], eq_beta=p["eq_beta"], eq_n=eq_n, n_V=nV,
                         postgelu=p["postgelu"])
    o.initialize_intervals(g["x"])
    rng = np.random.default_rng(5)
    # geometric grid with a different ratio per block, shuffled: nothing like the reference's uniform grid
    ratios = np.exp(np.linspace(np.log(0.35), np.log(1.4), eq_n + 1)).astype(np.float32)
    w_tab = np.stack([rng.permutation(ratios) * o.w_interval.reshape(-1)[v] * (1 + 0.1 * v) for v in range(nV)], axis=1)  # (eq_n+1, nV)
    a_tab = rng.permutation(ratios) * o.a_interval.reshape(-1)[0]                                                     # (eq_n+1,)
    st = eng.LinearStepper(weight=_t(g["weight"]), bias=_t(g["bias"]) if "bias" in g else None, x=_t(g["x"]),
                           out=_t(g["out"]), grad=_t(g["grad"]), w_bit=p["w_bit"], a_bit=p["a_bit"], metric=p["metric"],
                           eq_n=eq_n, n_V=nV, n_H=1, n_a=1, postgelu=p["postgelu"])
    w0, a0 = st.init_intervals()
    np.testing.assert_array_equal(w0.cpu().numpy(), o.w_interval.reshape(-1))
    np.testing.assert_array_equal(a0.cpu().numpy(), o.a_interval.reshape(-1))
    grad = g["grad"] if p["metric"] == "hessian" else None
    # weight pass
    o.search_w(g["x"], g["out"], grad, w_tab.reshape(eq_n + 1, nV, 1, 1, 1))
    w1, sw, bw = st.search_w(_t(w_tab), w0, a0, want_scores=True)
    ref_w = o.trace[-1][1]
    assert_scores_close(sw.cpu().numpy(), ref_w, what=name + " w table")
    flips = assert_argmax_tie_aware(bw.cpu().numpy(), ref_w, what=name + " w argmax")
    if flips == 0:
        np.testing.assert_array_equal(w1.cpu().numpy(), o.w_interval.reshape(-1))
    # activation pass against the weight interval just selected (by the oracle: keeps the two sides comparable)
    w_sel = _t(o.w_interval.reshape(-1).copy())
    o.search_a(g["x"], g["out"], grad, a_tab.reshape(1, 1, eq_n + 1))
    a1, sa, ba = st.search_a(_t(a_tab), w_sel, a0, want_scores=True)
    ref_a = o.trace[-1][1].reshape(-1, 1)
    assert_scores_close(sa.cpu().numpy()[:, :1], ref_a, what=name + " a table")
    flips = assert_argmax_tie_aware(ba.cpu().numpy()[:1], ref_a, what=name + " a argmax")
    if flips == 0:
        np.testing.assert_array_equal(a1.cpu().numpy(), o.a_interval.reshape(-1))


def test_matmul_passes_honour_a_custom_candidate_table(eng):
    """matmul.py:483-563 with caller-supplied (non-uniform, per-head) candidate tables, against the oracle."""
    from oracle import ptq4vit_oracle as orc
    from tests.helpers import assert_argmax_tie_aware, assert_scores_close
    name = "matmul_qk_hessian_w8a8"
    g = load_golden(name)
    p = g["params"]
    eq_n, H = p["eq_n"], g["A"].shape[1]
    o = orc.MatMulOracle(A_bit=p["A_bit"], B_bit=p["B_bit"], metric=p["metric"], search_round=1, eq_alpha=p["eq_alpha"],
                         eq_beta=p["eq_beta"], eq_n=eq_n)
    o.initialize_intervals(g["A"], g["B"])
    rng = np.random.default_rng(11)
    ratios = np.exp(np.linspace(np.log(0.3), np.log(1.5), eq_n + 1)).astype(np.float32)
    A_tab = np.stack([rng.permutation(ratios) * o.A_interval.reshape(-1)[h] for h in range(H)], axis=1)   # (eq_n+1, H)
    B_tab = np.stack([rng.permutation(ratios) * o.B_interval.reshape(-1)[h] for h in range(H)], axis=1)
    st = eng.MatMulStepper(A=_t(g["A"]), B=_t(g["B"]), out=_t(g["out"]), grad=_t(g["grad"]), A_bit=p["A_bit"],
                           B_bit=p["B_bit"], metric=p["metric"], eq_n=eq_n)
    A0, B0 = st.init_intervals()
    np.testing.assert_array_equal(A0.cpu().numpy(), o.A_interval.reshape(-1))
    np.testing.assert_array_equal(B0.cpu().numpy(), o.B_interval.reshape(-1))
    o._search_blockwise("A", g["A"], g["B"], g["out"], g["grad"], A_tab.reshape(eq_n + 1, 1, H, 1, 1, 1, 1, 1))
    A1, sA, bA = st.search_A(_t(A_tab), A0, B0, want_scores=True)
    assert_scores_close(sA.cpu().numpy(), o.trace[-1][1], what="A table")
    if assert_argmax_tie_aware(bA.cpu().numpy(), o.trace[-1][1], what="A argmax") == 0:
        np.testing.assert_array_equal(A1.cpu().numpy(), o.A_interval.reshape(-1))
    A_sel = _t(o.A_interval.reshape(-1).copy())
    o._search_blockwise("B", g["A"], g["B"], g["out"], g["grad"], B_tab.reshape(eq_n + 1, 1, H, 1, 1, 1, 1, 1))
    B1, sB, bB = st.search_B(_t(B_tab), A_sel, B0, want_scores=True)
    assert_scores_close(sB.cpu().numpy(), o.trace[-1][1], what="B table")
    if assert_argmax_tie_aware(bB.cpu().numpy(), o.trace[-1][1], what="B argmax") == 0:
        np.testing.assert_array_equal(B1.cpu().numpy(), o.B_interval.reshape(-1))


def test_conv_pass_honours_a_custom_candidate_table(eng):
    """conv.py:526-557 with a caller-supplied per-channel candidate table, against the oracle."""
    from oracle import ptq4vit_oracle as orc
    from tests.helpers import assert_argmax_tie_aware, assert_scores_close
    g = load_golden("conv_channelwise_hessian")
    p = g["params"]
    eq_n, oc = p["eq_n"], g["weight"].shape[0]
    o = orc.ConvOracle(g["weight"], g["bias"], stride=p["stride"], w_bit=p["w_bit"], a_bit=p["a_bit"], metric=p["metric"],
                       search_round=1, eq_alpha=p["eq_alpha"], eq_beta=p["eq_beta"], eq_n=eq_n, channelwise=True)
    o.initialize_intervals(g["x"])
    rng = np.random.default_rng(13)
    ratios = np.exp(np.linspace(np.log(0.3), np.log(1.5), eq_n + 1)).astype(np.float32)
    w_tab = np.stack([rng.permutation(ratios) * np.asarray(o.w_interval).reshape(-1)[c] for c in range(oc)], axis=1)   # (eq_n+1, oc)
    st = eng.ConvStepper(weight=_t(g["weight"]), bias=_t(g["bias"]), x=_t(g["x"]), out=_t(g["out"]), grad=_t(g["grad"]),
                         stride=(p["stride"],) * 2, padding=(0, 0), dilation=(1, 1), w_bit=p["w_bit"], a_bit=p["a_bit"],
                         metric=p["metric"], eq_n=eq_n, channelwise=True)
    w0, a0 = st.init_intervals()
    np.testing.assert_array_equal(w0.cpu().numpy(), np.asarray(o.w_interval).reshape(-1))
    o.search_w(g["x"], g["out"], g["grad"], w_tab.reshape(eq_n + 1, oc, 1, 1, 1))
    w1, sw, bw = st.search_w(_t(w_tab), w0, a0, want_scores=True)
    assert_scores_close(sw.cpu().numpy(), o.trace[-1][1], what="conv w table")
    if assert_argmax_tie_aware(bw.cpu().numpy(), o.trace[-1][1], what="conv w argmax") == 0:
        np.testing.assert_array_equal(w1.cpu().numpy(), np.asarray(o.w_interval).reshape(-1))
