"""Pass-by-pass check of the engine's DEFAULT calls (exact candidate pruning + pass memo on, no score tables) against the
torch-CPU restatement of the reference (oracle/torch_port.py, itself pinned to the reference's golden files).

The fused entry points return only the final intervals, and after one near-tie resolved differently two implementations
continue from different intervals -- their later score tables are no longer comparable.  So the engine is called with
search_round = 1, 2, 3: the kernels are deterministic, hence the R-round call repeats the (R-1)-round call and adds one round,
and the three results are the engine's whole trajectory -- the input and the output of every search pass (the weight search
of round r sees the activation interval of round r-1, reference linear.py:476; the activation search sees the weights just
selected, :512; matmul.py:496,535 likewise).  The port then scores each pass FROM THE ENGINE'S OWN INPUT and the engine's
selection must be the port's argmax or a near-tie by the port's scores (TIE_RTOL): a hard bound on every pass, wherever a tie
fell earlier.  Every call asserts through the process-wide counters (p4v_prune_counters) that the three-stage pruned passes
-- not the full sweeps -- produced the result.
"""
import numpy as np
import torch

from tests.helpers import TIE_RTOL, assert_argmax_tie_aware


def _lookup(table, got, what):
    """Index of `got[j]` in column j of the candidate table (bit-exact: intervals are gathered from that fp32 table)."""
    table = np.asarray(table, dtype=np.float32).reshape(table.shape[0], -1)
    got = np.asarray(got, dtype=np.float32).reshape(-1)
    assert table.shape[1] == got.size, (what, table.shape, got.shape)
    idx = np.empty(got.size, dtype=np.int64)
    for j in range(got.size):
        hit = np.nonzero(table[:-1, j] == got[j])[0]          # the last entry of the table is never searched (linear.py:466)
        assert hit.size > 0, f"{what}: block {j}: interval {got[j]!r} is not an entry of the candidate table"
        idx[j] = hit[0]
    return idx


def _expect_staged(eng, what, min_staged=1):
    c = eng.prune_counters(reset=True)
    assert c["staged"] >= min_staged and c["not_eligible"] == 0 and c["kept_full_sweep"] == 0, \
        f"{what}: the default call did not run on the pruned passes: {c}"
    return c


class _Memo:
    """The port's tables by pass input (rounds 2-3 usually repeat an input: converged alternation)."""

    def __init__(self):
        self.d = {}

    def get(self, key, fn):
        k = np.asarray(key, dtype=np.float32).tobytes()
        if k not in self.d:
            self.d[k] = fn()
        return self.d[k]


def follow_linear(eng, *, weight, bias, x, out, grad, hp, rounds=3, what="linear", expect_pruned=True, tie_rtol=TIE_RTOL):
    """`hp`: engine.linear_calibrate keywords without search_round (w_bit, a_bit, metric, eq_*, n_V, postgelu ...).
    Tensors are CPU float32.  Returns (flips, final w_interval, final a_interval) of the engine."""
    from oracle.torch_port import TorchLinear
    dev = torch.device("cuda")
    args = dict(weight=weight.to(dev), bias=None if bias is None else bias.to(dev), x=x.to(dev), out=out.to(dev),
                grad=None if grad is None else grad.to(dev))
    traj = []
    for R in range(1, rounds + 1):
        eng.prune_counters(reset=True)
        w_iv, a_iv, sc, best = eng.linear_calibrate(search_round=R, n_H=1, n_a=1, **args, **hp)
        torch.cuda.synchronize()
        assert sc is None and best is None
        if expect_pruned:
            _expect_staged(eng, f"{what} R={R}")
        traj.append((w_iv.cpu(), a_iv.cpu()))
    port = TorchLinear(weight, bias, **{k: v for k, v in hp.items() if k not in ("n_H", "n_a")})
    w0, a0, w_c, a_c = port.initial(x)
    flips, a_cur = 0, a0
    memo_w, memo_a = _Memo(), _Memo()
    for r in range(rounds):
        w_got, a_got = traj[r]
        tab = memo_w.get(a_cur.numpy(), lambda: port.score_w(x, out, grad, w_c, a_cur).numpy())
        flips += assert_argmax_tie_aware(_lookup(w_c.numpy(), w_got.numpy(), f"{what} round {r} w"), tab, tie_rtol,
                                         what=f"{what} round {r} weight search")
        tab = memo_a.get(w_got.numpy(), lambda: port.score_a(x, out, grad, a_c, w_got).numpy())
        flips += assert_argmax_tie_aware(_lookup(a_c.numpy()[:, None], a_got.numpy(), f"{what} round {r} a"), tab, tie_rtol,
                                         what=f"{what} round {r} activation search")
        a_cur = a_got.reshape(())
    return flips, traj[-1][0], traj[-1][1]


def follow_matmul(eng, *, A, B, out, grad, hp, sos, rounds=3, what="matmul", expect_pruned=True, tie_rtol=TIE_RTOL):
    """A [b,H,M,K], B [b,H,K,N] (any strides), out / grad [b,H,M,N]: CPU float32.  Returns (flips, A_iv, B_iv, split)."""
    from oracle.torch_port import TorchMatMul
    dev = torch.device("cuda")
    Bd = B.to(dev)
    if not B.is_contiguous():                      # q.k^T hands B over as a transposed view (utils/models.py:16)
        Bd = B.transpose(-2, -1).contiguous().to(dev).transpose(-2, -1)
    args = dict(A=A.to(dev), B=Bd, out=out.to(dev), grad=None if grad is None else grad.to(dev))
    traj = []
    for R in range(1, rounds + 1):
        eng.prune_counters(reset=True)
        A_iv, B_iv, split, sc, best = eng.matmul_calibrate(search_round=R, sos=sos, **args, **hp)
        torch.cuda.synchronize()
        assert sc is None and best is None
        if expect_pruned:
            _expect_staged(eng, f"{what} R={R}")
        traj.append((A_iv.cpu(), B_iv.cpu(), None if split is None else split.cpu()))
    port = TorchMatMul(sos=sos, **hp)
    A0, B0, A_c, B_c = port.initial(A, B)
    flips, B_cur = 0, B0
    memo_A, memo_B = _Memo(), _Memo()
    for r in range(rounds):
        A_got, B_got, split_got = traj[r]
        if sos:
            tab = memo_A.get(np.zeros(1), lambda: port.score_split(A, B, out, grad).numpy())
            i = np.nonzero(np.asarray(port.SPLITS, dtype=np.float32) == np.float32(float(split_got)))[0]
            assert i.size == 1, f"{what}: split {float(split_got)!r} is not a power of two 2^-i, i < 20"
            flips += assert_argmax_tie_aware(i[:1], tab, tie_rtol, what=f"{what} round {r} split search")
            assert float(A_got) == float(np.float32(float(split_got)) / np.float32(port.Aq - 1)), f"{what}: A_interval != split / (qmax - 1)"
            As_key = split_got
        else:
            tab = memo_A.get(B_cur.numpy(), lambda: port.score_A(A, B, out, grad, A_c, B_cur).numpy())
            flips += assert_argmax_tie_aware(_lookup(A_c.numpy(), A_got.numpy(), f"{what} round {r} A"), tab, tie_rtol,
                                             what=f"{what} round {r} A search")
            As_key = A_got
        tab = memo_B.get(As_key.numpy(), lambda: port.score_B(port.quant_A(A, A_got, split_got), B, out, grad, B_c).numpy())
        flips += assert_argmax_tie_aware(_lookup(B_c.numpy(), B_got.numpy(), f"{what} round {r} B"), tab, tie_rtol,
                                         what=f"{what} round {r} B search")
        B_cur = B_got
    return (flips,) + traj[-1]


def follow_conv(eng, *, weight, bias, x, out, grad, stride, hp, channelwise=True, what="conv", expect_pruned=True, tie_rtol=TIE_RTOL):
    """Patch embedding (a_bit = 32: no activation search, every round repeats the weight search, conv.py:600)."""
    from oracle.torch_port import TorchConv
    dev = torch.device("cuda")
    eng.prune_counters(reset=True)
    w_iv, a_iv, sc, best = eng.conv_calibrate(weight=weight.to(dev), bias=bias.to(dev), x=x.to(dev), out=out.to(dev),
                                              grad=None if grad is None else grad.to(dev), stride=(stride, stride),
                                              padding=(0, 0), dilation=(1, 1), channelwise=channelwise, search_round=3, **hp)
    torch.cuda.synchronize()
    assert sc is None and best is None
    if expect_pruned:
        _expect_staged(eng, what)
    port = TorchConv(weight, bias, stride=stride, channelwise=channelwise, **hp)
    w0, w_c = port.initial()
    tab = port.score_w(x, out, grad, w_c).numpy()
    table = w_c.numpy().reshape(w_c.shape[0], -1)
    flips = assert_argmax_tie_aware(_lookup(table, w_iv.cpu().numpy(), what), tab, tie_rtol, what=f"{what} weight search")
    return flips, w_iv.cpu()
