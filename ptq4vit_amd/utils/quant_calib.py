"""Calibration orchestrators -- API mirror of the reference's utils/quant_calib.py.

``HessianQuantCalibrator.batching_quant_calib()`` (reference quant_calib.py:300-378) is the entry point every
reference experiment uses.  Differences in HOW (not WHAT):

* capture is ONE set of forward/backward sub-batch passes hooking every module this rank owns, with the
  hooked tensors kept on the GPU (the reference repeats the whole set of passes once per module and moves
  every hooked tensor to the host, quant_calib.py:317-356).  With ``sequential=False`` every module stays in
  "raw" mode during capture, so the captured tensors are identical either way;
* ``module.calibration_step2()`` runs on the GPU through the C ABI;
* with ``torch.distributed`` initialised the modules are sharded over the ranks (one process per GPU) and the
  calibrated intervals are exchanged with one all-gather at the end (ptq4vit_amd/utils/shard.py).
"""
import torch
import torch.nn.functional as F

from ..quant_layers.conv import MinMaxQuantConv2d
from ..quant_layers.linear import MinMaxQuantLinear
from ..quant_layers.matmul import MinMaxQuantMatMul
from . import shard

try:
    from tqdm import tqdm
except Exception:  # pragma: no cover
    def tqdm(x, **k):
        return x


def _dev_of(net):
    for p in net.parameters():
        return p.device
    return torch.device("cpu")


# ---- hook functions (reference quant_calib.py:173-201); tensors stay on their device ------------------
def grad_hook(module, grad_input, grad_output):
    if module.raw_grad is None:
        module.raw_grad = []
    module.raw_grad.append(grad_output[0].detach())


def linear_forward_hook(module, input, output):
    if module.raw_input is None:
        module.raw_input = []
    if module.raw_out is None:
        module.raw_out = []
    module.raw_input.append(input[0].detach())
    module.raw_out.append(output.detach())


conv2d_forward_hook = linear_forward_hook


def matmul_forward_hook(module, input, output):
    if module.raw_input is None:
        module.raw_input = [[], []]
    if module.raw_out is None:
        module.raw_out = []
    module.raw_input[0].append(input[0].detach())
    module.raw_input[1].append(input[1].detach())
    module.raw_out.append(output.detach())


def _register(module, with_grad):
    hooks = []
    if isinstance(module, MinMaxQuantLinear):
        hooks.append(module.register_forward_hook(linear_forward_hook))
    if isinstance(module, MinMaxQuantConv2d):
        hooks.append(module.register_forward_hook(conv2d_forward_hook))
    if isinstance(module, MinMaxQuantMatMul):
        hooks.append(module.register_forward_hook(matmul_forward_hook))
    if with_grad:
        hooks.append(module.register_full_backward_hook(grad_hook))
    return hooks


def _concat(module, with_grad):
    """Lists of per-sub-batch tensors -> one tensor per cache (reference quant_calib.py:343-354)."""
    if isinstance(module, MinMaxQuantMatMul):
        module.raw_input = [torch.cat(t, dim=0) for t in module.raw_input]
    else:
        module.raw_input = torch.cat(module.raw_input, dim=0)
    module.raw_out = torch.cat(module.raw_out, dim=0)
    if with_grad:
        module.raw_grad = torch.cat(module.raw_grad, dim=0)


class QuantCalibrator:
    """Reference quant_calib.py:9-171: forward-mode calibration (calibration_step1 / calibration_step2(x))."""

    def __init__(self, net, wrapped_modules, calib_loader, sequential=True):
        self.net = net
        self.wrapped_modules = wrapped_modules
        self.calib_loader = calib_loader
        self.sequential = sequential
        self.calibrated = False

    def _run_loader(self):
        dev = _dev_of(self.net)
        with torch.no_grad():
            for inp, _ in self.calib_loader:
                self.net(inp.to(dev))

    def sequential_quant_calib(self):
        """One module at a time; already-calibrated predecessors run quantised (reference :28-55)."""
        for name, module in tqdm(self.wrapped_modules.items(), desc="Calibration"):
            for step in ("calibration_step1", "calibration_step2"):
                module.mode = step
                self._run_loader()
            module.mode = "quant_forward"
        for module in self.wrapped_modules.values():
            module.mode = "quant_forward"
        print("sequential calibration finished")

    def parallel_quant_calib(self):
        """All modules collect raw data first, then calibrate one by one (reference :57-93)."""
        for module in self.wrapped_modules.values():
            module.mode = "calibration_step1"
        self._run_loader()
        for name, module in tqdm(self.wrapped_modules.items(), desc="Calibration"):
            module.mode = "calibration_step2"
            self._run_loader()
            module.mode = "raw"
        for module in self.wrapped_modules.values():
            module.mode = "quant_forward"
        print("calibration finished")

    def quant_calib(self):
        print(f"prepare parallel calibration for {list(self.wrapped_modules)}")
        if self.sequential:
            self.sequential_quant_calib()
        else:
            self.parallel_quant_calib()
        self.calibrated = True

    def batching_quant_calib(self):
        """Cached-tensor calibration without gradients (reference :95-171); any non-hessian metric."""
        HessianQuantCalibrator.batching_quant_calib(self, with_grad=False)


class HessianQuantCalibrator(QuantCalibrator):
    """Reference quant_calib.py:203-378."""

    def __init__(self, net, wrapped_modules, calib_loader, sequential=False, batch_size=1,
                 cache_budget_bytes=96 << 30):
        super().__init__(net, wrapped_modules, calib_loader, sequential=sequential)
        self.batch_size = batch_size
        self.cache_budget_bytes = cache_budget_bytes
        self.timings = {}

    # ---- capture ------------------------------------------------------------------------------------
    def _raw_pred_softmax(self):
        dev = _dev_of(self.net)
        with torch.no_grad():
            for inp, _ in self.calib_loader:
                raw_pred = self.net(inp.to(dev))
                raw_pred_softmax = F.softmax(raw_pred, dim=-1).detach()
        return raw_pred_softmax

    def _capture(self, names, raw_pred_softmax, with_grad):
        """Forward (+ backward of the KL loss, reference :333-339) over the calibration set in sub-batches,
        with hooks on `names` only."""
        dev = _dev_of(self.net)
        bs = getattr(self, "batch_size", None) or self.calib_loader.batch_size
        hooks = []
        for n in names:
            m = self.wrapped_modules[n]
            m.raw_input = m.raw_out = None
            if hasattr(m, "metric"):
                m.raw_grad = None   # step 2 deletes the caches (reference linear.py:554): re-create for re-calibration
            hooks += _register(m, with_grad and hasattr(m, "metric"))
        for inp, _ in self.calib_loader:
            total = inp.shape[0]
            for st in range(0, total, bs):
                inp_ = inp[st:st + bs].to(dev)
                if with_grad:
                    self.net.zero_grad()
                    pred = self.net(inp_)
                    loss = F.kl_div(F.log_softmax(pred, dim=-1), raw_pred_softmax[st:st + bs], reduction="batchmean")
                    loss.backward()
                else:
                    with torch.no_grad():
                        self.net(inp_)
        for h in hooks:
            h.remove()
        for n in names:
            m = self.wrapped_modules[n]
            _concat(m, with_grad and hasattr(m, "metric"))

    def _estimate_cache_bytes(self, names):
        """One cheap probe forward of a single image to size the caches of `names`."""
        dev = _dev_of(self.net)
        sizes = {}
        hooks = []

        def mk(n):
            def hook(mod, inp, out):
                numel = sum(t.numel() for t in inp if torch.is_tensor(t)) + 2 * out.numel()
                sizes[n] = numel * 4
            return hook

        for n in names:
            hooks.append(self.wrapped_modules[n].register_forward_hook(mk(n)))
        total = 0
        with torch.no_grad():
            for inp, _ in self.calib_loader:
                total = inp.shape[0]
                self.net(inp[:1].to(dev))
                break
        for h in hooks:
            h.remove()
        return {n: s * total for n, s in sizes.items()}

    # ---- entry points ------------------------------------------------------------------------------
    def quant_calib(self):
        """Non-batching Hessian calibration (reference :216-298): same capture, then calibration_step2(x)."""
        self._calibrate(batching=False, with_grad=True)

    def batching_quant_calib(self, with_grad=True):
        self._calibrate(batching=True, with_grad=with_grad)

    def _calibrate(self, batching, with_grad):
        import time
        names = list(self.wrapped_modules)
        print(f"prepare parallel calibration for {names}")
        print("start hessian calibration")
        rank, world = shard.rank_world()
        owner = shard.assign_modules(self.wrapped_modules, world) if (world > 1 and not self.sequential) else {n: rank for n in names}
        mine = [n for n in names if owner[n] == rank]
        self.owner = owner
        t0 = time.time()
        raw_pred_softmax = self._raw_pred_softmax() if with_grad else None

        if self.sequential:
            groups = [[n] for n in mine]  # predecessors must already run quantised: one capture per module
        else:
            sizes = self._estimate_cache_bytes(mine)
            groups, cur, acc = [], [], 0
            for n in mine:
                if cur and acc + sizes.get(n, 0) > self.cache_budget_bytes:
                    groups.append(cur)
                    cur, acc = [], 0
                cur.append(n)
                acc += sizes.get(n, 0)
            if cur:
                groups.append(cur)
        t_cap = t_cal = 0.0
        for grp in groups:
            t1 = time.time()
            self._capture(grp, raw_pred_softmax, with_grad)
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            t2 = time.time()
            for n in tqdm(grp, desc="Hessian"):
                module = self.wrapped_modules[n]
                with torch.no_grad():
                    if batching:
                        module.calibration_step2()
                    elif isinstance(module, MinMaxQuantMatMul):
                        module.calibration_step2(module.raw_input[0], module.raw_input[1])
                    else:
                        module.calibration_step2(module.raw_input)
                module.mode = "quant_forward" if self.sequential else "raw"
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            t_cap += t2 - t1
            t_cal += time.time() - t2
        if world > 1 and not self.sequential:
            shard.exchange_intervals(self.wrapped_modules, owner)
        for module in self.wrapped_modules.values():
            module.mode = "quant_forward"
        self.timings = {"capture_s": t_cap, "search_s": t_cal, "total_s": time.time() - t0,
                        "modules": len(names), "owned": len(mine)}
        self.calibrated = True
        print("hessian calibration finished")
