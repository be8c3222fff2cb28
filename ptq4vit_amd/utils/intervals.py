"""Checkpoint of a calibration: the intervals (and splits) of every wrapped module as one `state_dict`-style file.

The reference keeps intervals as plain tensor attributes -- not Parameters or buffers, so `net.state_dict()` does not hold
them -- and its only persistence is `torch.save` of the whole calibrated model object (example/get_int.py:26-27).  This is
the same information without the pickle of the code: {module name: {attribute: tensor}} for the attributes other components
read (`w_interval`, `a_interval`, `A_interval`, `B_interval`, `split`), which is also exactly the payload of
the multi-GPU interval exchange (utils/shard.py).  `load_intervals` puts a freshly wrapped, uncalibrated network into the state
`batching_quant_calib()` leaves behind: intervals installed, `calibrated = True`, caches absent, mode `quant_forward`.
"""
import torch

INTERVAL_ATTRS = ("w_interval", "a_interval", "A_interval", "B_interval", "split")
FORMAT = "ptq4vit_amd.intervals/1"


def module_intervals(module):
    """{attribute: detached fp32 CPU tensor} of one calibrated module (the non-batching post-GELU class keeps
    `a_interval = [positive tensor, fixed negative float]`: the tensor is what is searched and stored)."""
    out = {}
    for a in INTERVAL_ATTRS:
        v = getattr(module, a, None)
        if v is None:
            continue
        if isinstance(v, (list, tuple)):
            v = v[0]
        out[a] = torch.as_tensor(v, dtype=torch.float32).detach().cpu().clone()
    return out


def install_intervals(module, values, device=None, clone=True):
    """Give `module` the calibrated state: `values` = {attribute: tensor} as produced by `module_intervals` (or received from
    the rank that searched the module).  Mirrors what the module's own `calibration_step2()` leaves behind.  `clone=False`: the
    caller hands over tensors nobody else writes (views of the freshly gathered exchange buffer: no copy kernel per attribute)."""
    if device is None:
        device = next((p.device for p in module.parameters()), torch.device("cpu")) if hasattr(module, "parameters") else torch.device("cpu")
    for a, val in values.items():
        if a not in INTERVAL_ATTRS:
            raise KeyError(f"unknown interval attribute {a}")
        val = torch.as_tensor(val, dtype=torch.float32).to(device)
        if clone:
            val = val.clone()
        cur = getattr(module, a, None)
        if isinstance(cur, (list, tuple)) or (a == "a_interval" and hasattr(module, "_set_a_interval") and
                                              getattr(module, "_postgelu", False) and not hasattr(module, "a_neg_interval")):
            module._set_a_interval(val)        # non-batching post-GELU class: [positive tensor, fixed negative float]
        else:
            setattr(module, a, val)
        # the search also fixed the group count of the blocked view (head-wise: matmul.py:411-417); the block sizes / paddings
        # themselves are recomputed from the operand shapes on the first quant_forward
        if a in ("A_interval", "B_interval") and val.dim() == 7 and hasattr(module, f"n_G_{a[0]}"):
            setattr(module, f"n_G_{a[0]}", int(val.shape[1]))
    if hasattr(module, "n_G_A") and getattr(module, "_sos", False) and hasattr(module, "n_G_B"):
        module.n_G_A = module.n_G_B            # what _search_on_gpu leaves behind (quant_input_A of the split class ignores it)
    module.calibrated = True
    for cache in ("raw_input", "raw_out", "raw_grad"):       # every class deletes its caches at the end of step 2
        if hasattr(module, cache):
            try:
                delattr(module, cache)
            except AttributeError:
                pass


def intervals_state_dict(wrapped_modules):
    return {name: dict(module_intervals(m), **{"__class__": type(m).__name__}) for name, m in wrapped_modules.items()}


def save_intervals(wrapped_modules, path, meta=None):
    """Write the intervals of every wrapped module (all must be calibrated) to `path` (torch.save of plain tensors / strings)."""
    missing = [n for n, m in wrapped_modules.items() if not getattr(m, "calibrated", False)]
    if missing:
        raise RuntimeError(f"save_intervals: {len(missing)} modules are not calibrated (first: {missing[0]})")
    torch.save({"format": FORMAT, "meta": dict(meta or {}), "modules": intervals_state_dict(wrapped_modules)}, str(path))


def _net_device(wrapped_modules):
    """The device the wrapped network lives on: that of any parameter / buffer of any wrapped module (MatMul wrappers have
    none of their own -- their intervals must follow the Linear / Conv modules of the same network)."""
    for m in wrapped_modules.values():
        for t in list(m.parameters()) + list(m.buffers()):
            return t.device
    return torch.device("cpu")


def load_intervals(wrapped_modules, path, strict=True, mode="quant_forward", device=None):
    """Install the intervals stored by `save_intervals` into freshly wrapped modules; `strict`: the module names and classes
    must match exactly.  Sets `mode` on every module it touched (None: leave the modes).  `device`: where the interval tensors
    go (default: the device of the network's parameters -- also for the parameter-less MatMul wrappers).  Returns the stored `meta`."""
    if device is None:
        device = _net_device(wrapped_modules)
    blob = torch.load(str(path), map_location="cpu", weights_only=True)
    if not isinstance(blob, dict) or blob.get("format") != FORMAT:
        raise ValueError(f"{path}: not an interval checkpoint ({FORMAT})")
    stored = blob["modules"]
    if strict:
        a, b = set(stored), set(wrapped_modules)
        if a != b:
            raise KeyError(f"load_intervals: module names differ (only in file: {sorted(a - b)[:4]}, only in net: {sorted(b - a)[:4]})")
    for name, vals in stored.items():
        m = wrapped_modules.get(name)
        if m is None:
            continue
        vals = dict(vals)
        cls = vals.pop("__class__", None)
        if strict and cls is not None and cls != type(m).__name__:
            raise TypeError(f"load_intervals: {name} is a {type(m).__name__}, the file holds a {cls}")
        install_intervals(m, vals, device=device)
        if mode is not None:
            m.mode = mode
    return blob.get("meta", {})
