"""Checkpoint adapter: load the hot-path weights of a reference
Lightning checkpoint (`DDPMModule`, oa_reactdiff/trainer/pl_trainer.py:55-147; e.g.
`pretrained-ts1x-diff.ckpt`) into the MI355X `EGNNDynamics`.

The Lightning `state_dict` stores the dynamics under the prefix `ddpm.dynamics.`; the constructor
arguments are in `hyper_parameters` (`save_hyperparameters()`, pl_trainer.py:147)."""
from __future__ import annotations

import builtins
import collections
import io
import pickle
import types
from typing import Any, Dict, Mapping, Optional, Tuple

import torch

from .dynamics import EGNNDynamics

PREFIXES = ("ddpm.dynamics.", "dynamics.", "")


def extract_dynamics_state(state_dict: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Tensors of the dynamics module, prefix stripped (`model.*`, `encoders.*`, `decoders.*`)."""
    for pre in PREFIXES:
        sub = {k[len(pre):]: v for k, v in state_dict.items()
               if k.startswith(pre) and k[len(pre):].split(".")[0] in ("model", "encoders", "decoders")}
        if sub:
            return sub
    raise KeyError("no dynamics tensors (model.* / encoders.* / decoders.*) found in the state dict")


# ---------------------------------------------------------------------------------------------------------------------
# Reading the checkpoint FILE.  `DDPMModule.save_hyperparameters()` (pl_trainer.py:147) pickles the constructor
# arguments, among them the class object `model=LEFTNet` (pl_trainer.py:77; demo.py:269 restores it through
# `load_from_checkpoint`), and Lightning adds optimiser / callback / loop state that may reference its own classes.
# `torch.load(path)` therefore needs `oa_reactdiff` (and Lightning) importable, or it raises.  `load_checkpoint` reads
# the same file with a restricted unpickler: tensors, storages and plain containers are rebuilt, every other global
# becomes an inert placeholder that keeps its qualified name (so `hyper_parameters["model"].__name__ == "LEFTNet"`)
# and swallows whatever state the pickle feeds it.  No code from the pickle stream is ever imported or executed.
# ---------------------------------------------------------------------------------------------------------------------
class Placeholder:
    """Stands in for an object whose class is not importable here.  Accepts any construction / state / item protocol the
    pickle stream uses on it and keeps the data for inspection (`_args`, `_state`, `_items`)."""

    def __new__(cls, *args, **kwargs):                 # NEWOBJ / REDUCE both come through here (NEWOBJ skips __init__)
        self = object.__new__(cls)
        self._args, self._kwargs, self._state, self._items = args, kwargs, None, []
        return self

    def __init__(self, *args, **kwargs):
        pass

    def __setstate__(self, state):
        self._state = state

    def __setitem__(self, key, value):
        self._items.append((key, value))

    def append(self, value):
        self._items.append(value)

    def extend(self, values):
        self._items.extend(values)

    def add(self, value):
        self._items.append(value)

    def __repr__(self):
        c = type(self)
        return f"<placeholder {c.__module__}.{c.__qualname__}>"


def _placeholder_class(module: str, name: str) -> type:
    return type(name.rsplit(".", 1)[-1], (Placeholder,), {"__module__": module, "__qualname__": name})


# `getattr` and `object` are deliberately absent: `getattr` turns any allowed global into a gadget chain
# (`getattr(torch.Tensor, "__reduce_ex__").__globals__["__builtins__"]["eval"]`), `object` offers `__reduce_ex__` / `__subclasses__`.
_SAFE_BUILTINS = {"set", "frozenset", "slice", "complex", "range", "bytearray", "bytes", "list", "dict", "tuple", "int",
                  "float", "bool", "str"}
_SAFE_GLOBALS = {
    ("collections", "OrderedDict"): collections.OrderedDict,
    ("collections", "defaultdict"): collections.defaultdict,
    ("collections", "deque"): collections.deque,
    ("torch", "Size"): torch.Size,
    ("torch", "device"): torch.device,
    ("torch", "Tensor"): torch.Tensor,
    ("torch.nn.parameter", "Parameter"): torch.nn.Parameter,
}
for _n in ("_rebuild_tensor_v2", "_rebuild_tensor", "_rebuild_parameter", "_rebuild_parameter_with_state",
           "_rebuild_device_tensor_from_numpy", "_rebuild_qtensor"):
    if hasattr(torch._utils, _n) and _n != "_rebuild_qtensor":
        _SAFE_GLOBALS[("torch._utils", _n)] = getattr(torch._utils, _n)
for _n in ("float16", "float32", "float64", "bfloat16", "int8", "uint8", "int16", "int32", "int64", "bool"):
    _SAFE_GLOBALS[("torch", _n)] = getattr(torch, _n)
for _n in ("FloatStorage", "DoubleStorage", "HalfStorage", "BFloat16Storage", "LongStorage", "IntStorage", "ShortStorage",
           "CharStorage", "ByteStorage", "BoolStorage", "UntypedStorage"):
    if hasattr(torch, _n):
        _SAFE_GLOBALS[("torch", _n)] = getattr(torch, _n)
try:                                                    # numpy scalars / arrays (Lightning stores e.g. best_model_score)
    import numpy as _np
    _core = _np._core if hasattr(_np, "_core") else _np.core
    for _mod in ("numpy.core.multiarray", "numpy._core.multiarray"):
        _SAFE_GLOBALS[(_mod, "scalar")] = _core.multiarray.scalar
        _SAFE_GLOBALS[(_mod, "_reconstruct")] = _core.multiarray._reconstruct
    _SAFE_GLOBALS[("numpy", "dtype")] = _np.dtype
    _SAFE_GLOBALS[("numpy", "ndarray")] = _np.ndarray
except Exception:                                       # pragma: no cover
    pass


def _allowed_callable(f: Any) -> bool:
    """What REDUCE / NEWOBJ / INST / OBJ may call: an allow-listed global or a placeholder class - nothing that was COMPUTED by the
    stream (no bound methods, no attributes of allowed objects)."""
    if isinstance(f, type) and issubclass(f, Placeholder):
        return True
    try:
        if any(f is v for v in _SAFE_GLOBALS.values()):
            return True
    except Exception:                                   # pragma: no cover
        return False
    return isinstance(f, type) and f.__module__ == "builtins" and f.__name__ in _SAFE_BUILTINS


class _RestrictedUnpickler(pickle._Unpickler):
    """The pure-Python unpickler (its opcode handlers can be overridden; the pickle of a checkpoint is small - tensor data
    travel as separate zip records).  Three rules: (1) globals resolve through the allow-list or become placeholders,
    (2) only allow-listed callables and placeholder classes are ever CALLED, (3) BUILD never touches a class object, a
    function or a module, and never sets a dunder attribute on anything but a placeholder."""
    dispatch = dict(pickle._Unpickler.dispatch)

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.replaced = set()                           #: qualified names replaced by placeholders in this load (diagnostics)

    def find_class(self, module: str, name: str):
        if (module, name) in _SAFE_GLOBALS:
            return _SAFE_GLOBALS[(module, name)]
        if module == "builtins" and name in _SAFE_BUILTINS:
            return getattr(builtins, name)
        self.replaced.add(f"{module}.{name}")
        return _placeholder_class(module, name)

    @staticmethod
    def _check(f):
        if not _allowed_callable(f):
            raise pickle.UnpicklingError(f"checkpoint pickle tries to call {f!r}: not on the allow-list")

    def load_reduce(self):
        self._check(self.stack[-2])
        pickle._Unpickler.load_reduce(self)
    dispatch[pickle.REDUCE[0]] = load_reduce

    def load_newobj(self):
        self._check(self.stack[-2])
        pickle._Unpickler.load_newobj(self)
    dispatch[pickle.NEWOBJ[0]] = load_newobj

    def load_newobj_ex(self):
        self._check(self.stack[-3])
        pickle._Unpickler.load_newobj_ex(self)
    dispatch[pickle.NEWOBJ_EX[0]] = load_newobj_ex

    def _instantiate(self, klass, args):               # INST / OBJ
        self._check(klass)
        pickle._Unpickler._instantiate(self, klass, args)

    def load_build(self):
        inst, state = self.stack[-2], self.stack[-1]
        if isinstance(inst, (type, types.ModuleType, types.FunctionType, types.BuiltinFunctionType)):
            raise pickle.UnpicklingError(f"checkpoint pickle applies BUILD to {inst!r}")
        if not isinstance(inst, Placeholder):
            keys = []
            for part in (state if isinstance(state, tuple) and len(state) == 2 else (state,)):
                if isinstance(part, Mapping):
                    keys += list(part.keys())
            if any(isinstance(k, str) and k.startswith("__") for k in keys):
                raise pickle.UnpicklingError("checkpoint pickle sets a dunder attribute through BUILD")
        pickle._Unpickler.load_build(self)
    dispatch[pickle.BUILD[0]] = load_build


#: what `torch.load(..., pickle_module=...)` expects: a module-like object with `Unpickler`, `load`, `loads`
restricted_pickle = types.ModuleType("oareactdiff_amd.checkpoint.restricted_pickle")
restricted_pickle.Unpickler = _RestrictedUnpickler
restricted_pickle.load = lambda f, **kw: _RestrictedUnpickler(f, **kw).load()
restricted_pickle.loads = lambda b, **kw: _RestrictedUnpickler(io.BytesIO(b), **kw).load()
restricted_pickle.__name__ = "pickle"


def read_checkpoint(path: str) -> Dict[str, Any]:
    """The checkpoint file as a dict (tensors on the CPU) WITHOUT importing anything the pickle names: works on a box
    that has neither `oa_reactdiff` nor `pytorch_lightning`.  Replaces `torch.load(path)` / `load_from_checkpoint`
    (demo.py:269) for a file written by Lightning from `DDPMModule` (pl_trainer.py:55-147)."""
    ckpt = torch.load(path, map_location="cpu", pickle_module=restricted_pickle, weights_only=False)
    if not isinstance(ckpt, Mapping) or "state_dict" not in ckpt:
        raise KeyError(f"{path}: not a Lightning checkpoint (no `state_dict` entry)")
    return ckpt


def _plain(x: Any) -> Any:
    """hyper_parameters as plain containers: placeholder CLASSES stay (their name is information), placeholder INSTANCES
    of dict-like classes (Lightning's `AttributeDict`) become dicts."""
    if isinstance(x, Placeholder):
        if x._items and all(isinstance(i, tuple) and len(i) == 2 for i in x._items):
            return {k: _plain(v) for k, v in x._items}
        if isinstance(x._state, Mapping):
            return {k: _plain(v) for k, v in x._state.items()}
        return x
    if isinstance(x, Mapping):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_plain(v) for v in x)
    return x


def load_checkpoint(path: str, device: torch.device = torch.device("cuda"),
                    overrides: Optional[dict] = None) -> Tuple[EGNNDynamics, dict]:
    """`DDPMModule.load_from_checkpoint(path).ddpm.dynamics` (demo.py:269) for the MI355X backend: reads the file with the
    restricted unpickler and builds `EGNNDynamics` with the checkpoint's weights (strict).  Returns (dynamics, hyper_parameters)."""
    ckpt = dict(read_checkpoint(path))
    ckpt["hyper_parameters"] = _plain(ckpt.get("hyper_parameters", {}))
    return dynamics_from_checkpoint(ckpt, device=device, overrides=overrides)


def dynamics_from_checkpoint(ckpt: Mapping, device: torch.device = torch.device("cuda"),
                             overrides: Optional[dict] = None) -> Tuple[EGNNDynamics, dict]:
    """Builds `EGNNDynamics` from an already loaded Lightning checkpoint dict (`read_checkpoint(path)`, or `torch.load` where
    the reference package is importable) and loads its weights with strict=True.  Returns (dynamics, hyper_parameters)."""
    hp = dict(ckpt.get("hyper_parameters", {}))
    if overrides:
        hp.update(overrides)
    try:
        model_config, node_nfs = dict(hp["model_config"]), list(hp["node_nfs"])
    except KeyError as e:
        raise KeyError(f"checkpoint hyper_parameters lack {e}; pass them through `overrides`") from None
    model = hp.get("model")                                # the class object Lightning pickled (pl_trainer.py:77), or its placeholder
    if model is not None and getattr(model, "__name__", "LEFTNet") != "LEFTNet":
        raise NotImplementedError(f"checkpoint was trained with model={getattr(model, '__name__', model)!r}; "
                                  "the MI355X backend implements LEFTNet only")
    dyn = EGNNDynamics(
        model_config=model_config,
        fragment_names=list(hp.get("fragment_names", [f"frag{k}" for k in range(len(node_nfs))])),
        node_nfs=node_nfs, edge_nf=int(hp.get("edge_nf", 0)), condition_nf=int(hp.get("condition_nf", 0)),
        pos_dim=int(hp.get("pos_dim", 3)), update_pocket_coords=bool(hp.get("update_pocket_coords", True)),
        condition_time=bool(hp.get("condition_time", True)), edge_cutoff=hp.get("edge_cutoff"),
        device=device, enforce_same_encoding=hp.get("enforce_same_encoding"),
    )
    dyn.load_state_dict(extract_dynamics_state(ckpt["state_dict"]), strict=True)
    return dyn, hp
