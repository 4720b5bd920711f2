"""`EGNNDynamics` — drop-in for `oa_reactdiff.dynamics.EGNNDynamics(model=LEFTNet)` whose forward
pass runs on liboard_hip.so (hand-written gfx950 kernels) through the C ABI in include/oard.h.

Mirrors the reference interface (oa_reactdiff/dynamics/egnn_dynamics.py:14-168,
oa_reactdiff/dynamics/_base.py:10-132): same constructor arguments, same `forward` signature and
return convention `(List[Tensor], None)`, same attributes (`encoders`, `decoders`, `model`,
`pos_dim`, `node_nfs`, `fragment_names`, ...) and the same state-dict names and shapes, so a
checkpoint's `ddpm.dynamics.*` tensors load with `load_state_dict(strict=True)`.

There is no CPU / eager fallback: `forward` needs tensors on a ROCm device and the built library.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor, nn

from . import _capi
from .spec import model_dims, rbf_buffers, state_spec


class _Node(nn.Module):
    """Anonymous container used to reproduce the reference's module tree (names only)."""


def _xavier_names(name: str) -> bool:
    # layers the reference re-initialises with xavier_uniform_ (leftnet.py:235-242, 318-323, 558-564)
    keys = ("x_proj.0.weight", "x_proj.2.weight", "rbf_proj.weight", "vec_proj.weight", "xvec_proj.0.weight",
            "xvec_proj.2.weight", "vec1_proj.weight", "vec2_proj.weight", "update_net.0.weight", "update_net.2.weight")
    return name.endswith(keys)


class EGNNDynamics(nn.Module):
    def __init__(
        self,
        model_config: Dict,
        fragment_names: List[str],
        node_nfs: List[int],
        edge_nf: int,
        condition_nf: int = 0,
        pos_dim: int = 3,
        update_pocket_coords: bool = True,
        condition_time: bool = True,
        edge_cutoff: Optional[float] = None,
        model=None,
        device: torch.device = torch.device("cuda"),
        enforce_same_encoding: Optional[List] = None,
        source: Optional[Dict] = None,
    ) -> None:
        super().__init__()
        assert len(node_nfs) == len(fragment_names)                       # _base.py:44-46
        for nf in node_nfs:
            assert nf > pos_dim
        if "act_fn" not in model_config:                                  # _base.py:47-50 (mutates the dict)
            model_config["act_fn"] = "swish"
        if "in_node_nf" not in model_config:
            model_config["in_node_nf"] = model_config["in_hidden_channels"]
        if model is not None and getattr(model, "__name__", "LEFTNet") != "LEFTNet":
            raise NotImplementedError("the MI355X backend implements the LEFTNet denoiser only")
        if model_config["act_fn"] not in ("swish", "silu"):
            raise NotImplementedError("only the swish/silu activation is implemented")
        for key, want in (("legacy", True), ("update", True), ("pos_grad", False), ("single_layer_output", True),
                          ("object_aware", True), ("for_conf", False), ("ff", False)):       # reflect_equiv: both settings
            if model_config.get(key, want) != want:
                raise NotImplementedError(f"model_config[{key!r}] must be {want} for the MI355X backend")
        if "in_edge_nf" in model_config:
            raise NotImplementedError("edge attributes are not part of the LEFTNet denoising path")
        model_config.setdefault("in_hidden_channels", 8)
        self.model_config = model_config
        self.node_nfs = node_nfs
        self.edge_nf = edge_nf
        self.condition_nf = condition_nf
        self.fragment_names = fragment_names
        self.pos_dim = pos_dim
        self.update_pocket_coords = update_pocket_coords
        self.condition_time = condition_time
        self.edge_cutoff = edge_cutoff
        self.device = device
        self.dist_dim = 0
        self.embed_dim = model_config["in_node_nf"]                       # _base.py:69-77
        self.edge_embed_dim = 0
        if condition_time:
            self.embed_dim -= 1
        if condition_nf > 0:
            self.embed_dim -= condition_nf
        assert self.embed_dim > 0
        self.edge_encoder, self.edge_decoder = None, None

        H, R, L, Cc = model_dims(model_config)
        self._dims = (H, R, L, Cc)
        self._spec = state_spec(model_config, node_nfs, condition_nf, pos_dim, condition_time)
        self.model = _Node()
        self.encoders = nn.ModuleList([_Node() for _ in node_nfs])
        self.decoders = nn.ModuleList([_Node() for _ in node_nfs])
        means, betas = rbf_buffers(R, float(model_config.get("cutoff", 10.0)))
        for name, (shape, kind, fan_in) in self._spec.items():
            parts = name.split(".")
            mod: nn.Module = self
            for p in parts[:-1]:
                if p.isdigit() and isinstance(mod, nn.ModuleList):
                    mod = mod[int(p)]
                else:
                    if not hasattr(mod, p):
                        mod.add_module(p, _Node())
                    mod = getattr(mod, p)
            leaf = parts[-1]
            if kind == "buf_means":
                mod.register_buffer(leaf, means.clone().to(device))
            elif kind == "buf_betas":
                mod.register_buffer(leaf, betas.clone().to(device))
            else:
                t = torch.empty(shape, dtype=torch.float32, device=device)
                if kind == "ln_w":
                    nn.init.ones_(t)
                elif kind == "ln_b":
                    nn.init.zeros_(t)
                elif kind == "w" and _xavier_names(name):
                    nn.init.xavier_uniform_(t)
                elif kind == "b" and ".update_net." in name:
                    nn.init.zeros_(t)
                else:
                    bound = 1.0 / math.sqrt(fan_in)                      # nn.Linear default scale
                    nn.init.uniform_(t, -bound, bound)
                mod.register_parameter(leaf, nn.Parameter(t))
        if enforce_same_encoding is not None:                             # _base.py:110-113
            for ii in enforce_same_encoding:
                self.encoders[ii] = self.encoders[0]
                self.decoders[ii] = self.decoders[0]
        if source is not None:                                            # _base.py:65-66, 114-116
            self.model.load_state_dict(source["model"])
            self.encoders.load_state_dict(source["encoders"])
            self.decoders.load_state_dict(source["decoders"])

        #: "sync": reference behaviour (egnn_dynamics.py:138-143) — one host sync per call, NaN -> randn.
        #: "replace": the same replacement WITHOUT a host sync (oard_nan_replace reads the flag on the device; one randn draw per
        #: object and call whether or not it is used), flag kept like "async".
        #: "async": no host sync; the device-side flag of the last call is kept in `self.last_status` and OR-ed into
        #: the sticky flag `self.nan_seen` (reset it with `reset_nan_seen()`; the sampling loops read it once at the end).
        self.nan_check = "sync"
        #: Arithmetic of the two MFMA edge stages in INFERENCE calls.  "f32": the fp32 kernels.  "bf16x3": the split-precision
        #: kernels (csrc/oard_edge_b3.h: three bf16 terms per fp32 value, six bf16 MFMAs per K block, fp32 accumulation - fp32-grade
        #: results, ~1.3 x the step rate at B = 64).  None: the process default `default_edge_precision()` (fp32 unless the environment
        #: says OARD_GCL_B3=1 / OARD_EQUI_B3=1 - read per call, so that the answer never depends on what another module did).
        #: The choice travels with every library call (`oard_config.precision`) and is part of the packed-weights key: modules
        #: with different settings can run side by side, also from different threads / streams.
        self.edge_precision: Optional[str] = None
        #: The same choice for the TRAINING-mode forward (tape written; the backward kernels and weight-gradient GEMMs stay fp32).
        #: None: fp32 unless OARD_TRAIN_B3=1.
        self.train_edge_precision: Optional[str] = None
        self.last_status: Optional[Tensor] = None
        self.nan_seen: Optional[Tensor] = None
        self._packed: Optional[Tensor] = None
        self._packed_key = None
        self._packed_bwd: Optional[Tensor] = None
        self._packed_bwd_key = None
        self._topo_cache: "OrderedDict[tuple, _Topology]" = OrderedDict()
        self._train_topo_cache: "OrderedDict[tuple, object]" = OrderedDict()
        self._ws: Optional[Tensor] = None
        self._last_topo: Optional["_Topology"] = None
        #: "auto": the complete graph per sample runs the production kernels, any other edge list the general path (csrc/oard_general.h);
        #: "general": every inference call runs the general path (tests: two independent implementations of the same network)
        self.edge_list_path = "auto"
        self._warned_unbuilt = False

    # ------------------------------------------------------------------------------------------
    def _config(self) -> _capi.OardConfig:
        H, R, L, Cc = self._dims
        cfg = _capi.OardConfig()
        cfg.hidden, cfg.num_radial, cfg.num_layers, cfg.in_hidden = H, R, L, Cc
        cfg.n_obj = len(self.node_nfs)
        for k, nf in enumerate(self.node_nfs):
            cfg.node_nf[k] = nf
            cfg.enc_alias[k] = k      # aliasing is resolved through the parameter pointers we pass
        cfg.condition_nf = self.condition_nf
        cfg.condition_time = 1 if self.condition_time else 0
        cfg.pos_dim = self.pos_dim
        cfg.cutoff = float(self.model_config.get("cutoff", 10.0))
        cfg.reflect_equiv = 1 if self.model_config.get("reflect_equiv", True) else 0       # leftnet.py:268-272, 331, 794-796
        cfg.precision = self._precision_bits()
        return cfg

    def _precision_bits(self) -> int:
        """`oard_config.precision` (OARD_PREC_* of include/oard.h) from `edge_precision` / `train_edge_precision`; None resolves to the
        environment's default at THIS call - never to state left behind by another module."""
        for v in (self.edge_precision, self.train_edge_precision):
            if v not in (None, "f32", "bf16x3"):
                raise ValueError("edge_precision / train_edge_precision must be None, 'f32' or 'bf16x3'")
        env = lambda n: os.environ.get(n, "0") not in ("", "0")
        bits = 0
        if self.edge_precision == "bf16x3" or (self.edge_precision is None and env("OARD_GCL_B3")):
            bits |= _capi.PREC_GCL_BF16X3
        if self.edge_precision == "bf16x3" or (self.edge_precision is None and env("OARD_EQUI_B3")):
            bits |= _capi.PREC_EQUI_BF16X3
        if self.train_edge_precision == "bf16x3" or (self.train_edge_precision is None and env("OARD_TRAIN_B3")):
            bits |= _capi.PREC_TRAIN_BF16X3
        return bits

    def _ordered_tensors(self) -> List[Tensor]:
        """Tensors in the canonical order of include/oard.h (== state_spec order); encoder/decoder
        slots follow whatever module currently sits in `self.encoders[k]` / `self.decoders[k]`.
        The (owning module, attribute) pairs are resolved once per module tree (the walk is ~1 500 `nn.Module.__getattr__` calls,
        1.5 - 4 ms, and a training step asks five times); the tensors themselves are looked up afresh every call, so parameters that
        were re-assigned or re-loaded are seen."""
        tree = tuple(id(m) for m in self.modules())
        slots = self.__dict__.get("_slots")
        if slots is None or slots[0] != tree:
            pairs = []
            for name in self._spec:
                parts = name.split(".")
                mod: nn.Module = self
                for p in parts[:-1]:
                    mod = mod[int(p)] if (p.isdigit() and isinstance(mod, nn.ModuleList)) else getattr(mod, p)
                pairs.append((mod, parts[-1]))
            slots = self.__dict__["_slots"] = (tree, pairs)
        out: List[Tensor] = []
        for mod, attr in slots[1]:
            t = mod._parameters.get(attr)
            if t is None:
                t = mod._buffers.get(attr)
            out.append(t if t is not None else getattr(mod, attr))
        return out

    def _get_packed(self, cfg: _capi.OardConfig, stream: int) -> Tensor:
        tensors = self._ordered_tensors()
        L = _capi.lib()
        key = tuple((t.data_ptr(), t._version) for t in tensors) + (cfg.precision,)
        if self._packed is not None and key == self._packed_key:
            return self._packed
        dev = tensors[0].device
        for t in tensors:
            if t.device != dev or t.dtype != torch.float32 or not t.is_contiguous():
                raise _capi.OardError("parameters must be contiguous float32 tensors on one ROCm device")
        n = L.oard_param_count(C.byref(cfg))
        if n != len(tensors):
            raise _capi.OardError(f"parameter count mismatch: library expects {n}, module has {len(tensors)}")
        nbytes = L.oard_packed_bytes(C.byref(cfg))
        packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in tensors])
        _capi.check(L.oard_pack_weights(C.byref(cfg), ptrs, n, packed.data_ptr(), nbytes, stream), "oard_pack_weights")
        self._packed, self._packed_key = packed, key
        return packed

    def _get_packed_bwd(self, cfg: _capi.OardConfig, stream: int) -> Tensor:
        """Transposed weight streams of the backward edge kernels (repacked when a parameter changes)."""
        tensors = self._ordered_tensors()
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        if self._packed_bwd is not None and key == self._packed_bwd_key:
            return self._packed_bwd
        L = _capi.lib()
        n = L.oard_param_count(C.byref(cfg))
        nbytes = L.oard_packed_bwd_bytes(C.byref(cfg))
        packed = torch.empty(nbytes, dtype=torch.uint8, device=tensors[0].device)
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in tensors])
        _capi.check(L.oard_pack_weights_bwd(C.byref(cfg), ptrs, n, packed.data_ptr(), nbytes, stream), "oard_pack_weights_bwd")
        self._packed_bwd, self._packed_bwd_key = packed, key
        return packed

    def _param_names(self) -> List[str]:
        """Canonical (first) state-dict name of every distinct trainable parameter, in state-dict order."""
        seen, names = set(), []
        for name, t in zip(self._spec, self._ordered_tensors()):
            if isinstance(t, nn.Parameter) and id(t) not in seen:
                seen.add(id(t))
                names.append(name)
        return names

    def _param_dict(self) -> Dict[str, Tensor]:
        canon = set(self._param_names())
        return {name: t for name, t in zip(self._spec, self._ordered_tensors()) if name in canon}

    def _module_prefix(self, kind: str, k: int) -> str:
        """Canonical state-dict prefix of the module sitting at encoders[k] / decoders[k] (enforce_same_encoding aliases)."""
        mods = getattr(self, kind)
        for j in range(len(mods)):
            if mods[j] is mods[k]:
                return f"{kind}.{j}."
        return f"{kind}.{k}."

    @staticmethod
    def _c0row(P: Dict[str, Tensor], H: int, R: int) -> Tensor:
        """State of a masked (inter-object) edge: [lin3(0) x 2H | radial_lin(0) | 0 x R]  (leftnet.py:768-809 with dist = 0)."""
        m = "model."
        c0f = torch.nn.functional.linear(torch.nn.functional.silu(P[m + "radial_lin.0.bias"]), P[m + "radial_lin.2.weight"],
                                         P[m + "radial_lin.2.bias"])
        c0s = (torch.nn.functional.silu(P[m + "lin3.0.bias"]) @ P[m + "lin3.2.weight"].t() + P[m + "lin3.2.bias"]).reshape(())
        return torch.cat([c0s.expand(2 * H), c0f, torch.zeros(R, dtype=c0f.dtype, device=c0f.device)])

    def _get_topology(self, cfg, edge_index: Tensor, n_frag_switch: Tensor, combined_mask: Tensor,
                      stream: int, force_general: bool = False) -> "_Topology":
        force_general = force_general or self.edge_list_path == "general"
        key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape),
               n_frag_switch.data_ptr(), n_frag_switch._version, combined_mask.data_ptr(), combined_mask._version,
               combined_mask.numel(), force_general)
        topo = self._topo_cache.get(key)
        if topo is not None:
            self._topo_cache.move_to_end(key)
            return topo
        topo = _Topology(cfg, edge_index, n_frag_switch, combined_mask, stream, force_general=force_general)
        # the key is made of addresses and versions: the entry keeps the three tensors alive so that the caching
        # allocator cannot hand their storage to a different layout of the same size while the entry exists
        topo.key_tensors = (edge_index, n_frag_switch, combined_mask)
        self._topo_cache[key] = topo
        while len(self._topo_cache) > 8:
            self._topo_cache.popitem(last=False)
        return topo

    # ------------------------------------------------------------------------------------------
    def forward(
        self,
        xh: List[Tensor],
        edge_index: Tensor,
        t: Tensor,
        conditions: Tensor,
        n_frag_switch: Tensor,
        combined_mask: Tensor,
        edge_attr: Optional[Tensor] = None,
    ) -> Tuple[List[Tensor], Optional[Tensor]]:
        if not self.update_pocket_coords:
            raise NotImplementedError                                     # egnn_dynamics.py:125
        if edge_attr is not None:
            raise NotImplementedError("edge attributes are not part of the LEFTNet denoising path")
        train = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        dev = xh[0].device
        if dev.type != "cuda":
            raise _capi.OardError("EGNNDynamics.forward needs tensors on a ROCm device (no CPU fallback)")
        L = _capi.lib()
        cfg = self._config()
        # the production kernels are compiled per (hidden_channels, num_radial) pair (OARD_DIMS, oareactdiff_amd.build).  Any other pair
        # still runs - inference only - on the general-edge-list kernels, whose widths are run-time values (csrc/oard_general.h)
        built = L.oard_supported(C.byref(cfg)) == _capi.OARD_OK
        if not built:
            if train:
                _capi.check(L.oard_supported(C.byref(cfg)), "oard_supported (hidden_channels/num_radial not built: the backward pass exists for "
                                                            "built widths only - rebuild: OARD_DIMS=\"196x96,HxR\" python -m oareactdiff_amd.build)")
            if not self._warned_unbuilt:
                import warnings
                self._warned_unbuilt = True
                warnings.warn(f"hidden_channels={int(cfg.hidden)}, num_radial={int(cfg.num_radial)} is not a width pair the production kernels were built "
                              "for: this module's inference calls run the general-edge-list kernels (float64 accumulation, several times slower). "
                              f"For the production kernels rebuild: OARD_DIMS=\"196x96,{int(cfg.hidden)}x{int(cfg.num_radial)}\" python -m oareactdiff_amd.build")
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            packed = self._get_packed(cfg, stream) if built else None
            if train:
                return self._forward_train(cfg, packed, xh, edge_index, t, conditions, n_frag_switch, combined_mask, stream)
            topo = self._get_topology(cfg, edge_index, n_frag_switch, combined_mask, stream, force_general=not built)
            n_obj = len(self.node_nfs)
            xs = []
            for k in range(n_obj):
                x = xh[k]
                if x.dtype != torch.float32 or not x.is_contiguous():
                    x = x.contiguous().float()
                if x.shape != (topo.obj_counts[k], self.node_nfs[k]):
                    raise _capi.OardError(f"xh[{k}] has shape {tuple(x.shape)}, expected "
                                          f"{(topo.obj_counts[k], self.node_nfs[k])}")
                xs.append(x)
            outs = [torch.empty_like(x) for x in xs]
            tt, t_scalar = self._time_argument(t, dev, topo.max_sample_id)
            cond = None
            if self.condition_nf > 0:
                cond = conditions.detach().to(device=dev, dtype=torch.float32).contiguous()
                if cond.shape[0] <= topo.max_sample_id or cond.shape[1] != self.condition_nf:
                    raise _capi.OardError("conditions has the wrong shape")
            general = topo.graph is not None
            need = L.oard_graph_workspace_bytes(C.byref(cfg), topo.graph) if general else L.oard_workspace_bytes(C.byref(cfg), topo.handle)
            if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
                self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
            status = torch.zeros(2, dtype=torch.int32, device=dev)
            xp = (C.c_void_p * n_obj)(*[x.data_ptr() for x in xs])
            op = (C.c_void_p * n_obj)(*[o.data_ptr() for o in outs])
            if general:
                if self.nan_check == "replace":
                    raise _capi.OardError("nan_check='replace' is implemented for the complete-graph topology only")
                tensors = self._ordered_tensors()                 # the general path reads the raw parameters (no packed blob)
                for tns in tensors:
                    if tns is not None and (tns.dtype != torch.float32 or not tns.is_contiguous() or tns.device != dev):
                        raise _capi.OardError("the general-edge-list path needs contiguous float32 parameters on the call's device")
                pp = (C.c_void_p * len(tensors))(*[tns.data_ptr() if tns is not None else None for tns in tensors])
                rc = L.oard_graph_forward(C.byref(cfg), topo.graph, pp, len(tensors), xp, tt.data_ptr(), t_scalar,
                                          cond.data_ptr() if cond is not None else None, op, self._ws.data_ptr(), self._ws.numel(),
                                          status.data_ptr(), stream)
                _capi.check(rc, "oard_graph_forward")
            else:
                rc = L.oard_forward(C.byref(cfg), topo.handle, packed.data_ptr(), xp, tt.data_ptr(), t_scalar,
                                    cond.data_ptr() if cond is not None else None, op, self._ws.data_ptr(),
                                    self._ws.numel(), status.data_ptr(), stream)
                _capi.check(rc, "oard_forward")
            self._last_topo = topo
            self.last_status = status
            if self.nan_check != "sync":
                if self.nan_seen is None or self.nan_seen.device != dev:
                    self.nan_seen = torch.zeros(2, dtype=torch.int32, device=dev)
                self.nan_seen.bitwise_or_(status)         # device-side, no sync
            if self.nan_check == "replace":               # the reference's guard (:138-143) on the device: randn velocities iff the flag is set
                noise = [torch.randn(o.shape[0], self.pos_dim, device=dev) for o in outs]
                nz = (C.c_void_p * n_obj)(*[x.data_ptr() for x in noise])
                _capi.check(L.oard_nan_replace(C.byref(cfg), topo.handle, status.data_ptr(), nz, op, stream), "oard_nan_replace")
            if self.nan_check == "sync" and int(status[0].item()) != 0:   # egnn_dynamics.py:138-143
                print("Warning: detected nan in pos, resetting EGNN output to randn.")
                for k in range(n_obj):
                    v = torch.randn_like(outs[k][:, : self.pos_dim])
                    idx = topo.obj_masks[k]
                    if v.shape[0]:
                        mean = torch.zeros(int(idx.max()) + 1, self.pos_dim, device=dev).index_add_(0, idx, v)
                        cnt = torch.zeros(int(idx.max()) + 1, device=dev).index_add_(0, idx, torch.ones_like(idx, dtype=v.dtype))
                        v = v - (mean / cnt.clamp(min=1).unsqueeze(1))[idx]
                    outs[k][:, : self.pos_dim] = v
        return outs, None

    def reset_nan_seen(self) -> None:
        if self.nan_seen is not None:
            self.nan_seen.zero_()

    @staticmethod
    def _time_argument(t: Tensor, dev, max_sample_id: int) -> Tuple[Tensor, int]:
        """egnn_dynamics.py:106-114: a 1-D t is ONE value for every node (`t.item()`, which raises unless numel == 1);
        otherwise t is indexed by combined_mask, i.e. [B, 1] with a row per sample."""
        tt = t.detach().to(device=dev, dtype=torch.float32)
        if t.dim() == 1:
            if tt.numel() != 1:
                raise ValueError("a 1-D `t` must hold exactly one value (the reference calls t.item())")
            return tt.contiguous(), 1
        if t.dim() != 2 or tt.shape[1] != 1:
            raise _capi.OardError("t must be 1-D with one element or [B, 1]")
        if tt.shape[0] <= max_sample_id:
            raise _capi.OardError("t has fewer rows than samples")
        return tt.reshape(-1).contiguous(), 0

    def _get_train_topology(self, cfg, edge_index: Optional[Tensor], n_frag_switch: Tensor, combined_mask: Tensor, stream: int,
                            device=None):
        """One (single sub-batch) topology per layout.  `edge_index` given: verified against the implicit complete graph when the
        topology is built (one host read); `None` (DDPMTrainer's fused step, which never materialises an edge list): nothing to
        verify.  `combined_mask` / `n_frag_switch` may be host tensors (`device` = the GPU).  Cached by the identity of the tensors:
        a loader that reuses its batch tensors pays the table build once; a new layout costs ~1.5 ms of host time and no device
        synchronisation (training.TrainTopology)."""
        from . import training
        key = (0 if edge_index is None else edge_index.data_ptr(), 0 if edge_index is None else edge_index._version,
               () if edge_index is None else tuple(edge_index.shape),
               n_frag_switch.data_ptr(), n_frag_switch._version, combined_mask.data_ptr(), combined_mask._version,
               combined_mask.numel())
        topo = self._train_topo_cache.get(key)
        if topo is None:
            topo = training.TrainTopology(cfg, combined_mask, n_frag_switch, stream, edge_index=edge_index, device=device)
            topo.key_tensors = (edge_index, n_frag_switch, combined_mask)       # keep the addresses of the key alive
            self._train_topo_cache[key] = topo
            while len(self._train_topo_cache) > 4:
                self._train_topo_cache.popitem(last=False)
        else:
            self._train_topo_cache.move_to_end(key)
        return topo

    def _train_inputs(self, topo, xh: List[Tensor], t: Tensor, conditions: Tensor, dev):
        n_obj = len(self.node_nfs)
        xs = []
        for k in range(n_obj):
            x = xh[k].detach()
            if x.dtype != torch.float32 or not x.is_contiguous():
                x = x.contiguous().float()
            if x.shape != (topo.obj_counts[k], self.node_nfs[k]):
                raise _capi.OardError(f"xh[{k}] has shape {tuple(x.shape)}")
            xs.append(x)
        tt, t_scalar = self._time_argument(t, dev, topo.max_sample_id)
        cond = None
        if self.condition_nf > 0:
            cond = conditions.detach().to(device=dev, dtype=torch.float32).contiguous()
            if cond.shape[0] <= topo.max_sample_id or cond.shape[1] != self.condition_nf:
                raise _capi.OardError("conditions has the wrong shape")
        return xs, tt, t_scalar, cond

    def _run_forward_train(self, cfg, topo, packed: Tensor, xs: List[Tensor], tt: Tensor, t_scalar: int, cond: Optional[Tensor],
                           stream: int, reuse_tape: bool = False):
        """oard_forward_train on prepared inputs -> (outs, TrainState); no autograd involved (DDPMTrainer's fused step calls this
        directly and feeds the closed-form loss gradient to training.backward_sweep)."""
        from . import training
        L = _capi.lib()
        dev = xs[0].device
        n_obj = len(self.node_nfs)
        outs = [torch.empty_like(x) for x in xs]
        need = L.oard_workspace_bytes(C.byref(cfg), topo.handle)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        # one tape per call (it lives until that call's backward has run; the caching allocator recycles it).  reuse_tape (the fused
        # trainer: one forward, then its backward, on one stream): ONE grow-only buffer - batch layouts of changing size would otherwise
        # have the allocator carve multi-GB blocks of ever new sizes
        tape_bytes = L.oard_tape_bytes(C.byref(cfg), topo.handle)
        if reuse_tape:
            tb = getattr(self, "_tape_buf", None)
            if tb is None or tb.numel() < tape_bytes or tb.device != dev:
                self._tape_buf = None
                tb = self._tape_buf = torch.empty(tape_bytes + tape_bytes // 8, dtype=torch.uint8, device=dev)
            tape = tb[:tape_bytes]
        else:
            tape = torch.empty(tape_bytes, dtype=torch.uint8, device=dev)
        status = torch.zeros(2, dtype=torch.int32, device=dev)
        xp = (C.c_void_p * n_obj)(*[x.data_ptr() for x in xs])
        op = (C.c_void_p * n_obj)(*[o.data_ptr() for o in outs])
        rc = L.oard_forward_train(C.byref(cfg), topo.handle, packed.data_ptr(), xp, tt.data_ptr(), t_scalar,
                                  cond.data_ptr() if cond is not None else None, op, self._ws.data_ptr(), self._ws.numel(),
                                  tape.data_ptr(), tape.numel(), status.data_ptr(), stream)
        _capi.check(rc, "oard_forward_train")
        self.last_status = status
        if self.nan_check != "sync":
            if self.nan_seen is None or self.nan_seen.device != dev:
                self.nan_seen = torch.zeros(2, dtype=torch.int32, device=dev)
            self.nan_seen.bitwise_or_(status)         # device-side, no sync (DDPMTrainer reads it with the gradient norm)
        state = training.TrainState(cfg, topo, training.Tape(cfg, topo, tape), xs, tt, bool(t_scalar), cond)
        return outs, state

    def _forward_train(self, cfg, packed: Tensor, xh: List[Tensor], edge_index: Tensor, t: Tensor, conditions: Tensor,
                       n_frag_switch: Tensor, combined_mask: Tensor, stream: int):
        """Forward under autograd: training-mode HIP forward (tape) wrapped in `training.DynamicsFunction`."""
        from . import training
        dev = xh[0].device
        n_obj = len(self.node_nfs)
        topo = self._get_train_topology(cfg, edge_index, n_frag_switch, combined_mask, stream)
        xs, tt, t_scalar, cond = self._train_inputs(topo, xh, t, conditions, dev)

        def run_forward():
            return self._run_forward_train(cfg, topo, packed, xs, tt, t_scalar, cond, stream)

        names = self._param_names()
        P = self._param_dict()
        outs = training.DynamicsFunction.apply(self, run_forward, n_obj, *xs, *[P[n] for n in names])
        if self.nan_check == "sync" and int(self.last_status[0].item()) != 0:       # egnn_dynamics.py:138-143
            raise FloatingPointError("NaN in the predicted displacement during training (the reference would replace "
                                     "it with randn and carry on; a training step on that is meaningless)")
        return list(outs), None

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def debug_tap(self, which: int) -> Tensor:
        """Intermediate tensor of the last forward (reference node/edge order); tests only."""
        topo = self._last_topo
        assert topo is not None and self._ws is not None
        if topo.handle is None:
            raise _capi.OardError("stage taps exist on the complete-graph path only (the last call ran the general-edge-list path)")
        H, R, L_, Cc = self._dims
        cfg = self._config()
        shape = {
            _capi.TAP_S: (topo.n_nodes, H), _capi.TAP_VEC: (topo.n_nodes, 3 * H), _capi.TAP_EDGE: (topo.n_edges, 3 * H + R),
            _capi.TAP_POS_FRAME: (topo.n_nodes, 3), _capi.TAP_DPOS: (topo.n_nodes, 3), _capi.TAP_HOUT: (topo.n_nodes, Cc),
            _capi.TAP_LABELS: (topo.n_nodes, 1), _capi.TAP_NE1: (topo.n_nodes, 3 * H),
        }[which]
        dst = torch.empty(shape, dtype=torch.float32, device=self._ws.device)
        with torch.cuda.device(dst.device):
            stream = torch.cuda.current_stream(dst.device).cuda_stream
            _capi.check(_capi.lib().oard_tap(C.byref(cfg), topo.handle, self._ws.data_ptr(), which, 0, dst.data_ptr(), stream),
                        "oard_tap")
        return dst

    @torch.no_grad()
    def active_inner_edges(self) -> int:
        """Same-object edges that were inside the cutoff in the last inference call - the rows EquiMessage ran (the others carry an
        exactly-zero message, model/leftnet.py:748-753, and are skipped); -1 if that call built no list.  Synchronises: a measurement aid."""
        topo = self._last_topo
        assert topo is not None and self._ws is not None
        if topo.handle is None:
            return -1                                            # general-edge-list path: no active list
        cfg = self._config()
        out = C.c_int64(0)
        with torch.cuda.device(self._ws.device):
            stream = torch.cuda.current_stream(self._ws.device).cuda_stream
            _capi.check(_capi.lib().oard_active_inner_edges(C.byref(cfg), topo.handle, self._ws.data_ptr(), self._ws.numel(),
                                                            C.byref(out), stream), "oard_active_inner_edges")
        return int(out.value)

    @staticmethod
    def compute_frag_index(n_frag_switch: Tensor):
        """egnn_dynamics.py:177-182."""
        import numpy as np
        counts = [int((n_frag_switch == ii).sum()) for ii in torch.unique(n_frag_switch)]
        return np.concatenate([np.array([0]), np.cumsum(counts)])

    @staticmethod
    def remove_mean_batch(x: Tensor, indices: Tensor) -> Tensor:
        """egnn_dynamics.py:268-271."""
        n = int(indices.max()) + 1 if indices.numel() else 0
        s = torch.zeros(n, x.shape[1], dtype=x.dtype, device=x.device).index_add_(0, indices, x)
        c = torch.zeros(n, dtype=x.dtype, device=x.device).index_add_(0, indices, torch.ones_like(indices, dtype=x.dtype))
        return x - (s / c.clamp(min=1).unsqueeze(1))[indices]


class _Topology:
    """Device index tables for one (combined_mask, n_frag_switch, edge_index) triple."""

    def __init__(self, cfg, edge_index: Tensor, n_frag_switch: Tensor, combined_mask: Tensor, stream: int, force_general: bool = False):
        L = _capi.lib()
        cm = combined_mask.detach().to("cpu", torch.int64).contiguous()          # one-off host copy
        nfs = n_frag_switch.detach().to("cpu", torch.int64).contiguous()
        if cm.numel() != nfs.numel() or cm.numel() == 0:
            raise _capi.OardError("combined_mask / n_frag_switch size mismatch")
        self._lib = L
        self.handle, self.graph = None, None
        self.n_nodes = int(cm.numel())
        self.max_sample_id = int(cm.max())
        n_obj = cfg.n_obj
        self.obj_counts = [int((nfs == k).sum()) for k in range(n_obj)]
        dev = combined_mask.device
        starts = [0]
        for c in self.obj_counts:
            starts.append(starts[-1] + c)
        self.obj_masks = [combined_mask.detach()[starts[k]: starts[k + 1]].to(torch.int64) for k in range(n_obj)]
        ei = edge_index.detach()
        if ei.dim() != 2 or ei.shape[0] != 2 or ei.dtype != torch.int64 or ei.device != dev:
            raise _capi.OardError("edge_index must be an int64 [2, E] tensor on the same device")
        ei = ei.contiguous()
        # the production kernels assume the complete-per-sample graph (any ordering of its edge list: outputs are per node): verify once.
        # A layout their tables do not cover (a (sample, object) group of more than 1024 atoms) cannot be that path's either way.
        h = C.c_void_p()
        rc = L.oard_topology_create(C.byref(cfg), C.cast(cm.data_ptr(), C.POINTER(C.c_int64)),
                                    C.cast(nfs.data_ptr(), C.POINTER(C.c_int64)), cm.numel(), C.byref(h))
        complete = False
        if rc == _capi.OARD_OK:
            ok = torch.zeros(1, dtype=torch.int32, device=dev)
            _capi.check(L.oard_topology_check_edge_index(h, ei.data_ptr(), ei.shape[1], ok.data_ptr(), stream),
                        "oard_topology_check_edge_index")
            complete = int(ok.item()) == 1
            if complete and not force_general:
                self.handle = h
                self.n_edges = int(L.oard_topology_num_edges(h))
                self.n_inner = int(L.oard_topology_num_inner_edges(h))
                self.n_samples = int(L.oard_topology_num_samples(h))
                return
            L.oard_topology_destroy(h)
        # not the complete graph per sample (edge_cutoff graphs, disconnected components, arbitrary lists: egnn_dynamics.py:63-72
        # accepts them all): the general-edge-list path (csrc/oard_general.h) takes the call.  One host copy of the edge list per
        # topology, like the two masks above.
        ei_host = ei.to("cpu").contiguous()
        g = C.c_void_p()
        rc = L.oard_graph_create(C.byref(cfg), cm.data_ptr(), nfs.data_ptr(), cm.numel(), ei_host.data_ptr(), ei_host.shape[1], C.byref(g))
        _capi.check(rc, "oard_graph_create (node ids out of range, or n_frag_switch not in ascending object blocks)")
        self.graph = g
        self.n_edges = int(L.oard_graph_num_edges(g))
        self.n_inner = int((nfs[ei_host[0]] == nfs[ei_host[1]]).sum()) if ei_host.shape[1] else 0
        self.n_samples = int(torch.unique(cm).numel())
        if complete is False and rc == _capi.OARD_OK and not force_general and L.oard_graph_is_complete(g):
            import warnings
            warnings.warn("this batch layout is outside the production kernels' tables (a (sample, object) group of more than 1024 atoms): "
                          "the complete graph is served by the general-edge-list path, which is built for parity, not throughput")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self._lib.oard_topology_destroy(self.handle)
            if getattr(self, "graph", None):
                self._lib.oard_graph_destroy(self.graph)
        except Exception:
            pass
