"""One training step of the diffusion model around the HIP denoiser (BASELINE config 4).

`DDPMTrainer.training_step` mirrors `DDPMModule.training_step` + `configure_optimizers` + `configure_gradient_clipping`
(oa_reactdiff/trainer/pl_trainer.py:327-347, :150-160, :391-418) and the data-parallel strategy of
oa_reactdiff/trainer/train_ts1x.py:197-203 (Lightning `DDPStrategy`):

    nll, info = compute_loss(batch);  loss = nll.mean(0)            pl_trainer.py:328-329
    loss.backward()                                                  HIP backward (oareactdiff_amd/training.py)
    gradients averaged over the ranks                                DDP: here ONE all-reduce of one flat bucket
    adaptive gradient clipping (queue of recent norms)               pl_trainer.py:391-418
    AdamW(amsgrad=True) step                                         pl_trainer.py:150, train_ts1x.py:66-71

Multi-GPU: one process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm), every rank holds a
replica and its own shard of the batch.  All trainable gradients live in ONE contiguous fp32 buffer (`flat_grad`,
42.6 MB for the production model): `param.grad` are views into it, so the backward pass accumulates in place and the
step needs exactly one `all_reduce` — sized for xGMI's per-link bound rather than bucketed for overlap (the backward
of this model takes tens of ms, the all-reduce of 42.6 MB over 7 links well under 1 ms).  The two modules the forward
never uses (`model.distance_embedding`, `model.last_layer`; Lightning needs `find_unused_parameters=True` for them) are
left out of the bucket and never get a gradient, as in the reference.
"""
from __future__ import annotations

import copy
import contextlib
import math
import os
from collections.abc import Mapping
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist
from torch import Tensor, nn

from .loss import DiffusionLoss

UNUSED_PREFIXES = ("model.distance_embedding.", "model.last_layer.")
DEFAULT_OPTIMIZER = dict(lr=2.5e-4, betas=(0.9, 0.999), weight_decay=0, amsgrad=True)      # train_ts1x.py:66-71


class Queue:
    """oa_reactdiff/utils/training_tools.py:6-23."""

    def __init__(self, max_len: int = 50):
        self.items: List[float] = []
        self.max_len = max_len

    def __len__(self):
        return len(self.items)

    def add(self, item: float) -> None:
        self.items.insert(0, item)
        if len(self) > self.max_len:
            self.items.pop()

    def mean(self) -> float:
        return float(np.mean(self.items))

    def std(self) -> float:
        return float(np.std(self.items))


class LazyInfo(Mapping):
    """What `training_step` returns with `host_sync=False`: the same keys as the eager dict (loss, skipped, grad_norm, max_grad_norm,
    error_t_k, unorm_error_t_k), held as two small device tensors and copied to the host the first time any of them is read."""

    def __init__(self, stats: Tensor, out4: Tensor, K: int, scales: List[float], clip: bool):
        self._dev = (stats, out4)
        self._K, self._scales, self._clip = K, scales, clip
        self._d: Optional[Dict[str, float]] = None

    def _get(self) -> Dict[str, float]:
        if self._d is None:
            stats, out4 = (t.tolist() for t in self._dev)
            K, d = self._K, {}
            for k in range(K):
                d[f"error_t_{k}"] = stats[3 + k] / (self._scales[k] + 1e-4)
                d[f"unorm_error_t_{k}"] = stats[3 + K + k]
            if self._clip:
                d["grad_norm"], d["max_grad_norm"] = out4[0], out4[1]
            d["loss"] = stats[2]
            d["skipped"] = int(out4[3] != 0)
            self._d, self._dev = d, None
        return self._d

    def __getitem__(self, key):
        return self._get()[key]

    def __iter__(self):
        return iter(self._get())

    def __len__(self):
        return len(self._get())


class DDPMTrainer:
    def __init__(self, dynamics: nn.Module, noise_schedule: str = "polynomial_2", timesteps: int = 1000,
                 precision: float = 1e-5, norm_values: Sequence[float] = (1.0, 1.0, 1.0),
                 norm_biases: Sequence[float] = (0.0, 0.0, 0.0), loss_type: str = "l2", pos_only: bool = False,
                 scales: Sequence[float] = (1.0, 1.0, 1.0), fixed_idx: Optional[List[int]] = None,
                 optimizer_config: Optional[Dict] = None, clip_grad: bool = True,
                 process_group: Optional["dist.ProcessGroup"] = None, fused: Optional[bool] = None, host_sync: bool = True,
                 microbatches: int = 1):
        """`microbatches=2` (fused steps on batches that carry their host copies, `to_device`): the step's reactions are dealt to two
        halves that run their forward and their reverse sweep CONCURRENTLY on two streams (the caller's and the library's idle
        sub-batch stream), each with its own tape / workspace / scratch and its own gradient buffer, summed before the all-reduce.
        Same loss, same gradient up to the order of one addition; the low-occupancy node-side launches of one half then run beside
        the other half's edge kernels instead of alone on the chip (round 6).

        `host_sync=False` (fused steps only): the adaptive-clipping decision, the skip decision and AdamW's step-dependent scalars
        are evaluated on the device (`oard_adamw_step_dev`), so a training step contains NO device -> host read; `training_step`
        then returns a `LazyInfo` whose values are fetched when they are first looked at.  Arithmetic and state are the same as with
        `host_sync=True` (the clipping history is summed in numpy's order); the clipping / skip messages are not printed."""
        self.dynamics = dynamics
        self.host_sync = bool(host_sync)
        if int(microbatches) not in (1, 2):
            raise ValueError("microbatches must be 1 or 2")
        self.microbatches = int(microbatches)
        self._mb = None                                # second micro-batch: stream, gradient buffer, the module's per-call buffers
        self._clip_state = None                        # device mirror of (history, opt_step, skipped_steps) while host_sync is off
        self.loss = DiffusionLoss(dynamics, noise_schedule, timesteps, precision, norm_values=norm_values,
                                  norm_biases=norm_biases, pos_only=pos_only, fixed_idx=fixed_idx, loss_type=loss_type,
                                  scales=scales)
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.time_collectives = False                      # True: all_reduce_gradients brackets the collective with timing events (collective_report)
        self._collective_events = []
        # the collectives run with more than one rank - or, for a smoke test of the RCCL path on a one-GPU box, whenever a process
        # group exists and OARD_FORCE_COLLECTIVES is set (a one-rank broadcast / all-reduce is the identity)
        self.collectives = self.world > 1 or (dist.is_available() and dist.is_initialized()
                                              and bool(os.environ.get("OARD_FORCE_COLLECTIVES")))
        # distinct trainable parameters the forward uses, in state-dict order (shared encoders count once)
        seen, self.params, self.names = set(), [], []
        for name, p in dynamics.named_parameters():
            if p.requires_grad and id(p) not in seen and not name.startswith(UNUSED_PREFIXES):
                seen.add(id(p))
                self.params.append(p)
                self.names.append(name)
        n = sum(p.numel() for p in self.params)
        # one bucket: [gradients (n) | non-finite flag (1)]; the flag rides in the same all-reduce so that every rank takes the
        # same skip / step decision after the collective (a rank that bails out BEFORE it would leave the others hanging)
        self._bucket = torch.zeros(n + 1, dtype=self.params[0].dtype, device=self.params[0].device)
        self.flat_grad = self._bucket[:n]
        self.skipped_steps = 0
        if self.collectives:
            self.sync_replicas()
        off = 0
        for p in self.params:                          # p.grad = view into the bucket: backward accumulates in place
            p.grad = self.flat_grad[off: off + p.numel()].view_as(p)
            off += p.numel()
        self.opt_config = dict(DEFAULT_OPTIMIZER, **(optimizer_config or {}))
        can_fuse = (hasattr(dynamics, "_get_packed_bwd") and self.params[0].device.type == "cuda" and loss_type == "l2"
                    and self.params[0].dtype == torch.float32)
        if fused and not can_fuse:                     # the fused kernels implement the l2 objective in float32 on the HIP module only
            raise ValueError("fused=True needs the HIP EGNNDynamics on a ROCm device, float32 parameters and loss_type='l2' "
                             f"(got loss_type={loss_type!r}, dtype={self.params[0].dtype}, device={self.params[0].device})")
        self.fused = can_fuse if fused is None else bool(fused)
        if not self.host_sync and not self.fused:
            raise ValueError("host_sync=False is a mode of the fused step (HIP EGNNDynamics on a ROCm device, l2 loss)")
        if self.fused:
            # the HIP module: (1) its backward accumulates straight into the .grad views of the bucket; (2) the parameters
            # themselves become views of ONE flat buffer, so that AdamW is a single kernel over it (oard_adamw_step) and
            # state_dict() / load_state_dict() keep working on the same storage
            dynamics.grad_inplace = True
            self.flat_param = torch.empty(n, dtype=torch.float32, device=self.flat_grad.device)
            off = 0
            with torch.no_grad():
                for p in self.params:
                    self.flat_param[off: off + p.numel()].copy_(p.reshape(-1))
                    p.data = self.flat_param[off: off + p.numel()].view_as(p)
                    off += p.numel()
            self.exp_avg = torch.zeros_like(self.flat_param)
            self.exp_avg_sq = torch.zeros_like(self.flat_param)
            self.max_exp_avg_sq = torch.zeros_like(self.flat_param)
            self.opt_step = 0
        # `trainer.optimizer` is what the reference's configure_optimizers returns (pl_trainer.py:149-151): the generic path steps it;
        # the fused path runs oard_adamw_step on the flat buffers but reads lr / betas / eps / weight_decay / amsgrad from
        # `optimizer.param_groups[0]` at EVERY step, so an LR scheduler attached to it (or an edit of the group) takes effect in
        # both modes.  The fused moments live in `exp_avg` / `exp_avg_sq` / `max_exp_avg_sq` / `opt_step`: see state_dict().
        self.optimizer = torch.optim.AdamW(self.params, **self.opt_config)
        self._gamma_dev = None
        self.clip_grad = clip_grad
        if clip_grad:                                  # pl_trainer.py:143-146
            self.gradnorm_queue = Queue()
            self.gradnorm_queue.add(3000)

    # ---- resume: what Lightning's checkpoint keeps of the reference trainer (`optimizer_states`, pl_trainer.py:149-151; the clipping
    # history `gradnorm_queue` is NOT checkpointed by the reference and restarts at [3000], :143-146 - it is kept here) -------------
    def state_dict(self) -> Dict:
        """Everything a bit-identical continuation of the training needs besides the module's own state_dict(): the optimiser state
        (fused: flat AdamW moments + step counter + the hyper-parameters of `optimizer.param_groups`; generic: torch's), the clipping
        history and the skipped-step counter."""
        self._pull_clip_state()
        sd: Dict = {"fused": bool(self.fused), "names": list(self.names), "skipped_steps": int(self.skipped_steps),
                    "gradnorm_queue": list(self.gradnorm_queue.items) if self.clip_grad else None}
        if self.fused:
            sd["opt_step"] = int(self.opt_step)
            sd["param_groups"] = [{k: v for k, v in g.items() if k != "params"} for g in self.optimizer.param_groups]
            for k in ("exp_avg", "exp_avg_sq", "max_exp_avg_sq"):
                sd[k] = getattr(self, k).detach().clone()
        else:
            sd["optimizer"] = copy.deepcopy(self.optimizer.state_dict())     # a snapshot: torch hands out the live moment tensors
        return sd

    def load_state_dict(self, sd: Dict) -> None:
        if list(sd["names"]) != list(self.names):
            raise ValueError("trainer state belongs to a different parameter list")
        if bool(sd["fused"]) != bool(self.fused):
            raise ValueError(f"trainer state was saved with fused={sd['fused']}, this trainer runs fused={self.fused}")
        self._clip_state = None                        # rebuilt from the host copies at the next sync-free step
        self.skipped_steps = int(sd["skipped_steps"])
        if self.clip_grad and sd.get("gradnorm_queue") is not None:
            self.gradnorm_queue.items = [float(x) for x in sd["gradnorm_queue"]]
        if self.fused:
            self.opt_step = int(sd["opt_step"])
            for g, saved in zip(self.optimizer.param_groups, sd["param_groups"]):
                g.update(saved)
            with torch.no_grad():
                for k in ("exp_avg", "exp_avg_sq", "max_exp_avg_sq"):
                    getattr(self, k).copy_(sd[k].to(self.flat_param.device))
        else:
            self.optimizer.load_state_dict(sd["optimizer"])

    # ---- host_sync=False: history / step counter / skip counter live on the device between reads ------------------------------------------
    CLIP_CAPACITY = 50                                 # utils/training_tools.py:9 (Queue(max_len=50))

    def _push_clip_state(self) -> Tensor:
        """Device copy of the host-side clipping history, optimiser step count and skip count (layout: include/oard.h, oard_adamw_step_dev)."""
        cap = self.CLIP_CAPACITY
        items = list(self.gradnorm_queue.items) if self.clip_grad else []
        host = torch.zeros(4 + cap + 8, dtype=torch.float64)
        host[0], host[1], host[2] = len(items), self.opt_step, self.skipped_steps
        if items:
            host[4: 4 + len(items)] = torch.tensor(items, dtype=torch.float64)
        self._clip_state = host.to(self.flat_grad.device)
        return self._clip_state

    def _pull_clip_state(self) -> None:
        """Refresh `gradnorm_queue`, `opt_step`, `skipped_steps` from the device (one host sync; no-op while host_sync is on)."""
        if self._clip_state is None:
            return
        host = self._clip_state.cpu()
        n = int(host[0])
        self.opt_step, self.skipped_steps = int(host[1]), int(host[2])
        if self.clip_grad:
            self.gradnorm_queue.items = [float(x) for x in host[4: 4 + n].tolist()]

    @staticmethod
    def to_device(batch, device, non_blocking: bool = True):
        """A host batch (the dataset's collate output: per object {size, pos, one_hot, charge, mask}, conditions) moved to `device`
        with the HOST copies of `mask` / `size` kept beside the device tensors (`mask_host`, `size_host`): the fused step builds the
        batch layout from them, so a never-seen layout - every step of a real training run - costs no device -> host copy.  Batches
        without the host copies work as before (one blocking copy of the masks per new layout)."""
        reps, cond = batch
        out = []
        for r in reps:
            d = {k: (v.to(device, non_blocking=non_blocking) if isinstance(v, Tensor) else v) for k, v in r.items()}
            d["mask_host"], d["size_host"] = r["mask"].detach().cpu(), r["size"].detach().cpu()
            out.append(d)
        return out, (cond.to(device, non_blocking=non_blocking) if isinstance(cond, Tensor) else cond)

    # pl_trainer.py:208-282
    def compute_loss(self, batch, training: bool = True, **kw) -> Tuple[Tensor, Dict[str, float]]:
        representations, conditions = batch
        return self.loss.compute_loss(representations, conditions, training=training, **kw)

    def sync_replicas(self, check: bool = True) -> None:
        """What torch DDP does when it wraps a module (Lightning `DDPStrategy`, train_ts1x.py:197-203): rank 0's parameters
        and buffers are broadcast to every rank, so replicas built from different RNG states - or with a checkpoint loaded
        on rank 0 only - start identical.  One flat broadcast for the parameters, one for the floating-point buffers."""
        if not self.collectives:
            return
        with torch.no_grad():
            for tensors in (list(self.dynamics.parameters()),
                            [b for b in self.dynamics.buffers() if b.is_floating_point()]):
                uniq, seen = [], set()
                for t in tensors:                          # shared encoders appear once
                    if id(t) not in seen:
                        seen.add(id(t))
                        uniq.append(t)
                if not uniq:
                    continue
                flat = torch.cat([t.reshape(-1).to(torch.float64 if t.dtype == torch.float64 else torch.float32) for t in uniq])
                dist.broadcast(flat, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
                off = 0
                for t in uniq:
                    t.copy_(flat[off: off + t.numel()].view_as(t))      # in place: bumps the version -> weights are repacked
                    off += t.numel()
                if check:                                  # cheap invariant: every rank now holds the same bytes
                    cs = flat.double().abs().sum().reshape(1)
                    lo, hi = cs.clone(), cs.clone()
                    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
                    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
                    assert float(lo) == float(hi), "replicas differ after the initial broadcast"

    def all_reduce_gradients(self) -> None:
        """DDP's gradient averaging as one collective over the flat bucket (sum, then / world).  The last element of the
        bucket is the step's non-finite flag: summed with the gradients, > 0 on every rank if any rank raised it."""
        if self.collectives:
            ev = None
            if self.time_collectives and self._bucket.is_cuda:      # bench.py: the collective's own device time, events on the step's stream
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            dist.all_reduce(self._bucket, op=dist.ReduceOp.SUM, group=self.group)
            if ev is not None:
                ev[1].record()
                self._collective_events.append(ev)
            if self.world > 1:
                self.flat_grad.div_(self.world)

    def collective_report(self, reset: bool = True) -> Dict[str, float]:
        """Mean device time of the gradient all-reduce over the steps since the last report (`time_collectives = True`; synchronises)."""
        ms = []
        for a, b in self._collective_events:
            b.synchronize()
            ms.append(a.elapsed_time(b))
        if reset:
            self._collective_events = []
        return {"all_reduce_calls": len(ms), "all_reduce_ms_mean": sum(ms) / len(ms) if ms else None,
                "all_reduce_ms_max": max(ms) if ms else None, "all_reduce_bytes": int(self._bucket.numel() * self._bucket.element_size())}

    def clip_gradients(self, grad_norm: Optional[float] = None) -> Tuple[float, float]:
        """pl_trainer.py:391-418: allow 150 % of the recent mean norm + 3 standard deviations."""
        max_grad_norm = 1.5 * self.gradnorm_queue.mean() + 3 * self.gradnorm_queue.std()
        # get_grad_norm (training_tools.py:29-55): 2-norm of the per-parameter 2-norms == 2-norm of the flat bucket
        if grad_norm is None:
            grad_norm = float(torch.linalg.vector_norm(self.flat_grad, 2.0))
        if not math.isfinite(grad_norm):                # never poison the history of norms (nan > max is False: clipping would stay off)
            return grad_norm, max_grad_norm
        if grad_norm > max_grad_norm:                  # clip_grad_norm_: g *= max_norm / (norm + 1e-6)
            self.flat_grad.mul_(max_grad_norm / (grad_norm + 1e-6))
            self.gradnorm_queue.add(float(max_grad_norm))
            print(f"Clipped gradient with value {grad_norm:.1f} while allowed {max_grad_norm:.1f}")
        else:
            self.gradnorm_queue.add(grad_norm)
        return grad_norm, max_grad_norm

    # ---- fused step (HIP module): no autograd graph, ~20 launches around the network call ----------------------------------------
    def _fused_part(self, cfg, packed, reps, cond, t_int: Tensor, noise: List[Tensor], counts: List[int], host_masks, host_sizes,
                    terms: Tensor, col0: int):
        """One (micro-)batch on the CURRENT stream: oard_loss_prepare -> oard_forward_train -> oard_loss_terms.  `terms` [2K, B_total]: the
        step's logged terms; this part owns columns [col0, col0 + B) (oard_loss_terms takes B_total both as the row stride of `terms` and
        as the 1 / B of the mean nll).  -> (nll [B], d(mean nll)/d(net), TrainState)."""
        import ctypes as C
        from . import _capi
        dyn, ls = self.dynamics, self.loss
        dev = self.flat_grad.device
        K = len(reps)
        L = _capi.lib()
        # the layout: from the batch's HOST copies of mask / size when the loader kept them (to_device) - no device -> host copy, i.e. no
        # wait for the work queued on the stream; otherwise from the device tensors.  No edge list either way (the kernels walk the
        # implicit complete graph; a caller-supplied edge_index only exists on dynamics.forward's path, where it is verified).
        combined_mask, _, n_frag_switch = ls._layout(host_masks, host_sizes, need_edges=False)
        B = int(t_int.numel())
        stream = torch.cuda.current_stream(dev).cuda_stream
        topo = dyn._get_train_topology(cfg, None, n_frag_switch, combined_mask, stream, device=dev)
        nfs = list(dyn.node_nfs)
        gamma = self._gamma_dev
        if gamma is None or gamma.device != dev:
            gamma = self._gamma_dev = ls.schedule.gamma.to(device=dev, dtype=torch.float32).contiguous()
        pos = [r["pos"].detach().to(torch.float32).contiguous() for r in reps]
        one_hot = [r["one_hot"].detach().to(torch.int64).contiguous() for r in reps]
        charge = [r["charge"].detach().to(torch.int64).contiguous() for r in reps]
        z = [torch.empty(counts[k], nfs[k], device=dev) for k in range(K)]
        eps = [torch.empty_like(x) for x in z]
        arr = lambda ts: (C.c_void_p * K)(*[t.data_ptr() for t in ts])      # noqa: E731
        f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])              # noqa: E731
        nv, nb = f3(ls.norm_values), f3(ls.norm_biases)
        sc = (C.c_float * K)(*[float(x) for x in ls.scales[:K]])
        fixed_mask = sum(1 << int(k) for k in ls.fixed_idx)
        _capi.check(L.oard_loss_prepare(C.byref(cfg), topo.handle, arr(pos), arr(one_hot), arr(charge), arr(noise), t_int.data_ptr(),
                                        gamma.data_ptr(), ls.T, nv, nb, 1 if ls.pos_only else 0, fixed_mask, arr(z), arr(eps), stream),
                    "oard_loss_prepare")
        t = (t_int / ls.T).view(B, 1)
        xs, tt, t_scalar, cnd = dyn._train_inputs(topo, z, t, cond, dev)
        net, state = dyn._run_forward_train(cfg, topo, packed, xs, tt, t_scalar, cnd, stream, reuse_tape=True)
        nll = torch.empty(B, device=dev)
        dnet = [torch.empty_like(o) for o in net]
        B_total = int(terms.shape[1])
        _capi.check(L.oard_loss_terms(C.byref(cfg), topo.handle, arr(eps), arr(net), arr(z), arr(one_hot), arr(charge), t_int.data_ptr(),
                                      gamma.data_ptr(), ls.T, nv, nb, sc, 1 if ls.pos_only else 0, B_total, nll.data_ptr(),
                                      terms.data_ptr() + 4 * col0, arr(dnet), stream), "oard_loss_terms")
        state.keep_inputs = (pos, one_hot, charge, z, eps, net, t_int, noise)      # the launches above are asynchronous
        return nll, dnet, state

    def _second_slot(self, dev):
        """State of the second micro-batch: the library's idle sub-batch stream (no fifth stream in the process: the runtime serves a
        process's streams from 4 hardware queues), a gradient buffer with the bucket's layout, and the module's per-call buffers."""
        if self._mb is None or self._mb["device"] != dev:
            import ctypes as C
            from . import _capi
            h = C.c_void_p()
            with torch.cuda.device(dev):
                _capi.check(_capi.lib().oard_library_stream(1, C.byref(h)), "oard_library_stream")
            grad2 = torch.zeros_like(self.flat_grad)
            dests2, off = {}, 0
            for p in self.params:
                dests2[id(p)] = grad2[off: off + p.numel()].view_as(p)
                off += p.numel()
            self._mb = {"device": dev, "stream": torch.cuda.ExternalStream(h.value, device=dev), "grad": grad2, "dests": dests2, "buffers": {}}
        return self._mb

    @contextlib.contextmanager
    def _slot(self, i: int):
        """The module's per-call buffers (workspace, tape, sweep scratch, NaN flag) of micro-batch i: the second one has its own."""
        if i == 0:
            yield
            return
        dyn, names = self.dynamics, ("_ws", "_tape_buf", "_train_scratch", "nan_seen")
        saved = {n: getattr(dyn, n, None) for n in names}
        store = self._mb["buffers"]
        for n in names:
            setattr(dyn, n, store.get(n))
        try:
            yield
        finally:
            for n in names:
                store[n] = getattr(dyn, n, None)
                setattr(dyn, n, saved[n])

    def _fused_forward_backward(self, batch, t_int: Optional[Tensor] = None, draw=None):
        """loss terms + gradients into the bucket: oard_loss_prepare -> oard_forward_train -> oard_loss_terms -> the backward sweep
        (training.Sweep) fed with the closed-form d(mean nll)/d(net).  Returns (nll [B], terms [2K, B]) on the device.
        Injected draws (`draw`) follow DiffusionLoss's protocol (per object: randn(n, 3) then randn(n, nf - 3)); the default is ONE
        device randn for the whole step's noise (same distribution, one launch instead of nine)."""
        import ctypes as C
        from . import _capi, training
        reps, cond = batch
        dyn, ls = self.dynamics, self.loss
        dev = self.flat_grad.device
        K = len(reps)
        L = _capi.lib()
        masks, sizes = [r["mask"] for r in reps], [r["size"] for r in reps]
        have_host = all("mask_host" in r and "size_host" in r for r in reps)
        B = int(sizes[0].numel())
        with torch.cuda.device(dev), torch.no_grad():
            main = torch.cuda.current_stream(dev)
            cfg = dyn._config()
            _capi.check(L.oard_supported(C.byref(cfg)), "oard_supported (hidden_channels/num_radial not built)")
            packed = dyn._get_packed(cfg, main.cuda_stream)
            nfs = list(dyn.node_nfs)
            if t_int is None:
                t_int = torch.randint(0, ls.T + 1, size=(B, 1), device=dev).float()
            t_int = t_int.detach().to(device=dev, dtype=torch.float32).reshape(B).contiguous()
            counts = [int(m.numel()) for m in masks]
            if draw is None:                          # ONE draw for the whole step's noise, viewed per object ([n_k, nf_k] blocks)
                flat = torch.randn(sum(c * f for c, f in zip(counts, nfs)), device=dev)
                noise, off = [], 0
                for c, f in zip(counts, nfs):
                    noise.append(flat[off: off + c * f].view(c, f))
                    off += c * f
            else:                                     # injected draws follow DiffusionLoss's protocol: (n, 3) then (n, nf - 3) per object
                noise = [torch.cat([draw((c, 3)).float(), draw((c, f - 3)).float()], dim=1).contiguous() for c, f in zip(counts, nfs)]
            dests = {id(p): p.grad for p in self.params}
            terms = torch.empty(2 * K, B, device=dev)
            if self.microbatches == 2 and have_host and B >= 2:
                return self._fused_two(cfg, packed, reps, cond, t_int, noise, B, dests, main, terms)
            hm = [r["mask_host"] for r in reps] if have_host else masks
            hs = [r["size_host"] for r in reps] if have_host else sizes
            nll, dnet, state = self._fused_part(cfg, packed, reps, cond, t_int, noise, counts, hm, hs, terms, 0)
            training.backward_sweep(dyn, state, dnet, main.cuda_stream, dests)
        return nll, terms

    def _fused_two(self, cfg, packed, reps, cond, t_int, noise, B: int, dests, main, terms):
        """The step as two micro-batches (reactions [0, h) and [h, B)) on two streams; see `microbatches` in __init__."""
        from . import training
        dyn = self.dynamics
        dev = self.flat_grad.device
        mb = self._second_slot(dev)
        side = mb["stream"]
        K = len(reps)
        h = B // 2
        n0 = [int(r["size_host"][:h].sum()) for r in reps]                 # rows of the first half per object (host arithmetic)
        halves = []
        for lo, hi, first in ((0, h, True), (h, B, False)):
            rp, hm, hs, cn = [], [], [], []
            for k, r in enumerate(reps):
                a, b = (0, n0[k]) if first else (n0[k], int(r["mask_host"].numel()))
                rp.append({"pos": r["pos"][a:b], "one_hot": r["one_hot"][a:b], "charge": r["charge"][a:b]})
                hm.append(r["mask_host"][a:b] - lo)
                hs.append(r["size_host"][lo:hi])
                cn.append(b - a)
            nz = [x[(0 if first else n0[k]): (n0[k] if first else x.shape[0])] for k, x in enumerate(noise)]
            cd = cond[lo:hi] if isinstance(cond, Tensor) else cond
            halves.append((rp, cd, t_int[lo:hi], nz, cn, hm, hs, terms, lo))
        dyn._get_packed_bwd(cfg, main.cuda_stream)             # both packs on the caller's stream BEFORE the fork (the second half reads them)
        mb["grad"].zero_()
        if mb["buffers"].get("nan_seen") is not None:
            mb["buffers"]["nan_seen"].zero_()
        side.wait_stream(main)
        with torch.cuda.stream(side), self._slot(1):
            nll1, dnet1, st1 = self._fused_part(cfg, packed, *halves[1])
        nll0, dnet0, st0 = self._fused_part(cfg, packed, *halves[0])
        with torch.cuda.stream(side), self._slot(1):
            sw1 = training.Sweep(dyn, st1, dnet1, side.cuda_stream, mb["dests"])
        sw0 = training.Sweep(dyn, st0, dnet0, main.cuda_stream, dests)
        # the two sweeps step by step: the library's gradient stream then sees their weight-gradient work alternately
        steps = [("tail",)] + [("layer", l) for l in reversed(range(sw0.NL))] + [("init",)]
        for st in steps:
            with torch.cuda.stream(side), self._slot(1):
                getattr(sw1, st[0])(*st[1:])
            getattr(sw0, st[0])(*st[1:])
        main.wait_stream(side)
        self.flat_grad.add_(mb["grad"])
        seen1 = mb["buffers"].get("nan_seen")
        if seen1 is not None and dyn.nan_seen is not None:
            dyn.nan_seen.bitwise_or_(seen1)
        return torch.cat([nll0, nll1]), terms

    def _fused_step(self, batch, **kw) -> Dict[str, float]:
        import ctypes as C  # noqa: F401
        from . import _capi
        dyn = self.dynamics
        prev = dyn.nan_check
        dyn.nan_check = "async"
        dyn.reset_nan_seen()
        self._bucket.zero_()
        try:
            nll, terms = self._fused_forward_backward(batch, **kw)
        finally:
            dyn.nan_check = prev
        K = terms.shape[0] // 2
        means = torch.cat([nll.unsqueeze(0), terms]).mean(dim=1)                  # loss, err_n[k], err_t[k]
        # non-finite anywhere (loss, local gradient, the network's device-side flag) -> the bucket's last element, summed by the all-reduce
        norm = torch.linalg.vector_norm(self.flat_grad, 2.0)
        bad = (~torch.isfinite(norm + means[0])).to(torch.float32)
        if dyn.nan_seen is not None:
            bad = bad + (dyn.nan_seen[0] != 0).to(torch.float32)
        self._bucket[-1] = bad
        if self.collectives:
            self.all_reduce_gradients()
            norm = torch.linalg.vector_norm(self.flat_grad, 2.0)
        if not self.host_sync:
            # no host read: one device thread takes the clip / skip decision, AdamW reads its scalars from device memory
            stats_dev = torch.cat([norm.reshape(1), self._bucket[-1:], means])
            state = self._clip_state if self._clip_state is not None else self._push_clip_state()
            out4 = torch.empty(4, dtype=torch.float32, device=stats_dev.device)
            o = self.optimizer.param_groups[0]
            stream = torch.cuda.current_stream(self.flat_grad.device).cuda_stream
            with torch.cuda.device(self.flat_grad.device):
                _capi.check(_capi.lib().oard_adamw_step_dev(
                    self.flat_param.data_ptr(), self.flat_grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                    self.max_exp_avg_sq.data_ptr(), self.flat_param.numel(), float(o["lr"]), float(o["betas"][0]), float(o["betas"][1]),
                    float(o.get("eps", 1e-8)), float(o.get("weight_decay", 0.0)), 1 if o.get("amsgrad", False) else 0,
                    1 if self.clip_grad else 0, state.data_ptr(), self.CLIP_CAPACITY, stats_dev.data_ptr(),
                    stats_dev.data_ptr() + 4, out4.data_ptr(), stream), "oard_adamw_step_dev")
            dyn._packed_key = dyn._packed_bwd_key = None      # the weights changed (unless the step was skipped: repacking is harmless)
            return LazyInfo(stats_dev, out4, K, [float(x) for x in self.loss.scales], self.clip_grad)
        if self._clip_state is not None:               # host_sync was switched on again: the host copies become the truth
            self._pull_clip_state()
            self._clip_state = None
        stats = torch.cat([norm.reshape(1), self._bucket[-1:], means]).tolist()        # the one host sync
        grad_norm, flag = stats[0], stats[1]
        info: Dict[str, float] = {}
        for k in range(K):
            info[f"error_t_{k}"] = stats[3 + k] / (self.loss.scales[k] + 1e-4)
            info[f"unorm_error_t_{k}"] = stats[3 + K + k]
        skipped = flag != 0 or not math.isfinite(grad_norm)
        gscale = 1.0
        if skipped:
            self.skipped_steps += 1
            print(f"Warning: non-finite loss / gradient / network output on some rank: step skipped on all ranks "
                  f"({self.skipped_steps} so far)")
            if self.clip_grad:
                info["grad_norm"], info["max_grad_norm"] = grad_norm, float("nan")
        else:
            if self.clip_grad:                        # pl_trainer.py:391-418, with the factor folded into the optimiser kernel
                max_norm = 1.5 * self.gradnorm_queue.mean() + 3 * self.gradnorm_queue.std()
                if grad_norm > max_norm:
                    gscale = max_norm / (grad_norm + 1e-6)
                    self.gradnorm_queue.add(float(max_norm))
                    print(f"Clipped gradient with value {grad_norm:.1f} while allowed {max_norm:.1f}")
                else:
                    self.gradnorm_queue.add(grad_norm)
                info["grad_norm"], info["max_grad_norm"] = grad_norm, max_norm
            o = self.optimizer.param_groups[0]
            self.opt_step += 1
            stream = torch.cuda.current_stream(self.flat_grad.device).cuda_stream
            with torch.cuda.device(self.flat_grad.device):
                _capi.check(_capi.lib().oard_adamw_step(self.flat_param.data_ptr(), self.flat_grad.data_ptr(), self.exp_avg.data_ptr(),
                                                        self.exp_avg_sq.data_ptr(), self.max_exp_avg_sq.data_ptr(), self.flat_param.numel(),
                                                        float(o["lr"]), float(o["betas"][0]), float(o["betas"][1]), float(o.get("eps", 1e-8)),
                                                        float(o.get("weight_decay", 0.0)), self.opt_step, 1 if o.get("amsgrad", False) else 0,
                                                        float(gscale), stream), "oard_adamw_step")
            dyn._packed_key = dyn._packed_bwd_key = None      # the weights changed behind torch's version counters
        info["loss"] = stats[2]
        info["skipped"] = int(skipped)
        return info

    def training_step(self, batch, **kw) -> Dict[str, float]:
        """`kw` (t_int=, draw=) injects the step's randomness for tests; by default it is drawn like the reference does.
        With the HIP module on a ROCm device and the l2 loss this is the fused step (`_fused_step`: no autograd graph, the loss
        terms, their gradient and AdamW as HIP kernels); otherwise the generic autograd formulation below.

        Non-finite values (the reference replaces a NaN network output by randn and trains on, egnn_dynamics.py:138-143 - every
        DDP rank in lock-step): here the network's device-side NaN flag and the finiteness of the local loss / gradient travel
        in the gradient bucket's last element through the SAME all-reduce; if any rank raised it, every rank skips the
        optimiser step (weights, AdamW state and the clipping history untouched), counts it in `skipped_steps` and reports
        `info["skipped"] = 1`.  No rank leaves the step before the collective, and the only host sync is the one read of
        [gradient norm, flag, loss] after it."""
        if self.fused and not kw.get("generic", False):
            return self._fused_step(batch, **{k: v for k, v in kw.items() if k != "generic"})
        kw.pop("generic", None)
        dyn = self.dynamics
        prev = getattr(dyn, "nan_check", None)
        if prev is not None:
            dyn.nan_check = "async"                    # no host sync between forward and backward; the flag is read below
            if hasattr(dyn, "reset_nan_seen"):
                dyn.reset_nan_seen()
        self._bucket.zero_()
        try:
            nll, info = self.compute_loss(batch, training=True, lazy_info=True, **kw)   # logged means stay on the device until the end
            loss = nll.mean(0)                         # pl_trainer.py:329
            loss.backward()
        finally:
            if prev is not None:
                dyn.nan_check = prev
        bad = (~torch.isfinite(loss.detach())).to(self._bucket.dtype).reshape(())
        seen = getattr(dyn, "nan_seen", None)
        if seen is not None:
            bad = bad + (seen[0] != 0).to(self._bucket.dtype)
        # a non-finite local gradient must not reach the sum as NaN only (the flag element would survive, but be explicit)
        self._bucket[-1] = bad + (~torch.isfinite(self.flat_grad.sum())).to(self._bucket.dtype)
        self.all_reduce_gradients()
        stats = torch.stack([torch.linalg.vector_norm(self.flat_grad, 2.0).to(torch.float64), self._bucket[-1].to(torch.float64),
                             loss.detach().to(torch.float64)]).tolist()            # the step's one host sync
        grad_norm, flag, loss_value = stats
        skipped = flag != 0 or not math.isfinite(grad_norm)
        if skipped:
            self.skipped_steps += 1
            print(f"Warning: non-finite loss / gradient / network output on some rank: step skipped on all ranks "
                  f"({self.skipped_steps} so far)")
            if self.clip_grad:
                info["grad_norm"], info["max_grad_norm"] = grad_norm, float("nan")
        else:
            if self.clip_grad:
                info["grad_norm"], info["max_grad_norm"] = self.clip_gradients(grad_norm)
            self.optimizer.step()
        info["loss"] = loss_value
        info["skipped"] = int(skipped)
        return {k: (float(v) if isinstance(v, Tensor) else v) for k, v in info.items()}
