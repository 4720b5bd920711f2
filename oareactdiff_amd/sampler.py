"""On-device ancestral sampler around the HIP denoising call.

`DiffusionSampler.sample` mirrors `EnVariationalDiffusion.sample`
(oa_reactdiff/diffusion/en_diffusion.py:459-560): same arguments, same return value
`(out_samples, fragments_masks)`.  Per step it issues one `oard_forward` and one `oard_sampler_step`
(mu, CoM-free noise, CoM projection, pos_only feature reset fused); the schedule scalars come from a host
table, the NaN flag stays on the device, and the reference's four `.item()` asserts per step are gone, so
the loop never synchronises the host with the GPU."""
from __future__ import annotations

import ctypes as C
from typing import Callable, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import _capi
from .dynamics import EGNNDynamics
from .graph_tools import get_edges_index, get_mask_for_frag, get_n_frag_switch
from .schedule import Schedule, get_repaint_schedule


class DiffusionSampler:
    def __init__(self, dynamics: EGNNDynamics, noise_schedule: str = "polynomial_2", timesteps: int = 1000,
                 precision: float = 1e-5, pos_only: bool = False,
                 norm_values: Sequence[float] = (1.0, 1.0, 1.0), norm_biases: Sequence[float] = (0.0, 0.0, 0.0),
                 on_nan: str = "raise", gaussian_prior_std: Optional[float] = None):
        """`gaussian_prior_std` (measurement aid, default off): add to every network prediction the ideal denoiser of a Gaussian data
        prior x0 ~ N(0, s^2) per coordinate, eps = sigma_t z_t / (alpha_t^2 s^2 + sigma_t^2), on the position columns.  An UNTRAINED
        network predicts eps ~ 0, and the ancestral sampler then scales its state by 1 / alpha_t|s every step (en_diffusion.py:614-618):
        positions reach hundreds of Angstrom within a few calls, every pair leaves the cutoff and the remaining calls run on an empty
        radius graph.  With the prior term the chain is the exact reverse process of N(0, s^2) data: the state keeps the marginal
        alpha_t^2 s^2 + sigma_t^2 (= 1 for s = 1, the distribution of the headline's inputs) and every network call sees a
        molecule-sized cloud, as a trained model's would.  One in-place axpy per object and step on the loop's stream; the network
        call itself is untouched.  Used by bench.py's T = 1000 line and tests/test_configs.py (the float64 replay adds the same term).

        `on_nan`: what to do when any network call of a run predicted a NaN displacement.  The reference replaces that
        step's velocity by randn, prints a warning and keeps sampling (egnn_dynamics.py:138-143) - one host sync per
        step.  Here the loop never syncs: every call ORs its NaN flag into a device-side sticky flag which is read ONCE
        after the loop; "raise" (default) raises FloatingPointError, "warn" warns and returns the (NaN) samples,
        "replace" reproduces the reference: the step's velocity becomes CoM-free randn ON THE DEVICE (oard_nan_replace, keyed on the
        call's device-side flag; costs one extra randn draw per object and step, so the noise stream of a seeded run differs from the
        other modes) and a warning is printed once after the loop."""
        assert on_nan in ("raise", "warn", "replace")
        self.on_nan = on_nan
        self.prior_std = None if gaussian_prior_std is None else float(gaussian_prior_std)
        self.dynamics = dynamics
        self.schedule = Schedule(noise_schedule, timesteps, precision)
        self.T = timesteps
        self.pos_only = pos_only
        self.pos_dim = dynamics.pos_dim
        self.node_nfs = dynamics.node_nfs
        self.norm_values = tuple(norm_values)
        self.norm_biases = tuple(norm_biases)
        self._layout_cache = {}
        #: upper bound on the pre-drawn noise of the hipGraph-replayed loop (`sample(graph=True)` / small batches)
        self.noise_block_bytes = 64 << 20

    def _layout(self, fragments_nodes: List[Tensor], dev):
        """Batch-layout tensors (masks, combined_mask, edge_index, n_frag_switch) on the device, cached by the
        atom counts so that repeated sample()/inpaint() calls reuse the same tensors (and therefore the
        dynamics' cached index tables)."""
        key = tuple(tuple(int(v) for v in f.tolist()) for f in fragments_nodes)
        hit = self._layout_cache.get(key)
        if hit is None:
            fn = [f.to(dev) for f in fragments_nodes]
            masks = [get_mask_for_frag(f) for f in fn]
            combined_mask = torch.cat(masks)
            hit = (masks, combined_mask, get_edges_index(combined_mask, remove_self_edge=True), get_n_frag_switch(fn))
            if len(self._layout_cache) >= 4:
                self._layout_cache.pop(next(iter(self._layout_cache)))
            self._layout_cache[key] = hit
        return hit

    def _check_nan(self, dyn) -> None:
        """The single host read of a sampling run: did any of its network calls produce a NaN displacement?"""
        if dyn.nan_seen is not None and int(dyn.nan_seen[0].item()) != 0:
            if self.on_nan == "replace":
                print("Warning: detected nan in pos, resetting EGNN output to randn.")       # the reference's message (:139)
                return
            msg = ("a network call of this sampling run predicted NaN positions; the reference would have replaced that "
                   "step by randn (egnn_dynamics.py:138-143), this loop does not: the affected samples are NaN")
            if self.on_nan == "raise":
                raise FloatingPointError(msg)
            import warnings
            warnings.warn(msg)

    def prior_coefficient(self, step: int, n_steps: int) -> float:
        """sigma_t / (alpha_t^2 s^2 + sigma_t^2) at time step / n_steps: E[eps | z_t] = coefficient * z_t for x0 ~ N(0, s^2)."""
        a, sg = self.schedule.alpha_sigma(step, n_steps)
        return sg / (a * a * self.prior_std * self.prior_std + sg * sg)

    def _prior_table(self, timesteps: int, dev) -> Tensor:
        """[T + 1] float32 on the device: entry k = the coefficient at time k / T.  Built once per run; both loops read their coefficient
        from it as a [1] device tensor, so the eager loop and the hipGraph replay run the same kernel on the same operands (bit-identical)."""
        return torch.tensor([self.prior_coefficient(k, timesteps) for k in range(timesteps + 1)], dtype=torch.float32, device=dev)

    def _add_prior(self, eps_hat, z, coef: Tensor) -> None:
        """eps_hat[:, :pos_dim] += coef * z[:, :pos_dim] per object; `coef` a [1] device tensor."""
        pd = self.pos_dim
        for e, x in zip(eps_hat, z):
            if e.numel():
                e[:, :pd].addcmul_(x[:, :pd], coef)

    # --------------------------------------------------------------------------------------------
    def _step_kernel(self, topo, mode, z, eh, noise, h0, a, b, c, out, stream):
        L = _capi.lib()
        cfg = self.dynamics._config()
        n = len(self.node_nfs)
        arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts]) if ts is not None else None
        rc = L.oard_sampler_step(C.byref(cfg), topo.handle, mode, arr(z), arr(eh), arr(noise), arr(h0),
                                 C.c_float(a), C.c_float(b), C.c_float(c), 1 if self.pos_only else 0, arr(out), stream)
        _capi.check(rc, "oard_sampler_step")

    def _step_kernel_dev(self, topo, mode, z, eh, noise, h0, coef, out, stream):
        """As `_step_kernel` with the schedule scalars in device memory (`coef` [3]): the launch a hipGraph replays."""
        L = _capi.lib()
        cfg = self.dynamics._config()
        n = len(self.node_nfs)
        arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts]) if ts is not None else None
        rc = L.oard_sampler_step_dev(C.byref(cfg), topo.handle, mode, arr(z), arr(eh), arr(noise), arr(h0), coef.data_ptr(),
                                     1 if self.pos_only else 0, arr(out), stream)
        _capi.check(rc, "oard_sampler_step_dev")

    def _graphed_steps(self, dyn, topo, timesteps, za, draw, h0d, edge_index, conditions, n_frag_switch, combined_mask, dev):
        """The T ancestral steps of `sample` as ONE captured hipGraph replayed T times (small batches are launch-bound:
        ~100 kernel launches per step).  The per-step scalars live in device tables indexed by a device-side step counter:
        t, the three schedule coefficients and the step's noise draw; the graph holds [table lookups, oard_forward,
        oard_sampler_step_dev, z <- z_new, counter += 1].  Same kernels and same arithmetic as the eager loop, so the
        result is bit-identical.  Captured on a side stream after an eager warm-up step, as torch.cuda.graph requires."""
        n_obj = len(self.node_nfs)
        steps = list(reversed(range(timesteps)))
        coefs = [self.schedule.step(s, timesteps) for s in steps]
        coef_tab = torch.tensor([[c.alpha_ts, c.c_eps, c.sigma] for c in coefs], dtype=torch.float32, device=dev)      # [T,3]
        # the eager loop's t values, bit for bit: (arange(T + 1) / T)[s + 1]
        t_tab = (torch.arange(timesteps + 1, device=dev, dtype=torch.float32) / timesteps)[torch.tensor([s + 1 for s in steps], device=dev)]
        prior_tab = (self._prior_table(timesteps, dev)[torch.tensor([s + 1 for s in steps], device=dev)]
                     if self.prior_std is not None else None)
        # The noise of the steps is drawn in BLOCKS (not all T draws up front: that would be T x the eager loop's memory -
        # a [T, n, nf] table per object): at most `noise_block_bytes` of draws exist at a time, refilled between replays in
        # the eager loop's draw order (call 1 .. T), so a seeded run consumes the generator exactly like the eager loop.
        per_step = 4 * sum(int(x.numel()) for x in za)
        block = max(1, min(timesteps, self.noise_block_bytes // max(per_step, 1)))
        noise_tab = [torch.empty((block,) + tuple(x.shape), dtype=torch.float32, device=dev) for x in za]              # [block, n_k, nf_k]
        counter = torch.zeros(1, dtype=torch.long, device=dev)           # step index (t / coefficient tables)
        slot = torch.zeros(1, dtype=torch.long, device=dev)              # row of the current block's noise table
        z = [x.clone() for x in za]
        z_new = [torch.empty_like(x) for x in za]
        next_call = [1]

        def refill(n_rows):
            for r in range(n_rows):
                d = draw(next_call[0])
                next_call[0] += 1
                for k in range(n_obj):
                    noise_tab[k][r].copy_(d[k])
            slot.zero_()

        def one_step():
            t_cur = t_tab.index_select(0, counter)                       # [1]: a 1-D t = one value for every node
            coef = coef_tab.index_select(0, counter).view(3)
            noise = [nt.index_select(0, slot)[0] for nt in noise_tab]
            eps_hat, _ = dyn(z, edge_index, t_cur, conditions, n_frag_switch, combined_mask)
            if prior_tab is not None:
                self._add_prior(eps_hat, z, prior_tab.index_select(0, counter))
            stream = torch.cuda.current_stream(dev).cuda_stream
            self._step_kernel_dev(topo, 0, z, eps_hat, noise, h0d if self.pos_only else None, coef, z_new, stream)
            for a, b in zip(z, z_new):
                a.copy_(b)
            counter.add_(1)
            slot.add_(1)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        done = 0
        with torch.cuda.stream(side):                                    # eager warm-up on the capture stream (step 0)
            refill(min(block, timesteps))
            one_step()
            done, in_block = 1, 1
            if timesteps > 1:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    one_step()
                while done < timesteps:
                    if in_block == block:
                        refill(min(block, timesteps - done))
                        in_block = 0
                    g.replay()
                    done += 1
                    in_block += 1
        torch.cuda.current_stream(dev).wait_stream(side)
        return z

    @torch.no_grad()
    def sample(self, n_samples: int, fragments_nodes: List[Tensor], conditions: Optional[Tensor] = None,
               return_frames: int = 1, timesteps: Optional[int] = None, h0: Optional[List[Tensor]] = None,
               noise_fn: Optional[Callable[[int], List[Tensor]]] = None, graph: Optional[bool] = None,
               step_callback: Optional[Callable[[int], None]] = None) -> Tuple[list, List[Tensor]]:
        """`noise_fn(i)` (tests) supplies the i-th set of raw N(0,1) draws, one [n_k, node_nf_k] tensor per
        object (i = 0 initial state, 1..T the steps, T+1 the final draw); default: torch.randn on the device.
        `graph`: replay the step as a captured hipGraph (`_graphed_steps`; bit-identical to the eager loop, noise drawn in
        blocks of at most `self.noise_block_bytes`); None = automatically for launch-bound batches (n_samples <= 8, one returned
        frame) unless the caller's stream is itself being captured (then the eager loop runs, which is capturable).
        `step_callback(i)`: called on the host after the i-th network call of the eager loop has been enqueued (i = 1 ... T; progress
        bars, timing events - it must not synchronise if the loop is to stay free of host waits)."""
        timesteps = self.T if timesteps is None else timesteps
        assert 0 < return_frames <= timesteps and timesteps % return_frames == 0       # en_diffusion.py:473-475
        assert h0 is not None if self.pos_only else True
        dyn = self.dynamics
        dev = next(dyn.parameters()).device
        if dev.type != "cuda":
            raise _capi.OardError("DiffusionSampler needs the dynamics on a ROCm device (no CPU fallback)")
        n_obj = len(self.node_nfs)
        masks, combined_mask, edge_index, n_frag_switch = self._layout(fragments_nodes, dev)
        if conditions is None:
            conditions = torch.zeros(n_samples, max(dyn.condition_nf, 1), device=dev)
        conditions = conditions.to(dev)
        h0d = [h.to(device=dev, dtype=torch.float32).contiguous() for h in h0] if h0 is not None else None
        sizes = [int(m.numel()) for m in masks]
        old_nan = dyn.nan_check
        dyn.nan_check = "replace" if self.on_nan == "replace" else "async"
        dyn.reset_nan_seen()
        try:
            with torch.cuda.device(dev):
                stream = torch.cuda.current_stream(dev).cuda_stream
                cfg = dyn._config()
                topo = dyn._get_topology(cfg, edge_index, n_frag_switch, combined_mask, stream)
                if topo.handle is None:
                    raise _capi.OardError("the sampling loops run on the production kernels: hidden_channels / num_radial must be a built width "
                                          "pair (OARD_DIMS=... python -m oareactdiff_amd.build) and a (sample, object) group at most 1024 atoms")

                def draw(i):
                    if noise_fn is not None:
                        return [x.to(device=dev, dtype=torch.float32).contiguous() for x in noise_fn(i)]
                    return [torch.randn(sizes[k], self.node_nfs[k], device=dev) for k in range(n_obj)]

                za = [torch.empty(sizes[k], self.node_nfs[k], device=dev) for k in range(n_obj)]
                zb = [torch.empty_like(z) for z in za]
                self._step_kernel(topo, 2, None, None, draw(0), h0d if self.pos_only else None, 0.0, 0.0, 1.0, za, stream)
                t_table = torch.arange(timesteps + 1, device=dev, dtype=torch.float32) / timesteps
                ptab = self._prior_table(timesteps, dev) if self.prior_std is not None else None
                out_samples = [None] * return_frames
                call = 1
                use_graph = ((n_samples <= 8 and return_frames == 1 and not torch.cuda.is_current_stream_capturing())
                             if graph is None else bool(graph))
                if use_graph:
                    assert return_frames == 1, "intermediate frames need the eager loop"
                    za = self._graphed_steps(dyn, topo, timesteps, za, draw, h0d, edge_index, conditions, n_frag_switch,
                                             combined_mask, dev)
                    call = timesteps + 1
                for s in (reversed(range(timesteps)) if not use_graph else ()):
                    co = self.schedule.step(s, timesteps)
                    eps_hat, _ = dyn(za, edge_index, t_table[s + 1: s + 2], conditions, n_frag_switch, combined_mask)
                    if ptab is not None:
                        self._add_prior(eps_hat, za, ptab[s + 1: s + 2])
                    self._step_kernel(topo, 0, za, eps_hat, draw(call), h0d if self.pos_only else None,
                                      co.alpha_ts, co.c_eps, co.sigma, zb, stream)
                    if step_callback is not None:
                        step_callback(call)
                    call += 1
                    za, zb = zb, za
                    if (s * return_frames) % timesteps == 0 and return_frames > 1:
                        out_samples[(s * return_frames) // timesteps] = self._unnormalize_z([z.clone() for z in za])
                fc = self.schedule.final()
                eps_hat, _ = dyn(za, edge_index, t_table[0:1], conditions, n_frag_switch, combined_mask)
                if ptab is not None:
                    self._add_prior(eps_hat, za, ptab[0:1])
                self._step_kernel(topo, 1, za, eps_hat, draw(call), None, fc.inv_alpha_0, fc.sigma_0, fc.sigma_x, zb, stream)
                x = zb
                self.last_x = x
                self.last_status = dyn.last_status
        finally:
            dyn.nan_check = old_nan
        self._check_nan(dyn)
        nv, nb, pd = self.norm_values, self.norm_biases, self.pos_dim
        pos = [x[k][:, :pd] * nv[0] + nb[0] for k in range(n_obj)]                              # :680-683
        if self.pos_only:
            cat = [h[:, :-1] for h in h0d]                                                      # :542-544
            charge = [h[:, -1:] for h in h0d]
        else:
            cat = [torch.nn.functional.one_hot(torch.argmax(x[k][:, pd:-1] * nv[1] + nb[1], dim=1),
                                               self.node_nfs[k] - 4).long() for k in range(n_obj)]
            charge = [torch.round(x[k][:, -1:] * nv[2] + nb[2]).long() for k in range(n_obj)]
        out_samples[0] = [torch.cat([pos[k], cat[k], charge[k]], dim=1) for k in range(n_obj)]  # :554-557
        return out_samples, masks

    @torch.no_grad()
    def sample_sharded(self, fragments_nodes: List[Tensor], conditions: Optional[Tensor] = None, seed: int = 0,
                       gather: bool = False, h0: Optional[List[Tensor]] = None, **kw):
        """BASELINE configs[2]: a GLOBAL batch (per-object atom counts of all B reactions) sharded over the ranks of the
        default process group, one process per GPU.  Reactions are independent (utils/_graph_tools.py:30), so every rank
        samples its contiguous slice `shard_range(B, rank, world)` with its own RNG stream (`seed + rank`) and NO
        data-path collective.  `gather=True` adds one host-side gather of the final samples on rank 0 (the only
        communication of the whole run).  Returns (samples, masks, (lo, hi)); with `gather` rank 0 gets the samples of the
        whole batch per object (in batch order), the other ranks their own."""
        import torch.distributed as dist
        from .shard import shard_range
        on = dist.is_available() and dist.is_initialized()
        rank, world = (dist.get_rank(), dist.get_world_size()) if on else (0, 1)
        B = int(fragments_nodes[0].numel())
        lo, hi = shard_range(B, rank, world)
        local = [f[lo:hi] for f in fragments_nodes]
        cond = conditions[lo:hi] if conditions is not None else None
        h0l = None
        if h0 is not None:
            h0l = []
            for f, h in zip(fragments_nodes, h0):
                off = torch.cumsum(torch.cat([torch.zeros(1, dtype=torch.long), f.cpu().long()]), 0)
                h0l.append(h[int(off[lo]): int(off[hi])])
        torch.manual_seed(seed + rank)                     # seeds the device generators too
        out, masks = self.sample(hi - lo, local, conditions=cond, h0=h0l, **kw)
        if gather and world > 1:
            mine = (lo, [o.cpu() for o in out[0]])
            got = [None] * world if rank == 0 else None
            dist.gather_object(mine, got, dst=0)
            if rank == 0:
                got.sort(key=lambda t: t[0])
                out = list(out)
                out[0] = [torch.cat([g[1][k] for g in got], dim=0) for k in range(len(self.node_nfs))]
        return out, masks, (lo, hi)

    @torch.no_grad()
    def inpaint(self, n_samples: int, fragments_nodes: List[Tensor], conditions: Optional[Tensor] = None,
                return_frames: int = 1, resamplings: int = 1, jump_length: int = 1, timesteps: Optional[int] = None,
                xh_fixed: Optional[List[Tensor]] = None, frag_fixed: Optional[List[int]] = None,
                noise_fn: Optional[Callable[[int], List[Tensor]]] = None) -> Tuple[list, List[Tensor]]:
        """RePaint-style conditional generation, mirror of `EnVariationalDiffusion.inpaint`
        (en_diffusion.py:722-883): objects in `frag_fixed` follow q(z_s | x_fixed), the others are denoised,
        with `resamplings` x `jump_length` forward jumps.  Noise draws are consumed in the reference's order
        (per step: known-branch noise, denoising noise, then the jump noise if any)."""
        timesteps = self.T if timesteps is None else timesteps
        assert 0 < return_frames <= timesteps and timesteps % return_frames == 0
        assert xh_fixed is not None and len(xh_fixed)
        frag_fixed = list(frag_fixed or [])
        dyn = self.dynamics
        dev = next(dyn.parameters()).device
        if dev.type != "cuda":
            raise _capi.OardError("DiffusionSampler needs the dynamics on a ROCm device (no CPU fallback)")
        n_obj = len(self.node_nfs)
        masks, combined_mask, edge_index, n_frag_switch = self._layout(fragments_nodes, dev)
        if conditions is None:
            conditions = torch.zeros(n_samples, max(dyn.condition_nf, 1), device=dev)
        conditions = conditions.to(dev)
        pd = self.pos_dim
        xf = [x.to(device=dev, dtype=torch.float32).clone() for x in xh_fixed]
        h0 = [x[:, pd:].long().to(torch.float32).contiguous() for x in xf]                 # :753
        for k in range(n_obj):                                                              # :755-759
            xf[k][:, :pd] = EGNNDynamics.remove_mean_batch(xf[k][:, :pd], masks[k])
        sizes = [int(m.numel()) for m in masks]
        old_nan = dyn.nan_check
        dyn.nan_check = "replace" if self.on_nan == "replace" else "async"
        dyn.reset_nan_seen()
        try:
            with torch.cuda.device(dev):
                stream = torch.cuda.current_stream(dev).cuda_stream
                cfg = dyn._config()
                topo = dyn._get_topology(cfg, edge_index, n_frag_switch, combined_mask, stream)
                if topo.handle is None:
                    raise _capi.OardError("the sampling loops run on the production kernels: hidden_channels / num_radial must be a built width "
                                          "pair (OARD_DIMS=... python -m oareactdiff_amd.build) and a (sample, object) group at most 1024 atoms")
                counter = [0]

                def draw():
                    i = counter[0]
                    counter[0] += 1
                    if noise_fn is not None:
                        return [x.to(device=dev, dtype=torch.float32).contiguous() for x in noise_fn(i)]
                    return [torch.randn(sizes[k], self.node_nfs[k], device=dev) for k in range(n_obj)]

                new = lambda: [torch.empty(sizes[k], self.node_nfs[k], device=dev) for k in range(n_obj)]
                hsel = h0 if self.pos_only else None
                zt = new()
                self._step_kernel(topo, 2, None, None, draw(), hsel, 0.0, 0.0, 1.0, zt, stream)
                t_table = torch.arange(timesteps + 1, device=dev, dtype=torch.float32) / timesteps
                schedule = get_repaint_schedule(resamplings, jump_length, timesteps)
                s = timesteps - 1
                for i, n_denoise in enumerate(schedule):
                    for j in range(n_denoise):
                        a_s, sig_s = self.schedule.alpha_sigma(s, timesteps)
                        known = new()
                        self._step_kernel(topo, 3, xf, None, draw(), hsel, a_s, 0.0, sig_s, known, stream)   # :797-805
                        co = self.schedule.step(s, timesteps)
                        eps_hat, _ = dyn(zt, edge_index, t_table[s + 1: s + 2], conditions, n_frag_switch, combined_mask)
                        unknown = new()
                        self._step_kernel(topo, 0, zt, eps_hat, draw(), hsel, co.alpha_ts, co.c_eps, co.sigma, unknown, stream)
                        zt = [known[k] if k in frag_fixed else unknown[k] for k in range(n_obj)]            # :827-830
                        if j == n_denoise - 1 and i < len(schedule) - 1:                                   # :833-850
                            t = s + jump_length
                            a_ts, sig_ts = self.schedule.forward_jump(s, t, timesteps)
                            jumped = new()
                            self._step_kernel(topo, 4, zt, None, draw(), None, a_ts, 0.0, sig_ts, jumped, stream)
                            zt = jumped
                            s = t
                        s -= 1
                fc = self.schedule.final()
                eps_hat, _ = dyn(zt, edge_index, t_table[0:1], conditions, n_frag_switch, combined_mask)
                x = new()
                self._step_kernel(topo, 1, zt, eps_hat, draw(), None, fc.inv_alpha_0, fc.sigma_0, fc.sigma_x, x, stream)
                self.last_x = x
                self.last_status = dyn.last_status
        finally:
            dyn.nan_check = old_nan
        self._check_nan(dyn)
        nv, nb = self.norm_values, self.norm_biases
        pos = [x[k][:, :pd] * nv[0] + nb[0] for k in range(n_obj)]
        if self.pos_only:
            cat = [h[:, :-1].long() for h in h0]
            charge = [h[:, -1:].long() for h in h0]
        else:
            cat = [torch.nn.functional.one_hot(torch.argmax(x[k][:, pd:-1] * nv[1] + nb[1], dim=1),
                                               self.node_nfs[k] - 4).long() for k in range(n_obj)]
            charge = [torch.round(x[k][:, -1:] * nv[2] + nb[2]).long() for k in range(n_obj)]
        out_samples = [None] * return_frames
        out_samples[0] = [torch.cat([pos[k], cat[k], charge[k]], dim=1) for k in range(n_obj)]
        return out_samples, masks

    def _unnormalize_z(self, z: List[Tensor]) -> List[Tensor]:
        nv, nb, pd = self.norm_values, self.norm_biases, self.pos_dim
        for k in range(len(z)):
            z[k][:, :pd] = z[k][:, :pd] * nv[0] + nb[0]
            z[k][:, pd:-1] = z[k][:, pd:-1] * nv[1] + nb[1]
            z[k][:, -1:] = z[k][:, -1:] * nv[2] + nb[2]
        return z
