"""One GPS training step on MI355X: the body of the reference's `DefaultTrainer.train_step` /
`backward` (trainer/default_trainer.py:18-24,30-48) without its logging, as a reusable object.

    forward(data_dict) -> Loss -> backward -> clip_grad_norm_(grad_norm) -> AdamW.step -> LambdaLR.step

Data parallelism is the reference's only strategy (Accelerate -> torch DDP over NCCL,
trainer/build.py:66-75,121).  Here: one process per GPU, `torch.distributed` backend "nccl"
(= RCCL over xGMI), torch DDP with gradient buckets viewed in place and all-reduced (fp32, like the
reference) while backward is still running.  The reference sets `find_unused_parameters=True` because 13
trainable tensors never receive a gradient; here those are found by one probe
step, agreed across ranks and frozen, unless `find_unused_parameters=True` asks for the reference's setting.

bf16: the transformer stack, BERT, heads and losses run under `torch.autocast(bfloat16)`;
the point ops always compute in fp32 (indices must be bit-exact).

HIP graph (`graph=True`, single process): the step issues ~1 800 launches from ~37 ms of Python
per step, which is what bounds it once the kernels are fused.  After
`graph_warmup` eager steps the whole step (forward, losses, backward, clipping, AdamW) is captured
once into a HIP graph on static input buffers and then replayed: one `hipGraphLaunch` per step.
The learning rate is a device tensor the scheduler fills in place, attention dropout draws its
seed from a device word (fused_attention._next_device_seed), torch's own RNG ops are graph-safe,
so every replay is a fresh, correctly scheduled optimisation step.

HIP graph + data parallelism (`graph=True` with world_size > 1, "graph_dp"): torch DDP's bucket hooks
cannot live inside a replayed graph, and capturing RCCL collectives cannot be validated on a 1-GPU
box, so the step is split at its two exchange points instead and the collectives stay EAGER:

    graph 1  model forward                                        (replay)
    eager    all-gather of the contrastive features (C2)          RCCL, into static buffers
    graph 2a losses + backward of the TOP segment (joint encoder, heads) into the flat fp32 gradient buffer
    eager    all-reduce of the top range of the buffer (C1, part 1)  RCCL, asynchronous: runs beside graph 2b
    graph 2b backward of the TEXT encoder from its boundary gradients
    eager    all-reduce of the text range (C1, part 2), asynchronous: runs beside graph 2c
    graph 2c backward of the OBJECT encoder
    eager    all-reduce of the object range (C1, part 3), wait for all, / world
    graph 3  gradient clipping + AdamW                             (replay)

[r3] The backward pass is cut at the outputs of the text and object encoders (model._stage_boundary;
torch.autograd.backward(inputs=...) restricts each segment to its own parameters), so about half of the 491 MB
gradient exchange overlaps the bottom segment's ~5 ms of backward; only the bottom range's all-reduce is exposed.
Parameters are broadcast from rank 0 once.  The same code runs with world_size 1 (collectives are
no-ops), which is how tests/test_gpu_model.py exercises it on one GPU.
"""
from __future__ import annotations

import contextlib
from typing import Optional

import torch
import torch.nn as nn

from .common import dist_utils

_CAPTURE_MODE = "thread_local"     # split-graph data-parallel captures (see _graph_dp_step)
from .model.build import build_model
from .optim.build import build_optim


class GPSTrainStep:
    def __init__(self, cfg, device: torch.device | str = "cuda", total_steps: int = 100000,
                 amp_dtype: Optional[torch.dtype] = torch.bfloat16, ddp: Optional[bool] = None,
                 bucket_cap_mb: int = 64, seed: int = 42, graph: bool = False, graph_warmup: int = 3,
                 native_gemm: bool = True, grad_compress: Optional[str] = None, native_optimizer: bool = True,
                 fused_lm_loss: bool = True, find_unused_parameters: bool = False, wgrad_overlap: bool = False,
                 wgrad_group: bool = True):
        self.cfg = cfg
        self.device = torch.device(device)
        # projections / FFNs of the transformer stacks on libgps_hip.so's MFMA GEMMs (modules/layers/gemm.py);
        # False = hipBLASLt through F.linear, kept for A/B runs
        from .modules.layers import gemm as _gemm
        _gemm.set_gemm_backend(bool(native_gemm))
        _gemm.reset_grouped_bookkeeping()
        torch.manual_seed(seed)
        self.model = build_model(cfg).to(self.device)
        world = dist_utils.get_world_size()
        use_ddp = (world > 1) if ddp is None else ddp
        want_graph = bool(graph) and self.device.type == "cuda"
        # graph == "dp" forces the split-graph data-parallel form even at world_size 1 (tests)
        self.graph_dp = want_graph and (graph == "dp" or (use_ddp and world > 1))
        if self.graph_dp:
            use_ddp = False
        self.world = world
        self.graph = want_graph and not use_ddp and not self.graph_dp
        if self.graph or self.graph_dp:
            # [r4] torch warns when a parameter's AccumulateGrad node runs on another stream than the node that produced
            # its gradient ("... break CUDA graph capture ...").  Round 3 silenced that warning; it was the root cause of
            # the corrupted split-graph steps (see _work_stream below).  It is an ERROR here: a captured backward must be
            # a linear chain of nodes.
            # (scoped to this engine's steps: `_strict_accumulate_grad` -- no process-wide warning filter)
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(True)
        self.graph_warmup = max(1, int(graph_warmup))
        self._graph = None
        self._static = None
        param_groups = self.model.get_opt_params()
        if not native_optimizer:                       # A/B: torch's AdamW + clip_grad_norm_ instead of gps_adamw_step
            cfg.solver.optim.args["native_optimizer"] = False
        if self.graph or self.graph_dp:
            # capturable optimizer state: step counters and learning rates live on the device
            cfg.solver.optim.args["capturable"] = True
            # the live learning rates are 0-dim views of ONE device vector: the scheduler's update is one host-to-device
            # copy per step (torch's LambdaLR.step fills every group's tensor separately: 16 launches per step on the
            # stream the graph replays on)
            self._lr_dev = torch.tensor([float(g["lr"]) for g in param_groups], dtype=torch.float32, device=self.device)
            # pinned staging ring: an asynchronous copy reads its source when the GPU gets to it, possibly several steps
            # after the host queued it -- every step writes its own slot (the host never runs 256 steps ahead)
            self._lr_host = torch.empty((256, len(param_groups)), dtype=torch.float32).pin_memory()
            for gi, g in enumerate(param_groups):
                # float `initial_lr` keeps LambdaLR's arithmetic on the host (a tensor base lr would
                # cost one .item() sync per group and step); the live `lr` is filled in place
                g["initial_lr"] = float(g["lr"])
                g["lr"] = self._lr_dev[gi]
        self.loss, self.optimizer, self.scheduler = build_optim(cfg, param_groups, total_steps)
        self.loss = self.loss.to(self.device)
        self.grad_norm = cfg.solver.get("grad_norm", None)
        self.amp_dtype = amp_dtype if self.device.type == "cuda" else None
        self.net: nn.Module = self.model
        # DDP is built lazily at the first step: that step first runs ONE local forward + backward to find the
        # trainable tensors that never receive a gradient in this configuration (13 in the pre-train config: BERT
        # pooler, sem_cls_embed_layer, sem_mask_embeddings, obj_pred_head -- SURVEY.md 2b C1), freezes them, and
        # only then wraps the model: the reducer needs no per-iteration unused-parameter search
        # (find_unused_parameters=False; the reference pays for the search every step, trainer/build.py:66) and the
        # buckets hold exactly the tensors that are reduced.
        self._want_ddp = bool(use_ddp)
        self._ddp_kw = dict(gradient_as_bucket_view=True, bucket_cap_mb=bucket_cap_mb)
        # gradient exchange.  None (default) = the reference's: fp32 buckets all-reduced in fp32 (trainer/build.py:66-75).
        # Opt-in compression, used by bench.py and named in its config line:
        #   "bf16_fp32acc"  bf16 on the wire (246 MB instead of 491 MB per step), every rank's contribution rounded to
        #                   bf16 ONCE, the cross-rank sum accumulated in fp32 (all-to-all of bucket slices, local fp32
        #                   sum, all-gather of the bf16 mean): common/dist_utils.bf16_wire_fp32_acc_hook
        #   "bf16"          torch's bf16_compress_hook: all-reduce IN bf16 (the sum itself is accumulated in bf16,
        #                   ~2^-8 relative per element, growing with world size)
        self.grad_compress = grad_compress
        # True = the reference's DDP setting (per-step unused-parameter search, tolerates data-dependent branches);
        # False = freeze the tensors no rank ever gives a gradient in the probe step and skip the search
        self.find_unused_parameters = bool(find_unused_parameters)
        self.fused_lm_loss = bool(fused_lm_loss) and bool(native_gemm)
        # weight-gradient GEMMs on a side stream beside the input-gradient chain (modules/layers/gemm.deferred_wgrads);
        # never under torch DDP, whose reducer hooks need autograd's own gradient accumulation.  OFF by default:
        # measured slower on one MI355X (profiles/r3/bench_overlap_ab.json: 20.5 vs 19.3 ms per step) -- both GEMM
        # forms already occupy every CU's LDS with two workgroups, so a second stream's workgroups queue behind them
        # instead of filling idle matrix-pipe time, and the two kernels evict each other's L2 panels
        self.wgrad_overlap = bool(wgrad_overlap) and bool(native_gemm) and self.device.type == "cuda"
        # [r4] weight / bias gradients of the native Linears deferred to ONE grouped launch per backward segment
        # (modules/layers/gemm.grouped_wgrads -> gps_gemm_wgrad_grouped: no split over K, no partial tiles, results
        # written straight into param.grad / the flat gradient buffer).  Never under torch DDP (reducer hooks).
        self.wgrad_group = bool(wgrad_group) and bool(native_gemm) and self.device.type == "cuda" and not self.wgrad_overlap
        self.frozen_unused: list = []
        self.global_step = 0
        if not hasattr(self, "_lr_dev"):
            self._lr_dev = self._lr_host = None
        # [r4] ONE stream for everything the graph modes run outside a replay: the eager warm-up steps, every capture, the
        # eager steps after a capture.  A parameter's AccumulateGrad node remembers the stream that was current when it
        # was created and outlives a step whenever anything still references that step's autograd graph; its
        # `grad += dW` is then issued on THAT stream.  Inside a capture this forks the graph (the accumulations become
        # parallel branches of the compute chain), and ROCm 7's multi-queue graph executor does not keep such graphs in
        # order: the text encoder's saved activations were overwritten by later kernels of the same graph (wrong
        # gradients, memory-aperture violations; regression test: tests/test_gpu_graph_chain.py, DESIGN.md
        # section 9a).  With one stream every capture is a linear chain of nodes.
        self._work_stream = None
        self._acc_hooks, self._autograd_written = None, set()
        self._exchange_events = None          # (start, end) events around the EXPOSED part of the gradient exchange
        # tests / diagnostics: called with a stage name at the capture / replay points of the split-graph step
        self.stage_hook = None
        if self.graph_dp and dist_utils.is_dist():
            with torch.no_grad():                      # what DDP does at construction
                for t in list(self.model.parameters()) + list(self.model.buffers()):
                    torch.distributed.broadcast(t, src=0)

    # ---- torch DDP, built at the first step ----------------------------------------------------------
    def _needs_buffer_broadcast(self) -> bool:
        """Buffers that change during training (BatchNorm running statistics of an unfrozen encoder) must stay
        identical across ranks, as under the reference's Accelerate/DDP default; with every BN layer frozen
        (GPS pre-train / fine-tune configs) there is nothing to broadcast."""
        for m in self.model.modules():
            if isinstance(m, nn.modules.batchnorm._BatchNorm) and m.training and m.track_running_stats:
                return True
        return False

    def prepare(self, data_dict) -> None:
        """Build the DDP wrapper now (it is otherwise built by the first `step`)."""
        if self._want_ddp:
            self._build_ddp(data_dict)

    def _build_ddp(self, data_dict) -> None:
        from torch.nn.parallel import DistributedDataParallel as DDP
        self.model.train()
        if not self.find_unused_parameters:
            self._freeze_never_used(data_dict)
        kw = dict(self._ddp_kw, find_unused_parameters=self.find_unused_parameters,
                  broadcast_buffers=self._needs_buffer_broadcast())
        if self.device.type == "cuda":
            self.net = DDP(self.model, device_ids=[self.device.index], **kw)
        else:
            self.net = DDP(self.model, **kw)
        compress = self.grad_compress
        if compress == "bf16_fp32acc":
            self.net.register_comm_hook(state=None, hook=dist_utils.bf16_wire_fp32_acc_hook)
        elif compress == "bf16":
            from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
            self.net.register_comm_hook(state=None, hook=default_hooks.bf16_compress_hook)
        elif compress is not None:
            raise ValueError(f"grad_compress={compress!r}")
        self._want_ddp = False

    def _freeze_never_used(self, data_dict) -> None:
        """One local forward + backward on the first batch; a trainable tensor is frozen only if NO rank gave it a
        gradient (the per-rank masks are MAX-all-reduced, so every rank builds DDP over the same parameter list even
        when a data-dependent branch fired on some ranks only).  Buffers (BatchNorm running statistics of an unfrozen
        encoder) and the RNG streams are restored: the probe leaves no trace in the training state.  A tensor that is
        unused in the probe on every rank but used by a later batch would stay untrained -- configurations with such
        branches should pass find_unused_parameters=True (the reference's setting)."""
        named = dict(self.model.named_parameters())
        names = [n for n, p in named.items() if p.requires_grad]
        for n in names:
            named[n].grad = None
        buffers = [(b, b.detach().clone()) for b in self.model.buffers()]
        rng_cpu = torch.get_rng_state()
        rng_dev = torch.cuda.get_rng_state(self.device) if self.device.type == "cuda" else None
        seed_state = None
        if self.device.type == "cuda":                     # the device-side dropout seed block of the native kernels
            from .modules.layers import fused_attention
            seed_state = fused_attention.snapshot_seed_state(self.device)
        with self._autocast():
            out = self.model(dict(data_dict))
            total, _ = self.loss(out)
        total.backward()
        got = torch.tensor([1 if named[n].grad is not None else 0 for n in names], dtype=torch.int32, device=self.device)
        if dist_utils.is_dist() and dist_utils.get_world_size() > 1:
            torch.distributed.all_reduce(got, op=torch.distributed.ReduceOp.MAX)
        got = got.tolist()
        self.frozen_unused = sorted(n for n, g in zip(names, got) if not g)
        for n in self.frozen_unused:
            named[n].requires_grad_(False)
        for n in names:
            named[n].grad = None
        with torch.no_grad():
            for b, saved in buffers:
                b.copy_(saved)
        torch.set_rng_state(rng_cpu)                       # the probe must not shift the training RNG streams
        if rng_dev is not None:
            torch.cuda.set_rng_state(rng_dev, self.device)
        if seed_state is not None:
            fused_attention.restore_seed_state(self.device, seed_state)
        if self.frozen_unused and dist_utils.get_rank() == 0:
            import logging
            logging.getLogger("sceneverse_amd").warning(
                "DDP: %d trainable tensors receive no gradient on any rank in the probe step and were FROZEN "
                "(requires_grad=False stays on the model; pass find_unused_parameters=True for the reference's "
                "setting): %s", len(self.frozen_unused), ", ".join(self.frozen_unused))

    # ---- split-graph data parallelism ------------------------------------------------------------
    def _dist_losses(self):
        return [m for m in self.loss.modules() if hasattr(m, "gather_inputs") and getattr(m, "distributed", False)]

    def _gather_features(self, out):
        """Eager C2: all-gather what the between-batch losses need into their static buffers."""
        for m in self._dist_losses():
            with torch.no_grad(), self._autocast():
                ins = [t.detach().contiguous() for t in m.gather_inputs(out)]
            if m._gathered is None:
                m._gathered = [torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype,
                                           device=t.device) for t in ins]
            for buf, t in zip(m._gathered, ins):
                if dist_utils.is_dist() and self.world > 1:
                    torch.distributed.all_gather_into_tensor(buf, t)
                else:
                    buf.copy_(t)

    def _allreduce_grads(self):
        """Eager C1: all-reduce of the flat gradient buffer, then the mean."""
        self._wait_allreduce(self._allreduce_async(0, self._flat_grad.numel()))

    def _allreduce_async(self, lo: int, hi: int):
        """Start the all-reduce of `_flat_grad[lo:hi]` (the gradients of one backward segment) and return a handle for
        `_wait_allreduce`.  With RCCL the collective runs on the process group's own stream, ordered after what the
        current stream has issued so far -- i.e. beside the next backward segment, which is launched right after."""
        if not (dist_utils.is_dist() and self.world > 1) or hi <= lo:
            return None
        seg = self._flat_grad[lo:hi]
        return (torch.distributed.all_reduce(seg, async_op=True), seg)

    def _wait_allreduce(self, *handles) -> None:
        for h in handles:
            if h is None:
                continue
            work, seg = h
            work.wait()                                  # the current stream waits; the host does not
            seg.mul_(1.0 / self.world)

    def _clip_and_step(self):
        from .optim.fused_adamw import GpsAdamW
        if isinstance(self.optimizer, GpsAdamW):       # clipping happens inside the optimizer pass
            self.optimizer.step(max_grad_norm=self.grad_norm)
            return
        if self.grad_norm is not None:
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.grad_norm)
        self.optimizer.step()

    def _graph_dp_step(self, data_dict):
        tensors = {k: v for k, v in data_dict.items() if torch.is_tensor(v)}
        cur = torch.cuda.current_stream(self.device)
        if self._graph is None and self.global_step < self.graph_warmup:
            # eager warm-up with the same exchange points (lazy inits, hipBLASLt heuristics, ...)
            side = self._stream()
            side.wait_stream(cur)
            if self._acc_hooks is None:
                # which parameters does AUTOGRAD accumulate into during the warm-up steps?  Those must keep their zero
                # fill even if a grouped launch also writes them (a tied weight, a module also used through F.linear):
                # the "plain store" shortcut below is for parameters whose ONLY writer is the grouped launch
                self._acc_hooks = [p.register_post_accumulate_grad_hook(lambda q: self._autograd_written.add(id(q)))
                                   for p in self.model.parameters() if p.requires_grad]
            with torch.cuda.stream(side):
                self._begin_step()
                with self._autocast():
                    out = self.net(data_dict)
                self._gather_features(out)
                with self._autocast():
                    total, losses = self.loss(out)
                self.optimizer.zero_grad(set_to_none=True)
                self._backward(total)
                grads = [p.grad for p in self.model.parameters() if p.grad is not None]
                if dist_utils.is_dist() and self.world > 1:
                    flat = torch.cat([g.reshape(-1).float() for g in grads])
                    torch.distributed.all_reduce(flat)
                    flat.mul_(1.0 / self.world)
                    off = 0
                    for g in grads:
                        g.copy_(flat[off:off + g.numel()].view_as(g))
                        off += g.numel()
                self._clip_and_step()
            cur.wait_stream(side)
            return total.detach(), {k: v.detach() for k, v in losses.items()}
        if self._graph is None:
            self._static = {k: v.clone() for k, v in tensors.items()}
            static_dict = dict(data_dict)
            static_dict.update(self._static)
            # parameters whose gradient the grouped weight-gradient launch writes (learnt in the warm-up steps): their
            # part of the flat buffer is never zero-filled, the launch stores instead of adding (gemm.mark_stale_grads)
            from .modules.layers import gemm as _gemm
            direct_ids = (_gemm.grouped_written_ids() - self._autograd_written) if self.wgrad_group else set()
            for h in self._acc_hooks or []:
                h.remove()
            self._acc_hooks = []
            # gradients live as views of ONE flat fp32 buffer (only for parameters that do receive a
            # gradient: the never-used ones keep grad None, as under DDP / eager AdamW)
            used = [p for p in self.model.parameters() if p.grad is not None]
            # [r3] two backward segments: "top" = joint encoder, heads, loss parameters; "bottom" = the text and the
            # object encoder.  Their gradients sit in two contiguous ranges of the flat buffer, so the all-reduce of
            # the top range travels over xGMI while the bottom segment's backward graph replays.
            # [r4] the bottom segment itself is two independent autograd sub-graphs (text encoder, object encoder): each
            # gets its own backward graph and its own range, so the exchange is top | text | objects with only the LAST
            # (smallest) range exposed: the text range (BERT incl. its 94 MB word table) travels beside the object
            # encoder's backward.
            seg_of = {}
            for si, name in enumerate(("lang_encoder", "point_encoder")):
                sub = getattr(self.model, name, None)
                if sub is not None:
                    seg_of.update({id(p): si for p in sub.parameters()})
            by_kind = lambda ps: [p for p in ps if id(p) not in direct_ids] + [p for p in ps if id(p) in direct_ids]  # noqa: E731
            top = by_kind([p for p in used if id(p) not in seg_of])
            bottom_segs = [by_kind([p for p in used if seg_of.get(id(p)) == si]) for si in range(2)]
            bottom = bottom_segs[0] + bottom_segs[1]
            used = top + bottom
            # every view starts on a 16-byte boundary (kernels store gradients as 4-float vectors): sizes are rounded up
            # to 4 elements, the padding stays zero and travels with the all-reduce
            pad4 = lambda n: (n + 3) // 4 * 4  # noqa: E731
            self._flat_grad = torch.zeros(sum(pad4(p.numel()) for p in used), dtype=torch.float32, device=self.device)
            self._n_top = sum(pad4(p.numel()) for p in top)
            self._seg_ends = [self._n_top]                          # end offsets of the ranges: top, text, objects
            for seg in bottom_segs:
                self._seg_ends.append(self._seg_ends[-1] + sum(pad4(p.numel()) for p in seg))
            # inside every range: [accumulated by autograd: zero-filled each step | written by the grouped launch]
            zero_ranges, lo = [], 0
            for seg in (top, bottom_segs[0], bottom_segs[1]):
                n_acc = sum(pad4(p.numel()) for p in seg if id(p) not in direct_ids)
                if n_acc:
                    zero_ranges.append((lo, lo + n_acc))
                lo += sum(pad4(p.numel()) for p in seg)
            direct_params = [p for p in used if id(p) in direct_ids]
            off = 0
            self.optimizer.zero_grad(set_to_none=True)
            for p in used:
                p.grad = self._flat_grad[off:off + p.numel()].view_as(p)
                off += pad4(p.numel())
            torch.cuda.synchronize(self.device)
            g1, g2a, g3 = (self._new_graph() for _ in range(3))
            # thread-local capture mode: RCCL's watchdog thread polls its events (hipEventQuery) while we capture; in the
            # default global mode any such call from another thread invalidates the capture
            ws = self._stream()
            with torch.cuda.graph(g1, stream=ws, capture_error_mode=_CAPTURE_MODE):
                self._begin_step()
                with self._autocast():
                    out = self.net(static_dict)
            self._stage("captured_g1", out=out)
            self._gather_features(out)
            boundary = list(getattr(self.model, "_stage_boundary", None) or [])
            segmented = bool(boundary) and bool(top) and bool(bottom) and not self.wgrad_overlap
            torch.cuda.synchronize(self.device)
            with torch.cuda.graph(g2a, pool=g1.pool(), stream=ws, capture_error_mode=_CAPTURE_MODE):
                for lo, hi in zero_ranges:
                    self._flat_grad[lo:hi].zero_()
                _gemm.mark_stale_grads(direct_params)
                with self._autocast():
                    total, losses = self.loss(out)
                if segmented and not _cut_is_valid(total, boundary, bottom):
                    # a gradient path from the loss into a bottom-segment parameter that does not cross a boundary
                    # tensor (a head reading an encoder-internal tensor, a tied weight ...): the staged backward would
                    # silently drop it -- keep the backward pass in one piece instead
                    import logging
                    logging.getLogger("sceneverse_amd").warning(
                        "split-graph step: the stage boundary does not separate the text / object encoders from the "
                        "loss; the backward pass stays ONE graph (no overlap of the gradient exchange)")
                    segmented = False
                if segmented:
                    # gradients of the top parameters (into their flat views) and of the boundary tensors
                    with self._wgrad_ctx(only=top):
                        torch.autograd.backward(total, inputs=top + boundary, retain_graph=True)
                else:
                    self._backward(total)             # accumulates into the flat views
            torch.cuda.synchronize(self.device)
            self._stage("captured_g2a", out=out, total=total)
            if segmented:
                live = [t for t in boundary if t.grad is not None]
                groups = self._bottom_groups(live, boundary, bottom_segs)
                self._bottom_graphs_per_range = len(groups) == 2
                g2b = []
                for grp in groups:
                    gg = self._new_graph()
                    with torch.cuda.graph(gg, pool=g1.pool(), stream=ws, capture_error_mode=_CAPTURE_MODE), self._wgrad_ctx(only=bottom):
                        torch.autograd.backward(grp, grad_tensors=[t.grad for t in grp], inputs=bottom)
                    torch.cuda.synchronize(self.device)
                    g2b.append(gg)
                self._stage("captured_g2b")
            else:
                g2b = None
            # a buffer marked "written by the grouped launch" that no launch of this capture wrote would keep last step's
            # values for ever (cannot happen while warm-up and capture run the same step: fail loudly if it does)
            left = _gemm.stale_grads_left()
            if left:
                raise RuntimeError(f"split-graph step: {len(left)} gradient buffers marked for direct stores were not "
                                   "written during capture (the warm-up steps and the captured step differ)")
            with torch.cuda.graph(g3, stream=ws, capture_error_mode=_CAPTURE_MODE):
                self._clip_and_step()
            self._drop_previous_graph()
            # only the VALUES of the outputs are read after a replay: no reference to the captured autograd graph is kept
            self._graph, self._graph_out = (g1, g2a, g2b, g3), (
                {k: (v.detach() if torch.is_tensor(v) else v) for k, v in out.items()}, total.detach(),
                {k: v.detach() for k, v in losses.items()})
        else:
            self._fill_static(tensors)
        g1, g2a, g2b, g3 = self._graph
        out, total, losses = self._graph_out
        g1.replay()
        self._stage("replayed_g1")
        self._gather_features(out)
        self._stage("replayed_gather")
        g2a.replay()
        self._stage("replayed_g2a")
        ev = self._exchange_events
        if g2b is not None:
            handles = [self._allreduce_async(0, self._n_top)]          # overlaps the bottom segments' backward graphs
            for gi, gg in enumerate(g2b):
                gg.replay()
                self._stage("replayed_g2b" if gi + 1 == len(g2b) else f"replayed_g2b_part{gi}")
                if getattr(self, "_bottom_graphs_per_range", False) and gi + 1 < len(g2b):
                    handles.append(self._allreduce_async(self._seg_ends[gi], self._seg_ends[gi + 1]))
            done = self._seg_ends[len(handles) - 1]                    # ranges already on their way
            if ev is not None:
                ev[0].record()                                         # the last backward graph ends here
            handles.append(self._allreduce_async(done, self._flat_grad.numel()))
            self._wait_allreduce(*handles)
        else:
            if ev is not None:
                ev[0].record()
            self._allreduce_grads()
        if ev is not None:
            ev[1].record()                                             # every range reduced: clip + AdamW may start
        g3.replay()
        return total.detach().clone(), {k: v.detach().clone() for k, v in losses.items()}

    def time_exchange(self, on: bool = True) -> None:
        """Split-graph step: record an event pair around the part of the gradient exchange nothing overlaps (from the end
        of the last backward graph to the moment every range is reduced); `exposed_allreduce_ms()` reads the last step's."""
        self._exchange_events = ((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                                 if on and self.device.type == "cuda" else None)

    def exposed_allreduce_ms(self):
        ev = self._exchange_events
        if ev is None or self._graph is None or not self.graph_dp:
            return None
        ev[1].synchronize()
        return float(ev[0].elapsed_time(ev[1]))

    def _new_graph(self):
        """Every graph of the captured steps is made here (tests subclass this to keep the hipGraph_t for inspection)."""
        return torch.cuda.CUDAGraph()

    @staticmethod
    def _bottom_groups(live, boundary, bottom_segs):
        """Boundary tensors by the encoder that produced them (OpenVocab lists the text outputs first, the object
        encoder's output last): one backward graph per encoder, text first, so that the text range of the flat gradient
        buffer travels beside the object encoder's backward.  One group when either encoder has nothing to train."""
        groups = [[t for t in live if t is not boundary[-1]], [t for t in live if t is boundary[-1]]]
        return groups if all(groups) and all(bottom_segs) else [live]

    def _stream(self):
        if self._work_stream is None:
            self._work_stream = torch.cuda.Stream(device=self.device)
        return self._work_stream

    @contextlib.contextmanager
    def _strict_accumulate_grad(self):
        """Inside: torch's "AccumulateGrad node's stream does not match" warning is an exception (a captured backward
        must be a linear chain of nodes, see `_work_stream`).  The filter lives for the duration of this engine's own
        warm-up / capture / replay calls only -- user code and other engines keep the process's warning state."""
        import warnings
        with warnings.catch_warnings():
            warnings.filterwarnings("error", message=".*AccumulateGrad node's stream does not match.*")
            yield

    def _rebind_lr(self) -> None:
        """Graph modes: every param group's `lr` must BE a view of `_lr_dev` (the word the captured AdamW reads and the
        scheduler writes).  `Optimizer.load_state_dict` replaces it with a copy carrying the checkpoint's value: adopt
        the value, restore the alias.  Runs before every step that is not a replay (a load after capture is refused by
        GpsAdamW itself)."""
        if self._lr_dev is None:
            return
        for gi, g in enumerate(self.optimizer.param_groups):
            lr, word = g["lr"], self._lr_dev[gi]
            if torch.is_tensor(lr) and lr.device == word.device and lr.data_ptr() == word.data_ptr():
                continue
            with torch.no_grad():
                word.fill_(float(lr))
            g["lr"] = word
            if hasattr(self.optimizer, "_sig"):
                self.optimizer._sig = None               # tables hold the learning-rate words' addresses

    def _sched_step(self) -> None:
        """`scheduler.step()`; with device-resident learning rates (graph modes) and a LambdaLR: the same bookkeeping and
        values, written to the device with one copy."""
        from torch.optim.lr_scheduler import LambdaLR
        sch = self.scheduler
        if self._lr_dev is None or type(sch) is not LambdaLR or len(sch.base_lrs) != self._lr_dev.numel():
            sch.step()
            return
        sch._step_count += 1
        sch.last_epoch += 1
        vals = [float(base) * float(fn(sch.last_epoch)) for fn, base in zip(sch.lr_lambdas, sch.base_lrs)]
        slot = self._lr_host[sch._step_count % self._lr_host.shape[0]]
        for i, v in enumerate(vals):
            slot[i] = v
        self._lr_dev.copy_(slot, non_blocking=True)
        sch._last_lr = vals

    def _drop_previous_graph(self) -> None:
        """Nothing of the previous step's autograd graph may survive into this step's forward (see _work_stream): the
        stage boundary the model publishes is the one reference this package keeps."""
        if getattr(self.model, "_stage_boundary", None) is not None:
            self.model._stage_boundary = None

    def _stage(self, name: str, **kw) -> None:
        if self.stage_hook is not None:
            self.stage_hook(name, self, **kw)

    def _autocast(self):
        """Context of every model / loss evaluation of the step: bf16 autocast, and (train mode, native GEMMs) the
        masked-LM head in its fused linear + cross-entropy form (modules/heads/pretrain_head.py)."""
        if self.amp_dtype is None:
            return contextlib.nullcontext()
        stack = contextlib.ExitStack()
        stack.enter_context(torch.autocast(device_type="cuda", dtype=self.amp_dtype))
        if self.fused_lm_loss:
            from .modules.heads.pretrain_head import fused_lm_loss
            stack.enter_context(fused_lm_loss(True))
        return stack

    def _backward(self, total) -> None:
        """`total.backward()`; on a single GPU (or the split-graph form) with the weight gradients of the native
        Linears overlapped on a side stream and joined before anything reads `param.grad`."""
        from torch.nn.parallel import DistributedDataParallel as DDP
        if self.wgrad_overlap and not isinstance(self.net, DDP) and not self._want_ddp:
            from .modules.layers.gemm import deferred_wgrads
            with deferred_wgrads():
                total.backward()
        else:
            with self._wgrad_ctx():
                total.backward()

    def _wgrad_ctx(self, only=None):
        """Grouped weight gradients around a backward call (a no-op context under torch DDP or when switched off).
        only: the parameters of a backward pass restricted with `inputs=` (the deferred writes must honour it too)."""
        from torch.nn.parallel import DistributedDataParallel as DDP
        from .modules.layers.gemm import grouped_wgrads
        return grouped_wgrads(self.wgrad_group and not isinstance(self.net, DDP) and not self._want_ddp, only=only)

    def _begin_step(self):
        # one tiny launch that advances the device-side dropout seed block; it sits inside every captured
        # region that runs the model, so each graph replay draws fresh masks
        if torch.device(self.device).type == "cuda":
            from .modules.layers import fused_attention
            fused_attention.begin_step(self.device)

    def forward_loss(self, data_dict):
        self._begin_step()
        with self._autocast():
            out = self.net(data_dict)
            total, losses = self.loss(out)
        return out, total, losses

    def _eager_body(self, data_dict):
        out, total, losses = self.forward_loss(data_dict)
        self.optimizer.zero_grad(set_to_none=True)
        self._backward(total)
        self._clip_and_step()
        return total, losses

    def _graph_step(self, data_dict):
        """Warm-up steps run eagerly on a side stream (torch's capture protocol), then the step is
        captured once; afterwards: copy the batch into the static buffers, replay."""
        tensors = {k: v for k, v in data_dict.items() if torch.is_tensor(v)}
        if self._graph is None and self.global_step < self.graph_warmup:
            side = self._stream()
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                total, losses = self._eager_body(data_dict)
            torch.cuda.current_stream(self.device).wait_stream(side)
            return total.detach(), {k: v.detach() for k, v in losses.items()}
        if self._graph is None:
            self._static = {k: v.clone() for k, v in tensors.items()}
            static_dict = dict(data_dict)
            static_dict.update(self._static)
            self.optimizer.zero_grad(set_to_none=True)
            torch.cuda.synchronize(self.device)
            g = self._new_graph()
            with torch.cuda.graph(g, stream=self._stream()):
                total, losses = self._eager_body(static_dict)
            self._drop_previous_graph()
            self._graph, self._graph_out = g, (total.detach(), {k: v.detach() for k, v in losses.items()})
        else:
            self._fill_static(tensors)
        self._graph.replay()
        total, losses = self._graph_out
        return total.detach().clone(), {k: v.detach().clone() for k, v in losses.items()}

    def static_inputs(self):
        """The batch buffers the captured graph reads ({key: tensor}), or None before capture.  A loader that writes a
        batch straight into them (gps_obj_processing_post takes an output pointer) and passes them to `step` saves the
        per-step copy of the batch (126 MB of object points at B = 64): `_fill_static` skips identical storage."""
        return None if self._graph is None else dict(self._static)

    def _fill_static(self, tensors) -> None:
        """Copy a batch into the buffers the captured graph reads.  The graph was captured for ONE set of keys,
        shapes and dtypes: anything else (a smaller last batch, a missing or extra tensor) would replay stale or
        broadcast data, so it is an error here, not a silent copy."""
        if tensors.keys() != self._static.keys():
            raise ValueError("HIP-graph step: the batch has tensor keys "
                             f"{sorted(tensors.keys() ^ self._static.keys())} that differ from the captured batch")
        for k, v in tensors.items():
            buf = self._static[k]
            if v.shape != buf.shape or v.dtype != buf.dtype:
                raise ValueError(f"HIP-graph step: batch['{k}'] is {tuple(v.shape)} {v.dtype}, the graph was captured "
                                 f"for {tuple(buf.shape)} {buf.dtype} (drop the last partial batch or use graph=False)")
            if buf.data_ptr() != v.data_ptr():
                buf.copy_(v, non_blocking=True)

    def step(self, data_dict):
        """One optimisation step; returns (total_loss tensor, dict of loss tensors).  No host sync."""
        self.net.train()
        self._drop_previous_graph()
        # The text and the object stack run in lock-step with paired GEMM launches (modules/layers/gemm.py drive_pair).  The
        # split-graph step cuts the backward pass BETWEEN the two stacks (two graphs, `_bottom_groups`): there the pairs
        # share their forward launches only and keep one autograd node per stack.
        from .modules.layers import gemm as _gemm_mode
        _gemm_mode.set_twin_backward(not self.graph_dp)
        if self.graph or self.graph_dp:
            data_dict['cur_step'] = 0
            data_dict['total_steps'] = 1 << 30
            if self._graph is None:
                self._rebind_lr()
            # (pure replays run no autograd: the warning filter -- process-global state -- is only touched by the eager
            # warm-up steps and the capture)
            with (self._strict_accumulate_grad() if self._graph is None else contextlib.nullcontext()):
                total, losses = self._graph_dp_step(data_dict) if self.graph_dp else self._graph_step(data_dict)
            self._sched_step()
            self.global_step += 1
            return total, losses
        data_dict['cur_step'] = self.global_step
        data_dict['total_steps'] = 1 << 30
        if self._want_ddp:
            self._build_ddp(data_dict)
        if self._graph is not None:
            # an eager step after graph replays (bench's timing pass): replays update the masters without
            # touching their Python-side version counters, so the bf16 shadows must be rebuilt from them
            from .modules.layers import gemm as _gemm
            _gemm.invalidate_shadows()
        # an eager step of an engine that has captured graphs runs on the stream the captures used (see _work_stream)
        ctx, cur = contextlib.nullcontext(), None
        if self._graph is not None and self.device.type == "cuda":
            cur = torch.cuda.current_stream(self.device)
            self._stream().wait_stream(cur)
            ctx = torch.cuda.stream(self._stream())
        with ctx:
            out, total, losses = self.forward_loss(data_dict)
            self.optimizer.zero_grad(set_to_none=True)
            self._backward(total)
            self._clip_and_step()
        if cur is not None:
            cur.wait_stream(self._stream())
        self._sched_step()
        self.global_step += 1
        return total.detach(), {k: v.detach() for k, v in losses.items()}

    @torch.no_grad()
    def evaluate(self, data_dict):
        self.net.eval()
        out, total, losses = self.forward_loss(data_dict)
        return out, total, losses


def _cut_is_valid(total: torch.Tensor, boundary, bottom) -> bool:
    """True iff every autograd path from `total` to a parameter of `bottom` passes through one of the `boundary`
    tensors, i.e. backward(total, inputs=top + boundary) followed by backward(boundary, inputs=bottom) computes the
    same gradients as one backward pass.  Walks the graph above the boundary once (at capture time)."""
    bottom_ids = {id(p) for p in bottom}
    blocked = {(t.grad_fn, t.output_nr) for t in boundary if t.grad_fn is not None}
    keep = [fn for fn, _ in blocked]              # node wrappers stay alive: their identity is what `blocked` compares
    if total.grad_fn is None:
        return True
    seen, stack = set(), [total.grad_fn]
    while stack:
        fn = stack.pop()
        if id(fn) in seen:
            continue
        seen.add(id(fn))
        keep.append(fn)
        var = getattr(fn, "variable", None)       # AccumulateGrad: a leaf
        if var is not None and id(var) in bottom_ids:
            return False
        for nxt, idx in fn.next_functions:
            if nxt is not None and (nxt, idx) not in blocked:
                stack.append(nxt)
    return True


def scanrefer_accuracy(og3d_logits: torch.Tensor, iou25_onehot: torch.Tensor,
                       iou50_onehot: torch.Tensor) -> dict:
    """acc@0.25 / acc@0.5 exactly as the reference's ScanReferEval.batch_metrics computes them
    (evaluator/scanrefer_eval.py:14-87): the arg-max object counts as correct when its one-hot
    entry in `tgt_object_id_iou25/50` is set."""
    pred = torch.argmax(og3d_logits, dim=-1)
    pick = pred[:, None]
    hit25 = torch.gather(iou25_onehot, 1, pick).squeeze(1).bool()
    hit50 = torch.gather(iou50_onehot, 1, pick).squeeze(1).bool()
    n = float(max(1, pred.shape[0]))
    return {"og_acc_iou25": hit25.sum().item() / n, "og_acc_iou50": hit50.sum().item() / n}


def dp_self_check(make_engine, batches, rtol_loss: float = 2e-3, rtol_grad: float = 3e-2) -> dict:
    """Start-up check of a data-parallel engine against torch DDP in eager mode, before anything is timed (bench.py
    --gpus N): the first real contact of the split-graph step with an RCCL ring must fail LOUDLY, not hang or drift.

    make_engine(eager_ddp: bool) -> GPSTrainStep built from the SAME seed with every dropout probability zeroed (so the
    two runs are comparable step for step); batches: this rank's batches, enough to take the candidate through its eager
    warm-up steps, its capture and a first replay.  Both engines run them in order.  Checked on every rank, reduced
    over ranks:
      * the losses of every step agree (candidate vs eager DDP) to rtol_loss,
      * the gradients the last step applied agree: sum of squares and sum of magnitudes over every trainable tensor
        (the flat-gradient checksum) to rtol_grad -- a range that was not reduced carries this rank's own gradient
        instead of the mean, whose norm differs by tens of percent; bf16 noise moves the checksum by a few 1e-3,
      * every rank holds the same parameters (max - min of the per-rank checksums is exactly 0 for the candidate:
        identical initial weights + identical averaged gradients -- the invariant of data parallelism; a rank that
        applied its own, un-reduced gradient breaks it).
    -> {"ok": bool, "loss_rel_diff", "grad_checksum_rel_diff", "cross_rank_spread", "steps", "reason"}; collective: every
    rank must call it."""
    import torch.distributed as dist
    world = dist_utils.get_world_size()

    def run(eager_ddp):
        eng = make_engine(eager_ddp)
        losses = []
        for b in batches:
            loss, _ = eng.step(dict(b))
            losses.append(float(loss))
        with torch.no_grad():
            ps = [p.detach().double() for p in eng.model.parameters() if p.requires_grad]
            gs = [p.grad.detach().double() for p in eng.model.parameters() if p.requires_grad and p.grad is not None]
            chk = torch.stack([sum((p * p).sum() for p in ps), sum(p.abs().sum() for p in ps)])
            gchk = torch.stack([sum((g * g).sum() for g in gs), sum(g.abs().sum() for g in gs)])
        replayed = getattr(eng, "_graph", None) is not None
        del eng
        return losses, chk, gchk, replayed

    cand_losses, cand_chk, cand_g, replayed = run(False)
    ref_losses, _, ref_g, _ = run(True)
    dev = cand_chk.device
    loss_diff = max(abs(a - b) / max(1.0, abs(b)) for a, b in zip(cand_losses, ref_losses))
    sum_diff = float(((cand_g - ref_g).abs() / ref_g.abs().clamp_min(1e-30)).max())
    stats = torch.tensor([loss_diff, sum_diff, 0.0], dtype=torch.float64, device=dev)
    spread = 0.0
    if dist_utils.is_dist() and world > 1:
        lo, hi = cand_chk.clone(), cand_chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        spread = float(((hi - lo).abs() / hi.abs().clamp_min(1e-30)).max())
        stats[2] = spread
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
    loss_diff, sum_diff, spread = (float(x) for x in stats.tolist())
    reasons = []
    if not all(map(lambda v: v == v and abs(v) != float("inf"), cand_losses)):
        reasons.append("non-finite loss")
    if loss_diff > rtol_loss:
        reasons.append(f"losses differ from eager DDP by {loss_diff:.2e}")
    if sum_diff > rtol_grad:
        reasons.append(f"gradient checksums differ from eager DDP by {sum_diff:.2e}")
    if spread > 0.0:
        reasons.append(f"ranks hold different parameters (spread {spread:.2e})")
    return {"ok": not reasons, "loss_rel_diff": loss_diff, "grad_checksum_rel_diff": sum_diff, "cross_rank_spread": spread,
            "steps": len(batches), "replayed_a_graph": bool(replayed), "reason": "; ".join(reasons) or None}
