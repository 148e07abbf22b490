"""Device-side train steps: the whole of `Pix2Pix.train_step` (pix2pix.py:190-218) and
`CycleGAN.train_step` (cycle_gan.py:206-276) as sequences of C-ABI kernel launches — forward, losses,
explicit backward and TF-form Adam all on the GPU, optionally captured once into a hipGraph and replayed.

Semantics kept from the reference: both gradients are taken at the pre-update weights from a single
forward (simultaneous update); D(real) and D(fake) are separate BatchNormalization invocations (separate
batch statistics, two moving-average updates); dropout is active in every generator call; `training=False`
computes forward + losses only (pix2pix.py:208, :291-292).
"""
from __future__ import annotations

import contextlib
import ctypes as C

import torch

from . import _lib as L
from .nets import Ctx, DiscriminatorNet, GeneratorNet


# Other threads of the process may call into the runtime while a step is being captured (the RCCL watchdog of
# torch.distributed polls its events): only calls of the CAPTURING thread may invalidate the capture.
CAPTURE_MODE = "thread_local"


class _StepBase:
    def _bce(self, logits_ptr, count, target, loss_idx, loss_scale, acc, grad_scale, dx_ptr, ws=None):
        lib, ctx = self.ctx.lib, self.ctx
        rc = lib.gan_bce_logits(logits_ptr, count, target, loss_scale, int(acc), self.losses.data_ptr() + 4 * loss_idx,
                                grad_scale, ctx.dt, dx_ptr, 8, (ws if ws is not None else self.bce_ws).data_ptr(), ctx.ls_ptr, ctx.stream())
        L.check(rc, "bce_logits")

    def _l1(self, a, b, loss_idx, loss_scale, acc, grad_scale, da, stream=None, ws=None):
        lib, ctx = self.ctx.lib, self.ctx
        rc = lib.gan_l1(ctx.dt, C.byref(a), C.byref(b), loss_scale, int(acc), self.losses.data_ptr() + 4 * loss_idx,
                        grad_scale, C.byref(da) if da is not None else None, (ws if ws is not None else self.l1_ws).data_ptr(), ctx.ls_ptr,
                        stream.cuda_stream if stream is not None else ctx.stream())
        L.check(rc, "l1")

    def _pack(self, src_f32, dst_view):
        L.check(self.ctx.lib.gan_pack(self.ctx.dt, src_f32.data_ptr(), C.byref(dst_view), self.ctx.stream()), "pack")

    def _pack_multi(self, pairs):
        """[(src_f32, dst_view), ...] (<= 4, one shape) in a single launch."""
        n = len(pairs)
        srcs = (C.c_void_p * n)(*[s_.data_ptr() for s_, _ in pairs])
        dsts = (L.GanTensor * n)(*[d_ for _, d_ in pairs])
        L.check(self.ctx.lib.gan_pack_multi(self.ctx.dt, n, srcs, dsts, self.ctx.stream()), "pack_multi")

    def _copy(self, src_view, dst_view):
        L.check(self.ctx.lib.gan_copy_view(self.ctx.dt, C.byref(src_view), C.byref(dst_view), self.ctx.stream()), "copy_view")

    def _run(self, a, b, training=True):
        self._updating = training          # a full step: parts of the update may be scheduled inside the backward pass
        self._forward_backward(a, b, training)
        self._updating = False
        if training:
            if self.sync is not None:
                self.sync(unpack=not self._wire_adam())    # data-parallel gradient exchange (RCCL)
            self._update()
        return self.losses

    def _wire_adam(self):
        """Data-parallel, bf16 wire: Adam reads the exchanged gradient from the wire buffers (no unpack pass).  Not with fp16
        loss scaling: its inf/nan check must see the EXCHANGED gradients (every rank takes the same skip decision), which
        the unpacked fp32 buffers hold."""
        s_ = self.sync
        return s_ is not None and getattr(s_, 'compress', False) and getattr(s_, 'wire', None) is not None and self.ctx.ls is None

    def _update(self):
        gs = self.sync.grad_scale if self.sync is not None else 1.0
        wire = self._wire_adam()
        if wire:
            gs = 1.0 / self.sync.world
        wp = lambda net: (self.sync.wire[self.nets().index(net)].data_ptr() if wire else None)
        early, done = getattr(self, '_early_adam', None), getattr(self, '_adam_done', ())
        ctx = self.ctx
        if ctx.ls is not None:        # fp16: a non-finite gradient anywhere skips the whole step (every network), then the scale adapts
            for net in self.nets():
                ctx.run(net.params.grads_check_ops())
        wfused = getattr(self, '_adam_wfused', None) or {}
        for net in self.nets():
            if net in done:        # updated at the end of its own backward chain
                continue
            if net in wfused:      # its big kernels were updated by their own wgrad launches (GanAdamFuse): the rest + the vectors
                ctx.run(net.params.adam_rest_ops(wfused[net], self.b1, self.b2))
                continue
            if net is early:       # some kernel segments were updated beside the backward pass: the rest + the vectors
                rest = [k for k in range(len(net.params._segments)) if k not in self._early_segs]
                for j, k in enumerate(rest):
                    self.ctx.run(net.params.adam_segment_ops(k, self.b1, self.b2, grad_scale=gs, vectors=(j == len(rest) - 1), wire_ptr=wp(net)))
                if not rest:
                    self.ctx.run(net.params.adam_segment_ops(0, self.b1, self.b2, grad_scale=gs, vectors=True, kernels=False, wire_ptr=wp(net)))
            else:
                net.params.adam(self.lr, self.b1, self.b2, grad_scale=gs, wire_ptr=wp(net))
        if ctx.ls is not None:
            L.check(ctx.lib.gan_loss_scale_update(ctx.ls_ptr, ctx.ls_growth_interval, ctx.ls_max, ctx.stream()), "loss_scale_update")
        self._early_adam, self._adam_done, self._adam_wfused = None, (), None

    def _wgrad_adam_ok(self):
        """GanAdamFuse schedules: one GPU, no loss scaling, 16-bit storage."""
        return bool(getattr(self, 'fused_wgrad_adam', False) and self.sync is None and self.ctx.ls is None and self.ctx.dtype != 'f32')

    def _prebuild_fused_adam(self):
        pass

    # ---- hipGraph capture of a whole step --------------------------------------------------------
    def capture(self, training=True):
        """Capture one full step on static input buffers; returns a callable replaying it.  With a gradient
        exchange attached the step becomes graph(forward+backward) -> collectives -> graph(Adam)."""
        self._static_in = [torch.zeros_like(t) for t in self._example_inputs()]
        torch.cuda.synchronize()
        s = torch.cuda.Stream(device=self.ctx.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):           # warm-up outside capture (lazy inits, func attributes)
            self._run(*self._static_in, training=training)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        split = training and self.sync is not None and getattr(self.sync, 'active', self.sync.world > 1)
        if split and hasattr(self, '_capture_bucketed') and self.ctx.lanes and self.ctx.ls is None and self.ddp_buckets:
            return self._capture_bucketed()          # (fp16: the whole-step inf/nan check precedes every Adam -> phased schedule)
        if split:
            return self._capture_phased(training)
        if training:
            self._prebuild_fused_adam()         # (op lists and device tables of the captured schedule: no uploads inside the capture)
        def body():
            self._capturing = True          # (schedules that leave no fp32 gradients behind are for the replayed step only)
            try:
                self._run(*self._static_in, training=training)
            finally:
                self._capturing = False
        g1 = self.ctx.capture_graph(body, CAPTURE_MODE)      # (an exception inside leaves no forked lane behind: Ctx.capture_graph)
        self._graphs = (g1, None, None)

        def replay(*inputs):
            for dst, src in zip(self._static_in, inputs):
                if src is not dst:
                    dst.copy_(src, non_blocking=True)
            g1.replay()
            return self.losses
        replay.inputs = self._static_in      # a caller that fills these buffers itself (and passes them back) skips the copies
        return replay

    def _capture_phased(self, training):
        """Data-parallel schedule by PHASES (every step type has it; Pix2Pix bf16 prefers its finer bucketed schedule): the step is
        cut where a network's gradients become complete - ddp_phases() lists (phase id, networks complete after it) in the order
        the backward pass finishes them (cycle_gan.py:252-260 has four independent gradient sets) - one compute graph per phase;
        a finished network's exchange starts at once on the communicator's stream and runs beside the following phases; its Adam
        graph is replayed on a side stream as soon as the exchange has landed.  fp16: the inf/nan check must see every EXCHANGED
        gradient before any weight moves (all ranks take the same skip decision), so the exchanges still overlap the compute
        phases but one graph at the end checks, updates every network and adapts the loss scale."""
        ctx, sync = self.ctx, self.sync
        phases = self.ddp_phases()
        nets = self.nets()
        fp16 = ctx.ls is not None
        wire = self._wire_adam()
        main = torch.cuda.current_stream(ctx.device)
        lane4 = ctx.lane_stream(4)

        def graph(fn):
            return ctx.capture_graph(fn, CAPTURE_MODE)

        # every phase once eagerly: op lists, device tables and layer-stack plans are built (and uploaded) outside the captures
        s_ = torch.cuda.Stream(device=ctx.device)
        s_.wait_stream(main)
        with torch.cuda.stream(s_):
            for pid, _ in phases:
                self._forward_backward(*self._static_in, training, phase=pid)
        main.wait_stream(s_)
        torch.cuda.synchronize()
        G = [graph(lambda pid=pid: self._forward_backward(*self._static_in, training, phase=pid)) for pid, _ in phases]
        if fp16:
            A = [graph(self._update)]
        else:
            gs = 1.0 / sync.world if wire else sync.grad_scale
            wp = (lambda i: sync.wire[i].data_ptr()) if wire else (lambda i: None)
            A = [graph(lambda i=i, n=n: n.params.adam(self.lr, self.b1, self.b2, grad_scale=gs, wire_ptr=wp(i))) for i, n in enumerate(nets)]
        self._graphs = tuple(G + A)
        evs = [torch.cuda.Event() for _ in G]
        comm = torch.cuda.Stream(device=ctx.device)

        def boundary(k, pending):
            done = phases[k][1]
            if not done:
                return
            comm.wait_event(evs[k])
            with torch.cuda.stream(comm):
                for i in done:
                    sync.pack(i)                               # fp32 gradients -> wire format (no-op for the fp32 wire)
                started = [(i, sync.start(i)) for i in done]
            if fp16:
                pending += started
                return
            lane4.wait_event(evs[k])
            with torch.cuda.stream(lane4):
                for i, h in started:
                    sync.wait(h)                               # lane 4 waits for the collective; the host does not
                    if not wire:
                        sync.unpack(i)
                    A[i].replay()

        def replay(*inputs):
            for dst, src in zip(self._static_in, inputs):
                if src is not dst:
                    dst.copy_(src, non_blocking=True)
            cur = torch.cuda.current_stream(ctx.device)
            pending = []
            for k, gr in enumerate(G):
                gr.replay()
                evs[k].record(cur)
                if k > 0:
                    boundary(k - 1, pending)                   # issued behind the NEXT compute graph: the GPU never waits for the host
            boundary(len(G) - 1, pending)
            if fp16:
                for i, h in pending:
                    sync.wait(h)
                    sync.unpack(i)
                A[0].replay()
            else:
                ctx.join(cur, lane4)
            return self.losses
        replay.inputs = self._static_in
        return replay


class Pix2PixStep(_StepBase):
    # schedule constants (measured); attributes so that an experiment can change them per object
    dreal_on_side_lane = True    # D(real)'s forward on lane 2 beside the generator's launch-bound inner layers (+0.8 %)
    head_on_side_lane = False    # ... the discriminator's input packs + the dropout masks on that lane too, at the head of the step: measured
                                 # +-0 (5,672 / 5,688 / 5,687 with vs 5,695 / 5,682 / 5,707 without: the second launch and the event cost what they save)
    fused_wgrad_adam = True      # captured one-GPU step: the un-split wgrad launches apply Adam to their kernels themselves (GanAdamFuse)
    early_adam = False           # Adam + NK refresh of a stage's kernels on lane 4 as soon as its wgrads are done: +1.5 % in round 2,
    adam_lane = 4                # -0.5 % since the step became work-bound (round 3; adam_lane 3 = behind the wgrads on their lane: same)
    wgrad_cuts = (4, 8, 12)      # G's wgrad GEMMs in four coarse stages: up7..up4 | up3..up0 | down7..4 | down3..0 (finer: -3 %)
    d_wgrad_concurrent = 0       # planner hint of D's wgrad launches (GanWgradDesc.concurrent; 1: half-chip ping-pong plans with longer reductions)
    wgrad_alt = ('down3.kernel', 'down2.kernel', 'down1.kernel', 'down0.kernel')   # kernels of G whose wgrad launches run on a SECOND wgrad lane
                                 # (lane 4, own slab workspace): the last stage's GEMMs beside the optimiser-carrying (HBM-bound) launches of the
                                 # stage before instead of behind them: +1.0 % (5,363 -> 5,421 img/s, two interleaved pairs)
    adam_begin_early = True      # G's gan_adam_begin (step count, lr_t: one tiny launch) at the head of D(real)'s lane instead of on the main chain
                                 # in front of G's backward
    bias_grad_on_side = True     # the bias gradient of G's head (two launches that only feed Adam) as a side-stream op of the first wgrad stage,
                                 # off the dgrad chain
    ddp_buckets = True           # data parallel, bf16/f32: the bucketed schedule (False: the phased one)
    ddp_graphs = 4               # bucketed schedule: compute graphs per step (4, 3 or 2)
    ddp_late_comm = True         # a boundary's collectives are issued after the NEXT compute graph has been enqueued
    ddp_wire_direct = True       # bf16 all-reduce exchange: wgrad launches write their gradients straight into the wire buffer

    def __init__(self, ctx: Ctx, batch, size, channels=1, lam=100.0, lr=2e-4, beta_1=0.5, beta_2=0.999,
                 seed=123, dropout=True, nets=None, mask_stream=0):
        self.ctx, self.B, self.S, self.C = ctx, batch, size, channels
        self.lam, self.lr, self.b1, self.b2 = float(lam), lr, beta_1, beta_2
        if nets is not None:          # share weights with an existing step / model objects
            self.G, self.D = nets
        else:
            self.G = GeneratorNet(ctx, channels, 'batchnorm', seed=seed)          # pix2pix.py:29
            self.D = DiscriminatorNet(ctx, channels, True, 'batchnorm', seed=seed + 1)   # pix2pix.py:30
        self.g = self.G.new_call(batch, size, dropout=dropout, seed=seed, stream_id=mask_stream,   # mask_stream: train / val steps draw different masks
                                 wgrads_on_side_lane=ctx.lanes)
        self.d = self.D.new_call(batch, size, calls=2)
        self.losses = torch.zeros(8, dtype=torch.float32, device=ctx.device)
        self.l1_ws = torch.zeros(4096, dtype=torch.float32, device=ctx.device)
        self.bce_ws = torch.zeros(1024, dtype=torch.float32, device=ctx.device)
        self.sync = None             # GradSync for data-parallel runs
        self._early_adam = None

    def nets(self):
        return (self.G, self.D)

    def _prebuild_fused_adam(self):
        if not self.g._bwd_cache:
            self.g.alt_wgrad = frozenset(self.wgrad_alt)
            self.g.bias_grad_on_side = bool(self.bias_grad_on_side and self.ctx.lanes)
        if self._wgrad_adam_ok() and not self.early_adam:
            adam = (self.b1, self.b2)
            self.g.bwd_stages(list(self.wgrad_cuts), use_dgen2=True, adam=adam)       # (op lists, layer-stack plans, device tables)
            self.G.params.adam_rest_ops(self.g.adam_fused[(True, False, False, 'own', adam)], self.b1, self.b2)

    def _example_inputs(self):
        sh = (self.B, self.S, self.S, self.C)
        return [torch.zeros(sh, dtype=torch.float32, device=self.ctx.device) for _ in range(2)]

    def ddp_phases(self):
        return [(1, [0]), (2, [1])]          # (phase id, indices into nets() complete after it): G after phase 1, D after phase 2

    def _forward_backward(self, inp, tar, training=True, phase=0):
        B, Cc, g, d = self.B, self.C, self.g, self.d
        if not g._bwd_cache:
            g.alt_wgrad = frozenset(self.wgrad_alt)       # (settled before the first backward op list is built)
            g.bias_grad_on_side = bool(self.bias_grad_on_side and self.ctx.lanes)
            if not d._cache:
                d._bd2.wgrad_concurrent = int(self.d_wgrad_concurrent)
        if phase == 2:
            d.backward_params()
            return self.losses
        # inputs -> typed, channel-padded buffers.  D input = concat([inp, tar|gen]) (base_gan.py:139)
        dreal = self.ctx.lanes and self.dreal_on_side_lane
        head_split = bool(dreal and self.head_on_side_lane)
        if head_split:
            # the serial head of the step: only the generator's own input is needed at once; the discriminator's three input slices and
            # the dropout masks (first used by up0) are written on lane 2, where D(real) will run anyway, beside down0..down2
            main_, lane2_ = self.ctx.lane_stream(0), self.ctx.lane_stream(2)
            self._pack(inp, g.xin.view(0, Cc))
            lane2_.wait_stream(main_)
            with torch.cuda.stream(lane2_):
                self._pack_multi([(inp, d.xin.view(0, Cc, 0, B)), (inp, d.xin.view(0, Cc, B, B)), (tar, d.xin.view(Cc, Cc, 0, B))])
            masks_side = bool(g.auto_masks)
            if masks_side:
                self.ctx.run_on(g.mask_ops, lane2_)
            head_ev = torch.cuda.Event()
            head_ev.record(lane2_)
        else:
            self._pack_multi([(inp, g.xin.view(0, Cc)), (inp, d.xin.view(0, Cc, 0, B)), (inp, d.xin.view(0, Cc, B, B)),
                              (tar, d.xin.view(Cc, Cc, 0, B))])
        if dreal:
            # D(real) does not depend on the generator: it starts on lane 2 when G reaches its inner layers (down3 on:
            # launch-bound layers that leave the chip idle; measured best start point, +0.8 %) and D(fake) follows G on the main chain.  Same
            # BatchNormalization call order (real, then fake) as pix2pix.py:202-203.
            main_, lane2_ = self.ctx.lane_stream(0), self.ctx.lane_stream(2)

            # (the optimiser-carrying wgrad launches of G's backward need this step's lr_t: nothing before them reads it)
            begin_early = bool(self.adam_begin_early and training and phase == 0 and self._wgrad_adam_ok() and getattr(self, '_capturing', False)
                               and getattr(self, '_updating', False) and not self.early_adam)
            self._adam_begun = begin_early

            def start_dreal():
                if head_split:
                    main_.wait_event(head_ev)                        # masks (and the packs) are in place before the decoder needs them
                lane2_.wait_stream(main_)
                pre = self.G.params.adam_begin_ops(self.lr, self.b1, self.b2) if begin_early else []
                self.ctx.run_on(pre + d.forward_part_ops(0, lane=2), lane2_)
            if head_split:
                g.forward(inner_hook=start_dreal, masks_done=masks_side)
            else:
                g.forward(inner_hook=start_dreal)                     # pix2pix.py:200
        else:
            g.forward()                                               # pix2pix.py:200
        # generator loss (pix2pix.py:167-188): BCE(1, D(fake)) + lambda * mean|target - gen|; discriminator loss
        # (base_gan.py:233-245, factor 0.5 at pix2pix.py:206) - the L1 term (it only needs G's output: beside D's
        # forward when lanes are on), then all three BCE terms in one pass
        side = self.ctx.lane_stream(2) if self.ctx.lanes else None
        if dreal:
            self.ctx.join(self.ctx.lane_stream(0), side)                 # D(real) done (its BatchNorm updates come first)
        if side is not None:
            side.wait_stream(self.ctx.lane_stream(0))
        self._l1(g.out_view(), d.xin.view(Cc, Cc, 0, B), 2, 1.0, False, self.lam, g.dgen.view(0, Cc), stream=side)
        self._copy(g.out_view(), d.xin.view(Cc, Cc, B, B))
        if dreal:
            self.ctx.run(d.forward_part_ops(1))                       # pix2pix.py:203
        else:
            d.forward()                                               # pix2pix.py:202-203 (real ++ fake)
        if side is not None:
            self.ctx.join(self.ctx.lane_stream(0), side)
        real_ptr, cnt = d.logits_view(0)
        fake_ptr, _ = d.logits_view(1)
        lp = self.losses.data_ptr()
        L.check(self.ctx.lib.gan_patchgan_losses(real_ptr, fake_ptr, cnt, self.ctx.dt, d.dlogits_b.t.data_ptr(), d.dlogits_ptr(0),
                                                 d.dlogits_ptr(1), 8, self.lam, lp + 8, lp, lp + 4, lp + 12,
                                                 self.bce_ws.data_ptr(), self.ctx.ls_ptr, self.ctx.stream()), "patchgan_losses")
        if training:
            d.backward_input(1, dst=g.dgen2.view(0, Cc), c0=Cc)        # dL_G/d gen through D(fake), pre-update D: the `gen` channels only
            if phase == 3:                                            # bucketed data-parallel schedule: the caller stages G's backward
                return self.losses
            # two independent chains: D's parameter gradients (pix2pix.py:211) beside G's backward (:210)
            main, lane2 = self.ctx.lane_stream(0), self.ctx.lane_stream(2)
            if phase == 1 and self.ctx.lanes:   # data-parallel, phased: D's parameter pass is phase 2 (beside G's all-reduce)
                lane3 = self.ctx.lane_stream(3)
                g.wgrad_stream, g.wgrad_cuts, g.wgrad_stream2 = lane3, [8], None
                g.backward(use_dgen2=True, defer_wgrads='staged')
                self.ctx.join(main, lane3)
            elif phase == 1:
                g.backward(use_dgen2=True)
            elif self.ctx.lanes:              # three chains: D params | G dgrad/norm chain | G wgrads in coarse stages
                lane3 = self.ctx.lane_stream(3)
                lane2.wait_stream(main)
                self.ctx.run_on(d.params_ops(), lane2)
                if getattr(self, '_updating', False) and self.sync is None and self.ctx.ls is None:
                    # nothing else reads D's weights in this step: its (small) update runs at the end of its own chain
                    self.D.params.adam(self.lr, self.b1, self.b2, stream=lane2)
                    self._adam_done = (self.D,)
                # the decoder's wgrad GEMMs start on lane 3 once the main chain has passed the decoder, the encoder's
                # at its end (+2.5 % over one stage at the end; per-op dependencies, mode 5, lose 9 %)
                g.wgrad_stream, g.wgrad_cuts = lane3, list(self.wgrad_cuts)
                g.wgrad_stream2 = self.ctx.lane_stream(4) if g.alt_wgrad else None
                g.stage_hook = None
                if getattr(self, '_updating', False) and self.sync is None and self.ctx.ls is None and self.early_adam:
                    # a segment's kernel gradients are complete once its wgrads (a stage on lane 3) are done: its Adam +
                    # NK refresh (HBM-bound) runs on lane 4 beside the rest of the backward pass.  Stages: decoder
                    # (last, up6..up0) | down7..down4 | down3..down0 (the tail, updated after the join with the vectors)
                    P, lane4 = self.G.params, self.ctx.lane_stream(self.adam_lane)
                    g.wgrad_cuts = [8, 12]          # the segments below are cut at exactly these wgrads
                    if P._segments is None or len(P._segments) != 3:
                        P.split_kernels_at('down4.kernel', 'up0.kernel')
                    nst = len(g.wgrad_cuts)

                    self._early_segs = set()

                    def hook(k):
                        if k >= nst or k > 1:          # the last stage's segment is updated after the join
                            return
                        if lane4 is not lane3:
                            self.ctx.join(lane4, lane3)
                        ops = P.adam_begin_ops(self.lr, self.b1, self.b2) if k == 0 else []
                        self.ctx.run_on(ops + P.adam_segment_ops(2 - k, self.b1, self.b2), lane4)
                        self._early_segs.add(2 - k)
                    g.stage_hook = hook
                    self._early_adam = self.G
                wf = self._wgrad_adam_ok() and getattr(self, '_capturing', False) and getattr(self, '_updating', False) and not self.early_adam
                if wf:
                    # lr_t of this step must exist before the first fused wgrad; a stage's wgrads start after every dgrad of its
                    # layers has been enqueued (staged order), so rewriting those layers' weights there is safe
                    adam = (self.b1, self.b2)
                    if not (dreal and getattr(self, '_adam_begun', False)):
                        self.ctx.run(self.G.params.adam_begin_ops(self.lr, self.b1, self.b2))
                    g.backward(use_dgen2=True, defer_wgrads='staged', adam=adam)
                    self._adam_wfused = {self.G: g.adam_fused[(True, False, False, 'own', adam)]}
                else:
                    g.backward(use_dgen2=True, defer_wgrads='staged')
                g.stage_hook = None
                self.ctx.join(main, lane2)
                self.ctx.join(main, lane3)
                if g.wgrad_stream2 is not None:
                    self.ctx.join(main, g.wgrad_stream2)
                if self._early_adam is not None and self.adam_lane != 3:
                    self.ctx.join(main, self.ctx.lane_stream(self.adam_lane))
            elif self._wgrad_adam_ok() and getattr(self, '_capturing', False) and getattr(self, '_updating', False) and not self.early_adam:
                # one stream (profiling runs): the same fused-Adam wgrad launches, a stage's wgrads behind its dgrad chain
                adam = (self.b1, self.b2)
                self.ctx.run(self.G.params.adam_begin_ops(self.lr, self.b1, self.b2))
                for ops, wops in g.bwd_stages(list(self.wgrad_cuts), use_dgen2=True, adam=adam):
                    self.ctx.run(ops + wops)
                self._adam_wfused = {self.G: g.adam_fused[(True, False, False, 'own', adam)]}
                d.backward_params()
            else:
                g.backward(use_dgen2=True)
                d.backward_params()
        return self.losses                                            # Adam: _update() (pix2pix.py:213-216)

    # ---- data-parallel schedule: bucketed exchange overlapped with the backward pass ---------------------------
    def _capture_bucketed(self):
        """Five compute graphs with a gradient bucket leaving after each of the last three, and one Adam graph per
        bucket on a side stream as soon as that bucket's all-reduce has landed (the reference is
        single-device).  Stage k's wgrad GEMMs run beside stage k+1's dgrad/norm chain exactly as in the one-GPU
        schedule:

          G1  forward, losses, D's input-gradient pass, G backward stage 0 chain (decoder)
                                        ||  D's parameter-gradient pass -> bucket 4 = D (whole network)
          G2  stage 1 chain (down7..4)  ||  stage 0 wgrads              -> bucket 0 = decoder kernels
          G3  stage 2 chain (down3..0)  ||  stage 1 wgrads              -> bucket 1 = down7..4 kernels
          G4  stage 2 wgrads                                            -> bucket 2 = down3..0 kernels, 3 = G's vectors
        Weights are only rewritten by an Adam graph after every kernel that reads them in this step has been
        enqueued: a segment's NK copies feed the dgrads of its own stage, which precede its bucket."""
        ctx, g, d, sync = self.ctx, self.g, self.d, self.sync
        P, PD = self.G.params, self.D.params
        P.split_kernels_at('down4.kernel', 'up0.kernel')            # segments 0: down0..3 | 1: down4..7 | 2: up0..last
        o4, ou = P.entries['down4.kernel'][0], P.entries['up0.kernel'][0]
        self.buckets = [(0, ou, P.vec_start), (0, o4, ou), (0, 0, o4), (0, P.vec_start, P.total), (1, 0, PD.total)]
        # bf16 wire + all-reduce: the wgrad launches (their slab reduces / epilogues) write the wire format themselves - no fp32
        # gradient, no cast pass for those kernels; gan_grad_pack then covers only what is left of a bucket (the tap-folded first /
        # last layers and the vectors).  'rs_ag' reduces the fp32 gradients and keeps the cast of its own shard.
        direct = bool(self.ddp_wire_direct and sync.compress and getattr(sync, 'wire', None) is not None
                      and getattr(sync, 'exchange', 'allreduce') == 'allreduce')
        wG = sync.wire[0].data_ptr() if direct else None
        wD = sync.wire[1].data_ptr() if direct else None
        stages = g.bwd_stages([8, 12], use_dgen2=True, wire=wG)
        d_params = d.params_ops(wire=wD)
        assert len(stages) == 3
        skip = {0: set(g.wire_direct[(True, False, False, 'own', None, wG)]) if direct else set(), 1: set(d.wire_direct) if direct else set()}
        self._wire_direct_names = skip

        def pack_bucket(b):
            """Cast what the wgrad launches did not already write in the wire format: the bucket minus the direct kernels' ranges."""
            i, lo, hi = self.buckets[b]
            PS = (P, PD)[i]
            cuts = sorted((o, o + (int(torch.tensor(shape).prod()) + PS.ALIGN - 1) // PS.ALIGN * PS.ALIGN)
                          for n_, (o, shape) in PS.entries.items() if n_ in skip[i] and lo <= o < hi)
            a = lo
            for c0, c1 in cuts:
                if c0 > a:
                    sync.pack(i, a, c0)
                a = max(a, c1)
            if a < hi:
                sync.pack(i, a, hi)
        main = torch.cuda.current_stream(ctx.device)
        lane2, lane3, lane4 = ctx.lane_stream(2), ctx.lane_stream(3), ctx.lane_stream(4)
        self._static_in = [torch.zeros_like(t) for t in self._example_inputs()]
        torch.cuda.synchronize()
        s = torch.cuda.Stream(device=ctx.device)
        s.wait_stream(main)
        with torch.cuda.stream(s):           # warm-up outside capture (lazy inits, func attributes)
            self._forward_backward(*self._static_in, True, phase=1)
            self._forward_backward(*self._static_in, True, phase=2)
            for b in range(len(self.buckets)):
                pack_bucket(b)
        main.wait_stream(s)
        torch.cuda.synchronize()

        def graph(fn):
            return ctx.capture_graph(fn, CAPTURE_MODE)

        def fork_join(side_ops, side_stream, main_fn, bucket=None):
            """side_ops (a stage's wgrad GEMMs) and then the cast of `bucket` to the wire format on the side stream,
            beside main_fn on the current one."""
            cur = torch.cuda.current_stream(ctx.device)
            side_stream.wait_stream(cur)
            ctx.run_on(side_ops, side_stream)
            if bucket is not None:
                with torch.cuda.stream(side_stream):
                    pack_bucket(bucket)
            main_fn()
            ctx.join(cur, side_stream)

        def g1():
            self._forward_backward(*self._static_in, True, phase=3)          # everything up to G's backward
            # D's parameter-gradient pass runs on lane 2 beside the decoder chain, as in the one-GPU schedule; nothing
            # reads D's weights after this graph, so D's bucket is the first to leave
            cur = torch.cuda.current_stream(ctx.device)
            lane2.wait_stream(cur)
            ctx.run_on(d_params, lane2)
            with torch.cuda.stream(lane2):
                pack_bucket(4)
            ctx.run(stages[0][0])
            ctx.join(cur, lane2)

        def g2():
            fork_join(stages[0][1], lane3, lambda: ctx.run(stages[1][0]), bucket=0)

        def g3():
            fork_join(stages[1][1], lane3, lambda: ctx.run(stages[2][0]), bucket=1)

        def g4():
            ctx.run(stages[2][1])
            for b in (2, 3):
                pack_bucket(b)

        # Adam per bucket.  bf16 wire: Adam reads the exchanged gradient straight from the wire buffer (x 1/world) - no
        # unpack pass, and the fp32 gradient buffer keeps this rank's own gradient; fp32 wire: reduced in place, x 1/world
        gs = 1.0 / sync.world if sync.compress else sync.grad_scale
        wp = [w.data_ptr() for w in sync.wire] if sync.compress else [None, None]

        def a0():
            ctx.run(P.adam_begin_ops(self.lr, self.b1, self.b2) + P.adam_segment_ops(2, self.b1, self.b2, grad_scale=gs, wire_ptr=wp[0]))

        def a1():
            ctx.run(P.adam_segment_ops(1, self.b1, self.b2, grad_scale=gs, wire_ptr=wp[0]))

        def a2():
            ctx.run(P.adam_segment_ops(0, self.b1, self.b2, grad_scale=gs, vectors=True, wire_ptr=wp[0]))

        def a3():
            PD.adam(self.lr, self.b1, self.b2, grad_scale=gs, wire_ptr=wp[1])

        # The four logical stages can be captured as fewer graphs: ddp_graphs = 4 (one per stage; default: every
        # bucket leaves as early as it can), 3 (stages 3+4 together) or 2 (1+2 | 3+4).  On the one-rank rehearsal the three
        # are within noise of each other (3.57-3.65 ms): the boundaries (~50 us of drained lanes each) are not what the
        # schedule costs.  A bucket leaves at the end of the graph that holds its stage.
        stages_fn = (g1, g2, g3, g4)
        base_plan = {0: [(4, 3)], 1: [(0, 0)], 2: [(1, 1)], 3: [(2, None), (3, 2)]}
        grouping = {4: [[0], [1], [2], [3]], 3: [[0], [1], [2, 3]], 2: [[0, 1], [2, 3]]}[self.ddp_graphs]

        def group_fn(idx):
            def f():
                for k in idx:
                    stages_fn[k]()
            return f

        G = [graph(group_fn(idx)) for idx in grouping]
        A = [graph(f) for f in (a0, a1, a2, a3)]
        self._graphs = tuple(G + A)
        # after G_j: which buckets leave, and which Adam graph follows each of them
        plan = {j: [x for k in idx for x in base_plan[k]] for j, idx in enumerate(grouping)}
        # Host order matters: a compute graph is enqueued BEFORE the collectives / Adam graphs of the boundary behind it are
        # issued (they hang off an event recorded at that boundary, on their own launcher stream), so the GPU never waits
        # for the host's communicator calls between two compute graphs (that was ~50 us of idle chip per boundary).
        evs = [torch.cuda.Event() for _ in G]
        comm = torch.cuda.Stream(device=ctx.device)
        late = self.ddp_late_comm

        def boundary(k):
            todo = plan.get(k, ())
            if not todo:
                return
            comm.wait_event(evs[k])
            with torch.cuda.stream(comm):
                started = [(sync.start(*self.buckets[b]), ai) for b, ai in todo]
            lane4.wait_event(evs[k])                       # (the Adam graphs also read what the compute graphs wrote)
            with torch.cuda.stream(lane4):
                for h, ai in started:
                    sync.wait(h)                           # lane 4 waits for the collective; the host does not
                    if ai is not None:
                        A[ai].replay()

        def replay(*inputs):
            for dst, src in zip(self._static_in, inputs):
                if src is not dst:
                    dst.copy_(src, non_blocking=True)
            cur = torch.cuda.current_stream(ctx.device)
            for k, gr in enumerate(G):
                gr.replay()
                evs[k].record(cur)
                if not late:
                    boundary(k)
                elif k > 0:
                    boundary(k - 1)
            if late:
                boundary(len(G) - 1)
            ctx.join(cur, lane4)
            return self.losses
        replay.inputs = self._static_in
        return replay

    def train_step(self, input_image, target, training=True):
        """(gen_total_loss, gen_gan_loss, gen_l1_loss, disc_loss) as a 4-element device tensor view."""
        return self._run(input_image, target, training)[:4]


class CycleGANStep(_StepBase):
    ddp_buckets = False
    two_chains = True            # one-GPU step: the G_g-side and the G_f-side chains on two lanes (_forward_backward_merged)
    early_adam = True            # ... and every network's Adam where its gradients complete, inside the chains
    fused_wgrad_adam = True      # ... whose un-split launches apply Adam to their kernels themselves (captured step; GanAdamFuse)
    wide_wgrads = True           # ... one wgrad GEMM per layer over a generator's three invocations (host + guest call)
    adam_delay = (0, 0)          # ... stages by which chain A / B hold a segment's Adam back (measured: no offset is best)

    def __init__(self, ctx: Ctx, batch, size, channels=1, lam=10.0, lr=2e-4, beta_1=0.5, beta_2=0.999,
                 seed=123, dropout=True, nets=None, mask_stream=0, merged=True):
        self.ctx, self.B, self.S, self.C = ctx, batch, size, channels
        self.lam, self.lr, self.b1, self.b2 = float(lam), lr, beta_1, beta_2
        n = 'instancenorm'                                                    # cycle_gan.py:30-33
        if nets is not None:
            self.Gg, self.Gf, self.Dx, self.Dy = nets
        else:
            self.Gg = GeneratorNet(ctx, channels, n, seed=seed)
            self.Gf = GeneratorNet(ctx, channels, n, seed=seed + 1)
            self.Dx = DiscriminatorNet(ctx, channels, False, n, seed=seed + 2)
            self.Dy = DiscriminatorNet(ctx, channels, False, n, seed=seed + 3)
        mk = lambda net, sid, lane=0: net.new_call(batch, size, dropout=dropout, seed=seed, stream_id=mask_stream + sid, lane=lane)
        # The step is launch- and small-grid-bound (~1,200 launches): G_g(x) and G_g(y) - likewise G_f(y), G_f(x) - use the same
        # weights and InstanceNormalization is per sample, so the two invocations run as ONE call of batch 2B (exactly the
        # same arithmetic per sample, a third fewer generator launches); the cycle calls depend on their outputs and stay.
        # Measured +30 % (B=1) ... +15 % (B=16) pairs/s; merged=False keeps the six separate calls (equivalence test, A/B runs).
        self.merged = bool(merged)
        # chain "A" (lane 0 workspaces): G_g([x ; y]) -> G_f(fake_y) -> D_y;  chain "B" (lane 2): G_f([y ; x]) -> G_g(fake_x) -> D_x
        if self.merged:
            # the cycle call of a generator is the GUEST of its batched call (shared, wider saved tensors): one wgrad GEMM per
            # layer then covers all three invocations of the generator instead of a write + accumulate pair
            self.gA = self.Gg.new_call(2 * batch, size, dropout=dropout, seed=seed, stream_id=mask_stream + 0, guest_batch=batch)   # [fake_y ; same_y]
            self.gB = self.Gf.new_call(2 * batch, size, dropout=dropout, seed=seed, stream_id=mask_stream + 2, lane=2, guest_batch=batch)   # [fake_x ; same_x]
            mkc = lambda net, sid, lane, host: net.new_call(batch, size, dropout=dropout, seed=seed, stream_id=mask_stream + sid, lane=lane, host=host)
            self.cx, self.cy = mkc(self.Gf, 1, 0, self.gB), mkc(self.Gg, 3, 2, self.gA)      # cycled_x = G_f(fake_y); cycled_y = G_g(fake_x)
        else:
            self.cx, self.cy = mk(self.Gf, 1), mk(self.Gg, 3)
        self._wide = None
        if self.merged and ctx.lanes:       # planner hint of the wgrad GEMMs: they run beside the mirror chain's
            for c_ in (self.gA, self.gB, self.cx, self.cy):
                c_._bd.wgrad_concurrent = 2
        if self.merged:
            self.fy, self.sy = self.gA.half(0, batch), self.gA.half(batch, batch)
            self.fx, self.sx = self.gB.half(0, batch), self.gB.half(batch, batch)
        else:
            self.fy, self.fx = mk(self.Gg, 0), mk(self.Gf, 2)  # fake_y = G_g(x); fake_x = G_f(y)
            self.sx, self.sy = mk(self.Gf, 4), mk(self.Gg, 5)  # same_x = G_f(x); same_y = G_g(y)
        if self.merged:
            # (own workspaces for the parameter passes: they run beside the generators' wgrad lanes)
            self.dx = self.Dx.new_call(batch, size, calls=2, lane=2, params_lane=6)       # D_x(real_x) ++ D_x(fake_x): chain B
            self.dy = self.Dy.new_call(batch, size, calls=2, lane=0, params_lane=4)
            if ctx.lanes:
                self.dx._bd2.wgrad_concurrent = self.dy._bd2.wgrad_concurrent = 2
        else:
            self.dx = self.Dx.new_call(batch, size, calls=2)
            self.dy = self.Dy.new_call(batch, size, calls=2)
        self.losses = torch.zeros(12, dtype=torch.float32, device=ctx.device)     # [0..8] as below; [9] cycle term of chain B; [10] stays 0
        self.l1_ws = torch.zeros(4096, dtype=torch.float32, device=ctx.device)
        self.bce_ws = torch.zeros(1024, dtype=torch.float32, device=ctx.device)
        self.l1_ws_b = torch.zeros(4096, dtype=torch.float32, device=ctx.device)   # chain B's loss kernels run beside chain A's
        self.bce_ws_b = torch.zeros(1024, dtype=torch.float32, device=ctx.device)
        self.sync = None

    def nets(self):
        return (self.Gg, self.Gf, self.Dx, self.Dy)

    def _prebuild_fused_adam(self):
        if self._wgrad_adam_ok() and self.merged and self.two_chains and self.early_adam and self.wide_wgrads and self._wide:
            adam = (self.b1, self.b2)
            for call, net in ((self.gA, self.Gg), (self.gB, self.Gf)):
                call.bwd_stages([8, 12], use_dgen2=True, accumulate=True, wgrads='wide', adam=adam)
                net.params.adam_rest_ops(call.adam_fused[(True, False, True, 'wide', adam)], self.b1, self.b2)

    def gen_calls(self):
        return dict(fake_y=self.fy, cycled_x=self.cx, fake_x=self.fx, cycled_y=self.cy, same_x=self.sx, same_y=self.sy)

    def _example_inputs(self):
        sh = (self.B, self.S, self.S, self.C)
        return [torch.zeros(sh, dtype=torch.float32, device=self.ctx.device) for _ in range(2)]

    def ddp_phases(self):
        """Gradient sets in the order the backward pass completes them (cycle_gan.py:252-260): G_g | G_f | D_x | D_y."""
        if self.merged and self.ctx.lanes and self.two_chains:
            # the two chains complete both generators together, then both discriminators: two phases (the four-phase serial
            # schedule costs twice the compute per step: 8.5 against 4.7 ms at batch 4 in the one-rank rehearsal)
            return [(21, [0, 1]), (22, [2, 3])]
        if self.merged:
            return [(11, [0]), (12, [1]), (13, [2]), (14, [3])]
        return [(1, [0, 1]), (2, [2, 3])]

    def _forward_backward(self, real_x, real_y, training=True, phase=0):
        B, Cc, lam = self.B, self.C, self.lam
        if phase == 2:
            self.dx.backward_params(); self.dy.backward_params()
            return self.losses
        if phase == 12:
            self.gB.backward(use_dgen2=True, accumulate=True)          # G_f complete
            return self.losses
        if phase in (13, 14):
            (self.dx if phase == 13 else self.dy).backward_params()
            return self.losses
        if phase == 22:                                                # both discriminators' parameter passes, one per chain
            main, l2 = self.ctx.lane_stream(0), self.ctx.lane_stream(2)
            l2.wait_stream(main)
            self.dy.backward_params()
            with torch.cuda.stream(l2):
                self.dx.backward_params()
            self.ctx.join(main, l2)
            return self.losses
        if self.merged:
            return self._forward_backward_merged(real_x, real_y, training, phase)
        fy, cx, fx, cy, sx, sy, dx, dy = self.fy, self.cx, self.fx, self.cy, self.sx, self.sy, self.dx, self.dy
        self._pack_multi([(real_x, fy.xin.view(0, Cc)), (real_x, sx.xin.view(0, Cc)), (real_y, fx.xin.view(0, Cc)), (real_y, sy.xin.view(0, Cc))])
        self._pack_multi([(real_x, dx.xin.view(0, Cc, 0, B)), (real_y, dy.xin.view(0, Cc, 0, B))])
        fy.forward()                                                  # cycle_gan.py:220
        self._copy(fy.out_view(), cx.xin.view(0, Cc)); self._copy(fy.out_view(), dy.xin.view(0, Cc, B, B))
        cx.forward()                                                  # :221
        fx.forward()                                                  # :223
        self._copy(fx.out_view(), cy.xin.view(0, Cc)); self._copy(fx.out_view(), dx.xin.view(0, Cc, B, B))
        cy.forward()                                                  # :224
        sx.forward(); sy.forward()                                    # :227-228
        dx.forward(); dy.forward()                                    # :230-234
        rx_ptr, cnt = dx.logits_view(0); fxl_ptr, _ = dx.logits_view(1)
        ry_ptr, _ = dy.logits_view(0); fyl_ptr, _ = dy.logits_view(1)
        xv, yv = fy.xin.view(0, Cc), fx.xin.view(0, Cc)               # typed real_x / real_y
        self._bce(fyl_ptr, cnt, 1.0, 0, 1.0, False, 1.0, dy.dlogits_b.t.data_ptr())      # gen_g_loss :237
        self._bce(fxl_ptr, cnt, 1.0, 1, 1.0, False, 1.0, dx.dlogits_b.t.data_ptr())      # gen_f_loss :238
        self._l1(cx.out_view(), xv, 2, lam, False, lam, cx.dgen.view(0, Cc))              # total_cycle_loss :240
        self._l1(cy.out_view(), yv, 2, lam, True, lam, cy.dgen.view(0, Cc))
        self._l1(sy.out_view(), yv, 7, lam * 0.5, False, lam * 0.5, sy.dgen.view(0, Cc))  # identity :243
        self._l1(sx.out_view(), xv, 8, lam * 0.5, False, lam * 0.5, sx.dgen.view(0, Cc))  # identity :244
        lp = self.losses.data_ptr()               # total_gen_g / total_gen_f (:243-244): gen + cycle + identity, one tiny launch
        L.check(self.ctx.lib.gan_sum3(lp, lp + 8, lp + 28, lp + 12, 1, self.ctx.stream()), "sum3")      # [3] = [0] + [2] + [7]
        L.check(self.ctx.lib.gan_sum3(lp + 4, lp + 8, lp + 32, lp + 16, 1, self.ctx.stream()), "sum3")  # [4] = [1] + [2] + [8]
        self._bce(rx_ptr, cnt, 1.0, 5, 0.5, False, 0.5, dx.dlogits_ptr(0))                # disc_x_loss :246
        self._bce(fxl_ptr, cnt, 0.0, 5, 0.5, True, 0.5, dx.dlogits_ptr(1))
        self._bce(ry_ptr, cnt, 1.0, 6, 0.5, False, 0.5, dy.dlogits_ptr(0))                # disc_y_loss :247
        self._bce(fyl_ptr, cnt, 0.0, 6, 0.5, True, 0.5, dy.dlogits_ptr(1))
        if training:
            # cycle terms: backward through the second generator of each cycle yields BOTH its parameter
            # gradients and the gradient w.r.t. fake_y / fake_x (computed once, used twice)
            cx.backward(need_dx=True, accumulate=False)               # G_f grads (cycle_x), d/d fake_y
            cy.backward(need_dx=True, accumulate=False)               # G_g grads (cycle_y), d/d fake_x
            dy.backward_input(1, dst=fy.dgen.view(0, Cc))             # adversarial term through D_y(fake_y)
            self._copy(cx.dxin.view(0, Cc), fy.dgen2.view(0, Cc))
            fy.backward(use_dgen2=True, accumulate=True)              # G_g
            dx.backward_input(1, dst=fx.dgen.view(0, Cc))
            self._copy(cy.dxin.view(0, Cc), fx.dgen2.view(0, Cc))
            fx.backward(use_dgen2=True, accumulate=True)              # G_f
            sy.backward(accumulate=True)                              # identity_y -> G_g
            sx.backward(accumulate=True)                              # identity_x -> G_f
            if phase != 1:
                dx.backward_params(); dy.backward_params()            # :257-260
        return self.losses                                            # four Adam applies: _update() (:263-273)

    def _forward_backward_merged(self, real_x, real_y, training, phase):
        """The same step with [fake_y ; same_y] = G_g([x ; y]) and [fake_x ; same_x] = G_f([y ; x]) as two batch-2B calls."""
        B, Cc, lam = self.B, self.C, self.lam
        gA, gB, cx, cy, dx, dy = self.gA, self.gB, self.cx, self.cy, self.dx, self.dy
        fy, sy, fx, sx = self.fy, self.sy, self.fx, self.sx
        # Two independent halves until the losses and again in the backward pass (cycle_gan.py:220-234 lists them interleaved):
        # chain A = G_g([x ; y]) -> G_f(fake_y) -> D_y, chain B = G_f([y ; x]) -> G_g(fake_x) -> D_x.  The step is launch- and
        # small-grid-bound at the reference's batch sizes, so the chains run on two lanes of the captured graph.
        two = bool(self.ctx.lanes and self.two_chains and phase in (0, 21))
        main, l2 = self.ctx.lane_stream(0), self.ctx.lane_stream(2)
        chain_b = (lambda: torch.cuda.stream(l2)) if two else contextlib.nullcontext
        self._pack_multi([(real_x, fy.xin_view()), (real_y, sy.xin_view()), (real_y, fx.xin_view()), (real_x, sx.xin_view())])
        self._pack_multi([(real_x, dx.xin.view(0, Cc, 0, B)), (real_y, dy.xin.view(0, Cc, 0, B))])
        rx_ptr, cnt = dx.logits_view(0); fxl_ptr, _ = dx.logits_view(1)
        ry_ptr, _ = dy.logits_view(0); fyl_ptr, _ = dy.logits_view(1)
        xv, yv = fy.xin_view(), sy.xin_view()                         # typed real_x / real_y
        lp = self.losses.data_ptr()
        if two:
            l2.wait_stream(main)
        gA.forward()                                                  # cycle_gan.py:220 and :228
        self._copy(fy.out_view(), cx.xin.view(0, Cc)); self._copy(fy.out_view(), dy.xin.view(0, Cc, B, B))
        cx.forward()                                                  # :221
        dy.forward()                                                  # :233-234
        if two:
            # each chain takes the loss terms (and their gradients) of its own outputs; only the reported totals need both
            # chains and are summed after the final join - no join between the forward and the backward pass
            self._bce(fyl_ptr, cnt, 1.0, 0, 1.0, False, 1.0, dy.dlogits_b.t.data_ptr())      # gen_g_loss :237
            self._l1(cx.out_view(), xv, 2, lam, False, lam, cx.dgen.view(0, Cc))              # cycle term of x :240
            self._l1(sy.out_view(), yv, 7, lam * 0.5, False, lam * 0.5, sy.dgen_view())       # identity :243
            self._bce(ry_ptr, cnt, 1.0, 6, 0.5, False, 0.5, dy.dlogits_ptr(0))                # disc_y_loss :247
            self._bce(fyl_ptr, cnt, 0.0, 6, 0.5, True, 0.5, dy.dlogits_ptr(1))
        with chain_b():
            gB.forward()                                              # :223 and :227
            self._copy(fx.out_view(), cy.xin.view(0, Cc)); self._copy(fx.out_view(), dx.xin.view(0, Cc, B, B))
            cy.forward()                                              # :224
            dx.forward()                                              # :230-231
            if two:
                wl, wb = self.l1_ws_b, self.bce_ws_b
                self._bce(fxl_ptr, cnt, 1.0, 1, 1.0, False, 1.0, dx.dlogits_b.t.data_ptr(), ws=wb)      # gen_f_loss :238
                self._l1(cy.out_view(), yv, 9, lam, False, lam, cy.dgen.view(0, Cc), ws=wl)              # cycle term of y
                self._l1(sx.out_view(), xv, 8, lam * 0.5, False, lam * 0.5, sx.dgen_view(), ws=wl)       # identity :244
                self._bce(rx_ptr, cnt, 1.0, 5, 0.5, False, 0.5, dx.dlogits_ptr(0), ws=wb)                # disc_x_loss :246
                self._bce(fxl_ptr, cnt, 0.0, 5, 0.5, True, 0.5, dx.dlogits_ptr(1), ws=wb)
        if not two:
            self._bce(fyl_ptr, cnt, 1.0, 0, 1.0, False, 1.0, dy.dlogits_b.t.data_ptr())      # gen_g_loss :237
            self._bce(fxl_ptr, cnt, 1.0, 1, 1.0, False, 1.0, dx.dlogits_b.t.data_ptr())      # gen_f_loss :238
            self._l1(cx.out_view(), xv, 2, lam, False, lam, cx.dgen.view(0, Cc))              # total_cycle_loss :240
            self._l1(cy.out_view(), yv, 2, lam, True, lam, cy.dgen.view(0, Cc))
            self._l1(sy.out_view(), yv, 7, lam * 0.5, False, lam * 0.5, sy.dgen_view())       # identity :243
            self._l1(sx.out_view(), xv, 8, lam * 0.5, False, lam * 0.5, sx.dgen_view())       # identity :244
            self._totals(False)
            self._bce(rx_ptr, cnt, 1.0, 5, 0.5, False, 0.5, dx.dlogits_ptr(0))                # disc_x_loss :246
            self._bce(fxl_ptr, cnt, 0.0, 5, 0.5, True, 0.5, dx.dlogits_ptr(1))
            self._bce(ry_ptr, cnt, 1.0, 6, 0.5, False, 0.5, dy.dlogits_ptr(0))                # disc_y_loss :247
            self._bce(fyl_ptr, cnt, 0.0, 6, 0.5, True, 0.5, dy.dlogits_ptr(1))
        elif not training:
            self.ctx.join(main, l2)
            self._totals(True)
        if training:
            if two:
                # A: cycle_x through G_f, D_y's input gradient, then G_g's own backward and D_y's parameter pass;  B: the mirror
                # image.  The second backward of each generator ACCUMULATES onto what the OTHER chain's first one wrote: one
                # cross-wait.  With nothing else pending (one GPU, no loss scaling) each network's Adam runs where its gradients
                # complete: the generators' kernel segments inside their chain as soon as the backward pass has left them
                # (HBM-bound, beside the other chain's launch-bound kernels), the discriminators' after their parameter pass.
                fused_adam = bool(getattr(self, '_updating', False) and self.sync is None and self.ctx.ls is None and self.early_adam)

                if self._wide is None:       # host and guest must keep down0's output gradient in the same buffer (plan dependent)
                    self._wide = all(h.bwd_ops(True, False, True, 'wide') is not None and g_.bwd_ops(False, True, False, 'none') is not None
                                     and h.dy0_in_dA == g_.dy0_in_dA for h, g_ in ((gA, cy), (gB, cx)))
                wide = self._wide and self.wide_wgrads
                w1, w2 = ('none', 'wide') if wide else ('own', 'own')

                wf = bool(fused_adam and wide and self._wgrad_adam_ok() and getattr(self, '_capturing', False))

                def second_backward(call, net, delay):
                    """call.backward(use_dgen2, accumulate) with the network's Adam inside the chain: HBM-bound work of one chain
                    beside the launch-bound kernels of the other.  With wide wgrads in the captured step the un-split wgrad
                    launches apply Adam to their kernels themselves (GanAdamFuse; a stage's wgrads follow every dgrad of its
                    layers) and one small launch pair updates the rest; otherwise a kernel segment (decoder | down7..4 | down3..0)
                    is updated `delay` stages after its last wgrad GEMM.  Returns True when the vectors have been updated too.
                    (Side lanes forked from lane 2 end the capture with "unjoined work" on this runtime, and wgrad GEMMs on side
                    lanes lose here: profiles/r03_experiments_not_kept.txt.)"""
                    if not fused_adam:
                        call.backward(use_dgen2=True, accumulate=True, wgrads=w2)
                        return False
                    P = net.params
                    if wf:
                        adam = (self.b1, self.b2)
                        self.ctx.run(P.adam_begin_ops(self.lr, self.b1, self.b2))
                        for ops, wops in call.bwd_stages([8, 12], use_dgen2=True, accumulate=True, wgrads='wide', adam=adam):
                            self.ctx.run(ops + wops)
                        self.ctx.run(P.adam_rest_ops(call.adam_fused[(True, False, True, 'wide', adam)], self.b1, self.b2))
                        return True
                    if P._segments is None or len(P._segments) != 3:
                        P.split_kernels_at('down4.kernel', 'up0.kernel')
                    adam = lambda k: self.ctx.run((P.adam_begin_ops(self.lr, self.b1, self.b2) if k == 0 else []) +
                                                  P.adam_segment_ops(2 - k, self.b1, self.b2, vectors=False))
                    stages = call.bwd_stages([8, 12], use_dgen2=True, accumulate=True, wgrads=w2)
                    for k, (ops, wops) in enumerate(stages):
                        self.ctx.run(ops + wops)
                        if k - delay >= 0:
                            adam(k - delay)
                    for k in range(max(len(stages) - delay, 0), len(stages)):
                        adam(k)
                    return False

                cx.backward(need_dx=True, accumulate=False, wgrads=w1)    # G_f grads (cycle_x), d/d fake_y
                dy.backward_input(1, dst=fy.dgen_view())              # adversarial term through D_y(fake_y)
                self._copy(cx.dxin.view(0, Cc), fy.dgen_view(second=True))
                with chain_b():
                    cy.backward(need_dx=True, accumulate=False, wgrads=w1)    # G_g grads (cycle_y), d/d fake_x
                    dx.backward_input(1, dst=fx.dgen_view())
                    self._copy(cy.dxin.view(0, Cc), fx.dgen_view(second=True))
                ea, eb = torch.cuda.Event(), torch.cuda.Event()
                ea.record(main); eb.record(l2)
                main.wait_event(eb); l2.wait_event(ea)
                vdone = second_backward(gA, self.Gg, self.adam_delay[0])      # G_g
                if phase == 21:                                        # data-parallel: the generators' exchange starts here
                    with chain_b():
                        second_backward(gB, self.Gf, self.adam_delay[1])
                    self.ctx.join(main, l2)
                    self._totals(True)
                    return self.losses
                dy.backward_params()
                if fused_adam:
                    self.Dy.params.adam(self.lr, self.b1, self.b2, stream=main)
                    if not vdone:
                        self.ctx.run(self.Gg.params.adam_segment_ops(0, self.b1, self.b2, vectors=True, kernels=False))
                with chain_b():
                    vdone = second_backward(gB, self.Gf, self.adam_delay[1])      # G_f
                    dx.backward_params()
                    if fused_adam:
                        self.Dx.params.adam(self.lr, self.b1, self.b2, stream=l2)
                        if not vdone:
                            self.ctx.run(self.Gf.params.adam_segment_ops(0, self.b1, self.b2, vectors=True, kernels=False))
                if fused_adam:
                    self._adam_done = self.nets()
                self.ctx.join(main, l2)
                self._totals(True)
                return self.losses
            cx.backward(need_dx=True, accumulate=False)               # G_f grads (cycle_x), d/d fake_y
            cy.backward(need_dx=True, accumulate=False)               # G_g grads (cycle_y), d/d fake_x
            dy.backward_input(1, dst=fy.dgen_view())                  # adversarial term through D_y(fake_y)
            self._copy(cx.dxin.view(0, Cc), fy.dgen_view(second=True))
            dx.backward_input(1, dst=fx.dgen_view())
            self._copy(cy.dxin.view(0, Cc), fx.dgen_view(second=True))
            # second upstream slot of the identity halves stays zero (never written); one backward per generator covers the
            # adversarial + cycle gradient of fake_* and the identity gradient of same_*
            gA.backward(use_dgen2=True, accumulate=True)              # G_g
            if phase == 11:                                           # phased data-parallel schedule: G_g's exchange starts here
                return self.losses
            gB.backward(use_dgen2=True, accumulate=True)              # G_f
            if phase != 1:
                dx.backward_params(); dy.backward_params()
        return self.losses

    def _totals(self, split_cycle):
        """total_cycle_loss and total_gen_g / total_gen_f (cycle_gan.py:240-244): [3] = [0] + [2] + [7], [4] = [1] + [2] + [8];
        split_cycle: the two chains left the cycle terms of x and y in [2] and [9] ([10] is never written: 0)."""
        lp, lib, st_ = self.losses.data_ptr(), self.ctx.lib, self.ctx.stream()
        if split_cycle:
            L.check(lib.gan_sum3(lp + 8, lp + 36, lp + 40, lp + 8, 1, st_), "sum3")         # [2] += [9]
        L.check(lib.gan_sum3(lp, lp + 8, lp + 28, lp + 12, 1, st_), "sum3")
        L.check(lib.gan_sum3(lp + 4, lp + 8, lp + 32, lp + 16, 1, st_), "sum3")

    def train_step(self, real_x, real_y, training=True):
        """7 losses in the reference's order (cycle_gan.py:275-276)."""
        return self._run(real_x, real_y, training)[:7]
