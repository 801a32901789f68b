"""AdaMVS inference (the reference's default model) on the MI355X plane-sweep engine:
per-pair visibility net, visibility-weighted correlation volume, slice-recurrent conv-GRU
regulariser and online exp-sum regression.

Mirror of mvs/mvs_cas/models/adamvs.py (inference classes only; same names, constructor
arguments, forward() contract and state_dict keys).  Differences in HOW, not WHAT:
  * the stage-1 pair pass (adamvs.py:466-475, 48 single-plane warps per source) is one
    fused ops.pair_corr_mean launch per source;
  * the per-plane warp + weighting + normalisation (adamvs.py:492-509, 4 warps and ~20
    elementwise launches per plane) is one ops.weighted_corr launch per stage that writes
    the whole [C,D,h,w] similarity volume; the GRU then walks its planes;
  * view weights are resampled once per stage, not once per plane (adamvs.py:502).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .dataset import extract_features
from . import config as _cfg
from . import ops
from .module import (Conv2d, ConvBnReLU, ConvGRUCell, ConvReLU, DeConv2dFuse, _no_train, _trunk, feature_conv, folded_bn, trunk_conv0,
                     plane_depths)


AFFINE_SWEEP = True   # stages 2+ hand the sweep (lo, step) maps (ops.AffineDepth) instead of the [D,h,w] hypothesis volume


class FeatureNet(nn.Module):
    """adamvs.py:50-153 -- image pyramid with pooled context branches (PyTorch-ROCm/MIOpen)."""

    def __init__(self, base_channels, num_stage=3, stride=4):
        super().__init__()
        assert num_stage == 3
        b = base_channels
        self.stride, self.base_channels, self.num_stage = stride, b, num_stage
        self.conv0, self.conv1, self.conv2 = _trunk(b)

        def branch(pool, cin, cout):
            return nn.Sequential(nn.AvgPool2d((pool, pool), stride=(pool, pool)),
                                 Conv2d(cin, cout, 1, stride=1, padding=0, dilation=1))

        self.branch1_1, self.branch1_2 = branch(4, b * 4, b * 2), branch(8, b * 4, b * 2)
        self.out1 = nn.Conv2d(b * 8, b * 4, 1, bias=False)
        self.deconv1 = DeConv2dFuse(b * 4, b * 2, 3)
        self.deconv2 = DeConv2dFuse(b * 2, b, 3)
        self.branch2_1, self.branch2_2 = branch(4, b * 2, b), branch(8, b * 2, b)
        self.branch3_1, self.branch3_2 = branch(4, b, b // 2), branch(8, b, b // 2)
        self.out2 = nn.Conv2d(b * 4, b * 2, 1, bias=False)
        self.out3 = nn.Conv2d(b * 2, b, 1, bias=False)
        self.out_channels = [4 * b, 2 * b, b]

    @staticmethod
    def _context(feat, br_a, br_b, head):
        """head(cat(up(br_a(feat)), up(br_b(feat)), feat)) (adamvs.py:116-151).  Fused form: both pools in one read of feat,
        the head applied to the branch outputs at their own resolution (a 1x1 convolution commutes with the bilinear
        resize) and one streaming kernel for W_f feat + up(.) + up(.) -- no upsampled tensors, no concat."""
        if (feat.is_cuda and feat.dtype == torch.float32 and head.kernel_size == (1, 1) and head.bias is None
                and isinstance(br_a[0], nn.AvgPool2d) and br_a[0].kernel_size == (4, 4) and br_b[0].kernel_size == (8, 8)):
            outs = []
            for i in range(feat.shape[0]):
                f = feat[i].contiguous()
                pools = ops.avgpool_4_8(f)
                if pools is None:
                    break
                a, b = br_a[1](pools[0][None])[0], br_b[1](pools[1][None])[0]
                Ca, Cb, Co = a.shape[0], b.shape[0], head.out_channels
                w = head.weight.reshape(Co, -1)
                wa = ops.derived_weight(head.weight, "ctx_a", lambda t: t.reshape(Co, -1)[:, :Ca])
                wb = ops.derived_weight(head.weight, "ctx_b", lambda t: t.reshape(Co, -1)[:, Ca:Ca + Cb])
                wf = ops.derived_weight(head.weight, "ctx_f", lambda t: t.reshape(Co, -1)[:, Ca + Cb:])
                if w.shape[1] != Ca + Cb + f.shape[0]:
                    break
                a2 = torch.matmul(wa, a.reshape(Ca, -1)).reshape(Co, a.shape[1], a.shape[2])   # at 1/16 and 1/64 of the pixels
                b2 = torch.matmul(wb, b.reshape(Cb, -1)).reshape(Co, b.shape[1], b.shape[2])
                y = ops.conv1x1_context(f, wf, a2, b2)
                if y is None:
                    break
                outs.append(y)
            else:
                return outs[0].unsqueeze(0) if len(outs) == 1 else torch.stack(outs)
        size = feat.shape[2:]
        up = lambda t: F.interpolate(t, size=size, mode="bilinear", align_corners=False)
        return feature_conv(head, torch.cat((up(br_a(feat)), up(br_b(feat)), feat), 1))

    def forward(self, x):
        c0 = trunk_conv0(self.conv0, x)
        c1 = self.conv1(c0)
        c2 = self.conv2(c1)
        out = {"stage1": self._context(c2, self.branch1_1, self.branch1_2, self.out1)}
        f = self.deconv1(c1, c2)
        out["stage2"] = self._context(f, self.branch2_1, self.branch2_2, self.out2)
        f = self.deconv2(c0, f)
        out["stage3"] = self._context(f, self.branch3_1, self.branch3_2, self.out3)
        return out


class _Up2D(nn.Sequential):
    """ConvTranspose2d + BatchNorm2d + ReLU at indices 0,1,2 (adamvs.py:212-225 key layout)."""

    def __init__(self, ch):
        super().__init__(nn.ConvTranspose2d(ch, ch, kernel_size=3, padding=1, output_padding=1, stride=2, bias=False),
                         nn.BatchNorm2d(ch), nn.ReLU(inplace=True))

    def forward(self, x, skip):
        _no_train(self)
        s, t = folded_bn(self[1])
        return ops.convtranspose2d_k3s2(x, self[0].weight, s, t, skip, skip_after_act=True, act=1)


class CostRegNet2D(nn.Module):
    """adamvs.py:198-238 -- pair visibility UNet over depth-as-channels. forward(x [D,h,w]) -> [D,h,w]."""

    def __init__(self, in_channels, base_channels=8):
        super().__init__()
        c = in_channels
        self.conv0 = ConvBnReLU(c, c)
        self.conv1 = ConvBnReLU(c, c, stride=2)
        self.conv2 = ConvBnReLU(c, c)
        self.conv3 = ConvBnReLU(c, c, stride=2)
        self.conv4 = ConvBnReLU(c, c)
        self.conv5 = ConvBnReLU(c, c, stride=2)
        self.conv6 = ConvBnReLU(c, c)
        self.conv7, self.conv9, self.conv11 = _Up2D(c), _Up2D(c), _Up2D(c)
        self.prob = nn.Conv2d(c, c, 3, stride=1, padding=1)

    def forward(self, x):
        if x.shape[1] % 8 or x.shape[2] % 8:
            raise ValueError("CostRegNet2D needs h,w divisible by 8 (got %s)" % (tuple(x.shape[1:]),))
        c0 = self.conv0(x)
        c2 = self.conv2(self.conv1(c0))
        c4 = self.conv4(self.conv3(c2))
        y = self.conv6(self.conv5(c4))
        y = self.conv7(y, c4)
        y = self.conv9(y, c2)
        y = self.conv11(y, c0)
        return ops.conv2d_k3(y, self.prob.weight, None, self.prob.bias, None, act=0, stride=1)


class SliceCostRegNetRED(nn.Module):
    """adamvs.py:403-427 -- one depth slice through the 2-level conv-GRU encoder-decoder.
    forward(cost [C,h,w], state1 [8,h,w], state2 [16,h/2,w/2]) -> (reg [1,2h,2w] | [1,h,w], state1, state2)."""

    def __init__(self, in_channels, up=True, base_channels=8):
        super().__init__()
        b = base_channels
        self.base_channels, self.up = b, up
        self.conv1 = ConvReLU(in_channels, b, 3, 1, 1)
        self.conv_gru1 = ConvGRUCell(b, b, 3)
        self.conv2 = ConvReLU(b, b * 2, 3, 2, 1)
        self.conv_gru2 = ConvGRUCell(b * 2, b * 2, 3)
        self.upconv1 = nn.ConvTranspose2d(b * 2, b, kernel_size=3, stride=2, padding=1, output_padding=1)
        if up:
            self.upconv2d = nn.ConvTranspose2d(b, 1, kernel_size=3, stride=2, padding=1, output_padding=1)
        else:
            self.upconv2d = nn.Conv2d(b, 1, kernel_size=3, stride=1, padding=1)

    def forward(self, cost, state1, state2):
        with ops.slice_tile_kernels():
            return self._forward(cost, state1, state2)

    @staticmethod
    def _cell(pre, gru, x, state, stride):
        """relu(pre(x)) -> gru: one launch in bf16 mode (ops.gru_cell_conv_fused), the separate layers otherwise."""
        g, c = gru.conv_gates[0], gru.convc[0]
        fused = ops.gru_cell_conv_fused(x, state, pre.conv.weight, g.weight, g.bias, c.weight, c.bias, stride)
        if fused is None and x.dim() == 4:   # a CL8 cost plane the fused kernel refused after all: its planar fp32 form (exact)
            x = ops.from_cl(ops.cl8_to_cl(x[None]))[:, 0].contiguous()
        return fused if fused is not None else gru(pre(x), state)[0]

    def _cells(self, cost, state1, state2):
        state1 = self._cell(self.conv1, self.conv_gru1, cost, state1, 1)
        state2 = self._cell(self.conv2, self.conv_gru2, state1, state2, 2)
        return state1, state2

    def _trunk(self, cost, state1, state2):
        state1, state2 = self._cells(cost, state1, state2)
        # relu(upconv1(state2) + state1): skip added before the activation (adamvs.py:423-424)
        up = ops.convtranspose2d_k3s2(state2, self.upconv1.weight, None, self.upconv1.bias, state1,
                                      skip_after_act=False, act=1)
        return up, state1, state2

    def _head(self, up):
        if self.up:
            return ops.convtranspose2d_k3s2(up, self.upconv2d.weight, None, self.upconv2d.bias, None, act=0)
        return ops.conv2d_k3(up, self.upconv2d.weight, None, self.upconv2d.bias, None, act=0)

    def _forward(self, cost, state1, state2):
        up, state1, state2 = self._trunk(cost, state1, state2)
        return self._head(up), state1, state2

    def takes_cl8(self, C, h, w):
        """Will the first cell of a slice run as the fused kernel on a CL8 cost plane?  (Only then may the sweep write that form:
        the unfused layers read planar fp32 planes.)"""
        # (8 channels -- the last stage -- stay planar: one 16-byte cell per pixel makes twice the staging tasks of the planar quads and
        #  the cell measures 255 against 233 us there; at 16 / 32 channels it is 80 / 36 against 98 / 48 us, tools/cl8_cell_bench.py)
        return (ops.conv_precision() == "h16" and not _cfg.off("gru_fused") and not _cfg.off("corr_cl8") and _cfg.get("D3D_CONV") != "direct"
                and C in (16, 32) and w % 4 == 0 and self.base_channels == 8)

    def loop_graph(self, C, h, w, D, dv):
        """The captured slice loop for this shape on the caller's stream (SliceLoopGraph), or None where it does not apply: h16
        mode with the fused cell / tail kernels only, not while somebody else is capturing, D3D_KERNELS_OFF=slice_graph."""
        import threading

        if (not dv.is_cuda or ops.conv_precision() != "h16" or _cfg.off("slice_graph") or _cfg.off("gru_fused") or D < 4 or h % 2 or w % 2
                or (_cfg.off("tail_fused") if self.up else _cfg.off("head_fused")) or torch.cuda.is_current_stream_capturing()
                or threading.current_thread() is not threading.main_thread()):
            # (worker threads run the launch loop: with two threads capturing and replaying their own graphs side by side the
            #  results were not bit-stable on this stack -- round 5, tests/test_parity_gpu.py::test_two_threads_... -- and a
            #  capture is a process-wide event in the HIP runtime; the main thread's forwards are where the gain is)
            return None
        return SliceLoopGraph.get(self, C, h, w, D, dv)

    def _tail(self, s2, s1, dplane, max_p, sum_d, sum_p):
        """upconv1 + skip + head + regression update of one slice from its two new states (the second half of step_regress)."""
        if self.up:
            return ops.slice_tail_regress(s2, self.upconv1.weight, self.upconv1.bias, s1, self.upconv2d.weight, self.upconv2d.bias,
                                          dplane, max_p, sum_d, sum_p)
        if ops.slice_tail_regress_same(s2, self.upconv1.weight, self.upconv1.bias, s1, False, self.upconv2d.weight, self.upconv2d.bias,
                                       dplane, max_p, sum_d, sum_p):   # (last stage: the head keeps `up`'s resolution)
            return True
        up = ops.convtranspose2d_k3s2(s2, self.upconv1.weight, None, self.upconv1.bias, s1, skip_after_act=False, act=1)
        return ops.slice_head_regress(up, self.upconv2d.weight, self.upconv2d.bias, self.up, dplane, max_p, sum_d, sum_p)

    def step_regress(self, cost, state1, state2, dplane, max_p, sum_d, sum_p):
        """One slice AND its online-regression update (adamvs.py:512-525 around this module): in bf16 mode the head layer and
        the update are one kernel and `reg` never reaches memory (ops.slice_head_regress).  Returns the new states."""
        with ops.slice_tile_kernels():
            if self.up:   # (stages 1 / 2: upconv1 + head + update in one launch; `up` stays in LDS)
                s1, s2 = self._cells(cost, state1, state2)
                if ops.slice_tail_regress(s2, self.upconv1.weight, self.upconv1.bias, s1, self.upconv2d.weight, self.upconv2d.bias,
                                          dplane, max_p, sum_d, sum_p):
                    return s1, s2
                up = ops.convtranspose2d_k3s2(s2, self.upconv1.weight, None, self.upconv1.bias, s1, skip_after_act=False, act=1)
                state1, state2 = s1, s2
            else:         # (last stage: the same with the head at `up`'s resolution)
                s1, s2 = self._cells(cost, state1, state2)
                if ops.slice_tail_regress_same(s2, self.upconv1.weight, self.upconv1.bias, s1, False, self.upconv2d.weight,
                                               self.upconv2d.bias, dplane, max_p, sum_d, sum_p):
                    return s1, s2
                up = ops.convtranspose2d_k3s2(s2, self.upconv1.weight, None, self.upconv1.bias, s1, skip_after_act=False, act=1)
                state1, state2 = s1, s2
            if not ops.slice_head_regress(up, self.upconv2d.weight, self.upconv2d.bias, self.up, dplane, max_p, sum_d, sum_p):
                ops.online_regress_update(self._head(up)[0], dplane, max_p, sum_d, sum_p)
        return state1, state2


class SliceLoopGraph(object):
    """The slice loop of one cascade stage (adamvs.py:492-525: D times cell 1 -> cell 2 -> upconv1 + skip + head + regression
    update) captured ONCE as a HIP graph and replayed per reference view.

    Why: the loop is 3 D launches of 30 - 200 us kernels of a few hundred workgroups each.  Across slices each of the three
    depends on ITS OWN predecessor only -- state1(d) needs state1(d - 1); state2(d) needs state2(d - 1) and state1(d); the
    accumulators need tail(d - 1) -- so cell 1 of slice d + 2, cell 2 of slice d + 1 and the tail of slice d can be in flight
    together and fill each other's launch / drain gaps and partial last rounds.  Issued eagerly from three streams that form
    is SLOWER than the serial loop (29.3 against 27.7 ms per view): Python cannot issue three launches and seven event
    operations in the 43 us a stage-1 kernel lasts.  Captured, the host issues ONE graph launch per stage.

    What is captured: three streams (the capture stream = chain 1; two side streams = chains 2 and 3), the states in rings of
    three buffers each (slice d + 3 reuses slice d's buffers behind an event of slice d's tail: nothing depends on the caching
    allocator inside the graph), the zeroing of the first states and of the accumulators.  Same kernels on the same operands
    as the serial loop: bit-identical (tests/test_parity_gpu.py::test_adamvs_slice_graph_is_the_serial_loop).
    Static inputs: `sim` [D,C,h,w] (the weighted sweep writes it in place: ops.weighted_corr(out=)) and `dv` (copied in, < 0.1 ms).
    One graph per (module, shape, caller stream, packed weights): two forwards in flight on two streams own separate graphs
    and buffers.  The first call of a shape runs the serial loop (it packs the weights and sets the kernels' LDS attributes --
    neither may happen inside a capture), the second captures."""

    _cache = {}
    _lock = __import__("threading").Lock()
    MAX_GRAPHS = 12
    CHAINS = 3   # 1: the serial order on one captured stream; 2: cells on one stream, tails on another
    UP_ONLY = False

    @classmethod
    def get(cls, mod, C, h, w, D, dv):
        dev = dv.device
        # (the kernel-selection switches are part of the key: a graph replays the kernels that were dispatched when it was captured)
        key = (id(mod), C, h, w, D, tuple(dv.shape), dev.index, torch.cuda.current_stream(dev).cuda_stream, ops.h16_dtype(),
               tuple(sorted(_cfg.switches.items())))
        with cls._lock:
            for k in [k for k, v in cls._cache.items() if v.mod() is None]:   # graphs of modules that are gone (and their buffers)
                del cls._cache[k]
            g = cls._cache.get(key)
            if g is None or g.mod() is not mod:
                if len(cls._cache) >= cls.MAX_GRAPHS:
                    cls._cache.pop(next(iter(cls._cache)))
                g = cls._cache[key] = cls(mod, C, h, w, D, dv)
        return g

    def __init__(self, mod, C, h, w, D, dv):
        import weakref

        self.mod = weakref.ref(mod)
        self.C, self.h, self.w, self.D, self.up = C, h, w, D, mod.up
        self.dev = dv.device
        self.calls, self.graph, self.counts, self.wkey, self.failed = 0, None, None, None, None
        self.sim = self.dv = None
        self.dv_shape = tuple(dv.shape)

    def _weights_key(self, mod):
        ps = [mod.conv1.conv.weight, mod.conv2.conv.weight, mod.upconv1.weight, mod.upconv2d.weight]
        for gru in (mod.conv_gru1, mod.conv_gru2):
            ps += [gru.conv_gates[0].weight, gru.convc[0].weight]
        return tuple((p.data_ptr(), p._version) for p in ps)

    def usable(self):
        """True from the second call on (the first one of a shape runs the serial loop); a changed weight starts over."""
        mod = self.mod()
        k = self._weights_key(mod)
        if k != self.wkey:
            self.wkey, self.calls, self.graph = k, 0, None
        self.calls += 1
        return self.calls >= 2 and self.failed is None

    def _alloc(self):
        dev, f32 = self.dev, torch.float32
        H, W = (2 * self.h, 2 * self.w) if self.up else (self.h, self.w)
        # the volume the sweep writes: CL8 16-bit cells where the cell takes them (ops.weighted_corr_cl8), the planar fp32 volume otherwise
        self.cl8 = self.mod().takes_cl8(self.C, self.h, self.w)
        self.sim = torch.empty((self.D, self.C // 8, self.h, self.w, 8), dtype=ops.h16_dtype(), device=dev) if self.cl8 else \
            torch.empty((self.D, self.C, self.h, self.w), dtype=f32, device=dev)
        self.dv = torch.empty(self.dv_shape, dtype=f32, device=dev)
        self.s1 = [torch.empty((8, self.h, self.w), dtype=f32, device=dev) for _ in range(4)]
        self.s2 = [torch.empty((16, self.h // 2, self.w // 2), dtype=f32, device=dev) for _ in range(4)]
        self.max_p = torch.empty((H, W), dtype=f32, device=dev)
        self.sum_d, self.sum_p = torch.empty_like(self.max_p), torch.empty_like(self.max_p)
        self.streams = [torch.cuda.Stream(dev) for _ in range(3)]   # capture stream (chain 1), chains 2 and 3

    def _capture(self):
        mod = self.mod()
        D = self.D
        dplane = (lambda d: self.dv[d].reshape(1, 1)) if self.dv.dim() == 1 else (lambda d: self.dv[d])
        g1, c1 = mod.conv_gru1.conv_gates[0], mod.conv_gru1.convc[0]
        g2, c2 = mod.conv_gru2.conv_gates[0], mod.conv_gru2.convc[0]
        sa, sb, sc = self.streams
        chains = SliceLoopGraph.CHAINS if (self.up or not SliceLoopGraph.UP_ONLY) else 1
        if chains == 1:   # (probe / fallback form: the serial order inside one captured stream)
            sb = sc = sa
        elif chains == 2:
            sb = sa
        before = dict(ops.dispatch_counts)
        # One slice eagerly first, on the ring buffers (the capture overwrites them): whatever the three kernels prepare on first
        # use -- packed weights in the caches, the kernels' LDS attributes -- exists afterwards, so the capture meets no operation
        # that cannot be captured even if a cache was emptied since the serial call.
        with ops.slice_tile_kernels():
            ops.gru_cell_conv_fused(self.sim[0], self.s1[3], mod.conv1.conv.weight, g1.weight, g1.bias, c1.weight, c1.bias, 1, out=self.s1[0])
            ops.gru_cell_conv_fused(self.s1[0], self.s2[3], mod.conv2.conv.weight, g2.weight, g2.bias, c2.weight, c2.bias, 2, out=self.s2[0])
            mod._tail(self.s2[0], self.s1[0], dplane(0), self.max_p, self.sum_d, self.sum_p)
        ops.dispatch_counts.clear()
        ops.dispatch_counts.update(before)
        graph = torch.cuda.CUDAGraph()
        sa.wait_stream(torch.cuda.current_stream(self.dev))
        keep = []   # every event of the capture stays alive until the capture has ended
        def rec(st):
            keep.append(st.record_event())
            return keep[-1]
        def wait(st, ev):
            if chains != 1:
                st.wait_event(ev)
        failure = None
        with torch.cuda.graph(graph, stream=sa, capture_error_mode="thread_local"), ops.slice_tile_kernels():
          try:
            # ring slot 3 holds the zero states of slice 0; slots 0 .. 2 take the slices' states in turn
            self.s1[3].zero_(); self.s2[3].zero_()
            self.max_p.zero_(); self.sum_d.zero_(); self.sum_p.zero_()
            if chains != 1:
                fork = rec(sa)
                if sb is not sa:
                    sb.wait_event(fork)
                sc.wait_event(fork)
            ec = [None] * D
            eb = None
            for d in range(D):
                a_in, a_out = self.s1[3 if d == 0 else (d - 1) % 3], self.s1[d % 3]
                b_in, b_out = self.s2[3 if d == 0 else (d - 1) % 3], self.s2[d % 3]
                if d >= 3:
                    wait(sa, ec[d - 3])   # the slot's previous occupant has been read by its cell 2 and its tail
                ok = ops.gru_cell_conv_fused(self.sim[d], a_in, mod.conv1.conv.weight, g1.weight, g1.bias, c1.weight, c1.bias, 1, out=a_out)
                ea = rec(sa)
                with torch.cuda.stream(sb):
                    wait(sb, ea)   # (cell 1 of this slice waited for tail(d - 3): the state-2 slot is free as well)
                    ok2 = ops.gru_cell_conv_fused(a_out, b_in, mod.conv2.conv.weight, g2.weight, g2.bias, c2.weight, c2.bias, 2, out=b_out)
                    eb = rec(sb)
                with torch.cuda.stream(sc):
                    wait(sc, eb)          # (behind cell 2 of this slice, hence behind its cell 1)
                    ok3 = mod._tail(b_out, a_out, dplane(d), self.max_p, self.sum_d, self.sum_p)
                    ec[d] = rec(sc)
                if ok is None or ok2 is None or not ok3:
                    raise RuntimeError("a fused slice kernel refused a shape the serial loop ran it on")
            wait(sa, eb)
            wait(sa, ec[D - 1])
          except Exception as e:
            # leave the capture in an orderly way -- every stream that joined it is joined back to the capture stream, so that
            # ending the capture releases all of them -- and report the failure afterwards
            failure = e
            try:
                if sb is not sa:
                    sa.wait_stream(sb)
                if sc is not sa:
                    sa.wait_stream(sc)
            except Exception:
                pass
        self._events = keep
        if failure is not None:
            raise failure
        after = dict(ops.dispatch_counts)
        self.counts = {k: v - before.get(k, 0) for k, v in after.items() if v != before.get(k, 0)}
        for k, v in self.counts.items():   # (the capture launched nothing: it does not count; every replay does)
            ops.dispatch_counts[k] -= v
        self.graph = graph

    def run(self, dv):
        """Replays the loop on the caller's stream over what the sweep has written into self.sim.  Returns (max_p, sum_d, sum_p)."""
        if self.graph is None:
            with SliceLoopGraph._lock:
                if self.sim is None:
                    raise RuntimeError("SliceLoopGraph.run before buffers()")
                before = dict(ops.dispatch_counts)
                try:
                    self._capture()
                except Exception as e:
                    # On this stack (ROCm 7.2) an invalidated capture leaves the thread's HIP state unusable ("operation failed due
                    # to a previous error during capture" on every later call), so there is no falling back to the launch loop:
                    # say what happened and how to run without the graph.
                    self.failed = repr(e)[:300]
                    raise RuntimeError("capturing the slice loop as a HIP graph failed (%s); run with D3D_KERNELS_OFF=slice_graph"
                                       % self.failed) from e
                torch.cuda.current_stream(self.dev).wait_stream(self.streams[0])
        self.dv.copy_(dv)
        self.graph.replay()
        for k, v in self.counts.items():
            ops.dispatch_counts[k] += v
        return self.max_p, self.sum_d, self.sum_p

    def buffers(self):
        if self.sim is None:
            self._alloc()
        return self


class InferDepthNet(nn.Module):
    """adamvs.py:429-531."""

    def __init__(self, in_depths, in_channels, in_up=True, base_channels=8):
        super().__init__()
        self.in_up = in_up
        self.reg = CostRegNet2D(in_depths, base_channels)
        self.reg_fuse = SliceCostRegNetRED(in_channels, in_up, base_channels)

    def _one(self, feats, proj44, dv, D, conf_in, dv_sweep=None):
        """feats: V x [C,h,w]; dv [D] or [D,h,w]; conf_in: None or [V-1,hc,wc]; dv_sweep: the (lo, step) maps that generate dv
        (ops.AffineDepth), handed to the sweep instead of the volume -- no depth load in its plane loop, same values."""
        C, h, w = feats[0].shape
        if h % 2 or w % 2:
            raise ValueError("feature map must have even size (got %dx%d)" % (h, w))
        dev = feats[0].device
        p34 = ops.compose_projections(proj44)
        pair_results = []
        if conf_in is None:  # stage 1: per-pair visibility (adamvs.py:465-489)
            def one_pair(i):   # (the pairs are independent chains of small launches: three of them in flight, ops.on_streams)
                corr = ops.pair_corr_mean(feats[0], feats[i], p34[i - 1], dv)
                return ops.pair_softmax_max(self.reg(corr), dv)

            res = ops.on_streams([(lambda i=i: one_pair(i)) for i in range(1, len(feats))], dev, "pair_streams")
            pair_results = [pd for _, pd in res]
            weights = torch.stack([vw for vw, _ in res])
        elif tuple(conf_in.shape[1:]) == (h, w):
            weights = conf_in
        else:  # adamvs.py:502, once per stage instead of once per plane
            weights = ops.resize_bilinear(conf_in, h, w)

        loop = self.reg_fuse.loop_graph(C, h, w, D, dv)
        acc = None
        dsweep = dv if dv_sweep is None else dv_sweep
        if loop is not None and loop.usable():
            # the slice loop as one HIP graph (SliceLoopGraph): the sweep writes the graph's static volume, the graph is replayed
            sim = loop.buffers().sim
            if loop.cl8:
                if ops.weighted_corr_cl8(feats, p34, weights, dsweep, out=sim) is None:
                    raise RuntimeError("the sweep refused the CL8 volume it wrote on the previous call of this shape")
            else:
                ops.weighted_corr(feats, p34, weights, dsweep, plane_major=True, out=sim)
            acc = loop.run(dv)
        else:
            # fast mode: the volume as CL8 16-bit cells (plane d = what the fused cell stages with 16-byte loads); else [D,C,h,w] fp32
            sim = ops.weighted_corr_cl8(feats, p34, weights, dsweep) if self.reg_fuse.takes_cl8(C, h, w) else None
            if sim is None:
                sim = ops.weighted_corr(feats, p34, weights, dsweep, plane_major=True)  # plane d is one contiguous block
        if acc is None:   # the launch loop
            H, W = (2 * h, 2 * w) if self.in_up else (h, w)
            s1 = torch.zeros((8, h, w), dtype=torch.float32, device=dev)
            s2 = torch.zeros((16, h // 2, w // 2), dtype=torch.float32, device=dev)
            max_p = torch.zeros((H, W), dtype=torch.float32, device=dev)
            sum_d = torch.zeros_like(max_p)
            sum_p = torch.zeros_like(max_p)
            for d in range(D):
                dplane = dv[d].reshape(1, 1) if dv.dim() == 1 else dv[d]
                s1, s2 = self.reg_fuse.step_regress(sim[d], s1, s2, dplane, max_p, sum_d, sum_p)
        else:
            max_p, sum_d, sum_p = acc
        depth, conf = ops.online_regress_finalize(max_p, sum_d, sum_p)
        return depth, conf, weights, pair_results

    def forward(self, features, proj_matrices, depth_values, num_depth, confidence_map=None, sweep_depth=None):
        assert len(features) == proj_matrices.shape[1], "Different number of images and projection matrices"
        assert depth_values.shape[1] == num_depth, "depth_values.shape[1]:{}  num_depth:{}".format(
            depth_values.shape[1], num_depth)
        _no_train(self)
        B = features[0].shape[0]
        nsrc = len(features) - 1
        depths, confs, weights, pairs = [], [], [], []
        for b in range(B):
            conf_in = None
            if confidence_map is not None:
                conf_in = torch.stack([confidence_map[i][b, 0] for i in range(nsrc)]).contiguous()
            d, c, wts, pr = self._one([f[b].contiguous() for f in features], proj_matrices[b].contiguous(),
                                      depth_values[b].contiguous(), num_depth, conf_in,
                                      None if sweep_depth is None else sweep_depth[b])
            depths.append(d)
            confs.append(c)
            weights.append(wts)
            pairs.append(pr)
        # pair_confidence: V-1 tensors [B,1,h,w] at this stage's resolution -- the entries of
        # the reference's list that the next stage actually reads (adamvs.py:502,612).
        pair_confidence = [torch.stack([weights[b][i] for b in range(B)]).unsqueeze(1) for i in range(nsrc)]
        pair_result = [torch.stack([pairs[b][i] for b in range(B)]) for i in range(len(pairs[0]))]
        return {"depth": torch.stack(depths), "photometric_confidence": torch.stack(confs),
                "pair_confidence": pair_confidence, "pair_result": pair_result}


class Infer_AdaMVSNet(nn.Module):
    """adamvs.py:535-617. forward(imgs [B,V,3,H,W], proj_matrices {stageN: [B,V,4,4]}, depth_values [B,2])."""

    def __init__(self, num_depth=384, ndepths=[48, 32, 8], depth_intervals_ratio=[4, 2, 1], share_cr=False,
                 cr_base_chs=[8, 8, 8]):
        super().__init__()
        assert len(ndepths) == len(depth_intervals_ratio)
        self.num_depth, self.share_cr, self.ndepths = num_depth, share_cr, list(ndepths)
        self.depth_intervals_ratio, self.cr_base_chs = list(depth_intervals_ratio), list(cr_base_chs)
        self.num_stage = len(ndepths)
        self.stage_infos = {"stage1": {"scale": 4.0}, "stage2": {"scale": 2.0}, "stage3": {"scale": 1.0}}
        self.feature = FeatureNet(base_channels=8, stride=4, num_stage=self.num_stage)
        oc = self.feature.out_channels
        self.DepthNet = nn.ModuleList([InferDepthNet(in_depths=self.ndepths[0], in_channels=oc[0]),
                                       InferDepthNet(in_depths=self.ndepths[0], in_channels=oc[1]),
                                       InferDepthNet(in_depths=self.ndepths[0], in_up=False, in_channels=oc[2])])

    feature_cache = None  # dataset.FeatureCache shared across reference views (set by the harness); see image_keys

    def forward(self, imgs, proj_matrices, depth_values, image_keys=None):
        """image_keys (optional, with self.feature_cache set): one hashable key per view; the feature pyramid of a
        key seen before is reused instead of recomputed, and imgs may then be a list whose cached entries are None."""
        dmin, dmax = ops.depth_range_host(depth_values)   # (host numbers: no device access when the caller noted them)
        depth_interval = (dmax - dmin) / self.num_depth
        features = extract_features(self.feature, imgs, image_keys, self.feature_cache)
        V = len(features)
        B, _, img_h, img_w = features[0]["stage3"].shape  # the finest level has the image's size
        outputs = {}
        depth, pair_confidence = None, None
        for s in range(self.num_stage):
            key = "stage%d" % (s + 1)
            feats = [f[key] for f in features]
            D = self.ndepths[s]
            if depth is None:
                dv = plane_depths(depth_values, D)  # [B,D]: linspace(min,max) (module.py:637-642)
            else:  # previous stage already has this stage's resolution (adamvs.py:589-592)
                dv = torch.stack([ops.depth_range_samples(depth[b].contiguous(), D,
                                                          self.depth_intervals_ratio[s] * depth_interval)
                                  for b in range(B)])
                # the same hypotheses as two maps for the sweep (bit-identical values; the volume stays for the per-slice regression)
                aff = [ops.depth_range_affine(depth[b].contiguous(), D, self.depth_intervals_ratio[s] * depth_interval)
                       for b in range(B)] if AFFINE_SWEEP else None
            out = self.DepthNet[s](feats, proj_matrices[key], depth_values=dv, num_depth=D,
                                   confidence_map=pair_confidence, sweep_depth=None if depth is None else aff)
            depth = out["depth"]
            pair_confidence = out["pair_confidence"]
            outputs[key] = out
            outputs.update(out)
        return outputs
