"""Cascade RED-Net inference (reference: mvs/mvs_cas/models/msrednet.py:337-527) on the HIP plane-sweep engine.

Same constructor, forward() contract and state_dict keys as the reference's Infer_CascadeREDNet.  Per depth
plane: fused warp + variance (d3d_variance_volume, D = 1) -> slice_RED_Regularization (four GroupNorm conv-GRUs
in a 2D encoder-decoder, on the matrix-core convolutions) -> online exp-sum regression.  Inference only.
"""
import torch
import torch.nn as nn

from .dataset import extract_features
from . import ops
from . import config as _cfg
from .module import ConvGRUCell2, ConvReLU, ConvTransReLU, FeatureNet_mvsnet, plane_depths


class slice_RED_Regularization(nn.Module):
    """msrednet.py:337-370.  forward(cost [C,h,w], state1..4) -> (reg [1,h,w], state1..4); with regress = (dplane, max_p, sum_d,
    sum_p) the online regression update of the slice is applied as well (reg is None when it was fused into the head).
    The reference feeds -cost to conv1 and conv_gru1; here the sign is folded into their weights."""

    def __init__(self, in_channels, base_channels=8):
        super().__init__()
        b = base_channels
        self.base_channels = b
        self.conv_gru1 = ConvGRUCell2(in_channels, b, 3)
        self.conv_gru2 = ConvGRUCell2(b * 2, b * 2, 3)
        self.conv_gru3 = ConvGRUCell2(b * 4, b * 4, 3)
        self.conv_gru4 = ConvGRUCell2(b * 8, b * 8, 3)
        self.conv1 = ConvReLU(in_channels, b * 2, 3, 2, 1)
        self.conv2 = ConvReLU(b * 2, b * 4, 3, 2, 1)
        self.conv3 = ConvReLU(b * 4, b * 8, 3, 2, 1)
        self.upconv3 = ConvTransReLU(b * 8, b * 4, 3, 2, 1, 1)
        self.upconv2 = ConvTransReLU(b * 4, b * 2, 3, 2, 1, 1)
        self.upconv1 = ConvTransReLU(b * 2, b, 3, 2, 1, 1)
        self.upconv2d = nn.ConvTranspose2d(b, 1, kernel_size=3, stride=1, padding=1, output_padding=0)


    def _head(self, up11, regress):
        """upconv2d: ConvTranspose2d(stride 1, pad 1) = correlation with the spatially flipped, transposed kernel.  With
        regress = (dplane, max_p, sum_d, sum_p) the layer and the online regression update of msrednet.py:418-437 are one kernel
        where it applies (h16 mode: ops.slice_head_regress) and None is returned: `reg` never reaches memory."""
        wc = ops.derived_weight(self.upconv2d.weight, "flipT", lambda w: w.flip(2, 3).transpose(0, 1))
        if regress is not None:
            dplane, max_p, sum_d, sum_p = regress
            if ops.slice_head_regress(up11, wc, self.upconv2d.bias, False, dplane, max_p, sum_d, sum_p):
                return None
        reg = ops.conv2d_k3(up11, wc, None, self.upconv2d.bias, None, act=0)
        if regress is not None:
            ops.online_regress_update(reg[0], *regress)
        return reg

    def _tail(self, up22, state1, regress):
        """upconv1 + skip, upconv2d and -- with `regress` -- the regression update: one kernel where it applies (h16 mode,
        ops.slice_tail_regress_same: `up11` and `reg` never reach memory; returns None), else the layers one by one."""
        if regress is not None:
            dplane, max_p, sum_d, sum_p = regress
            wc = ops.derived_weight(self.upconv2d.weight, "flipT", lambda w: w.flip(2, 3).transpose(0, 1))
            if ops.slice_tail_regress_same(up22, self.upconv1.conv.weight, None, state1, True, wc, self.upconv2d.bias, dplane, max_p, sum_d, sum_p):
                return None
        return self._head(self.upconv1(up22, skip=state1), regress)

    def encode_all(self, var):
        """The encoder of EVERY depth slice of a stage in three batched launches (msrednet.py:352-356: conv1(-cost), conv2, conv3 depend
        on the cost slices only, not on the recurrent state): var [D,C,h,w] -> (c1 [D,16,h/2,w/2], c2 [D,32,..], c3 [D,64,..]), slice
        d of each bit for bit what forward() computes itself; None where the batched kernel does not apply (h16 mode only).  Takes the
        three small convolutions out of every slice's chain -- 3 of 10 serial launches per slice -- and runs them on full grids."""
        # (conv3's input -- a sixteenth of the slice -- must still be a map the per-slice dispatch gives to the tile kernel, ops.conv2d_k3:
        #  the batched form runs that kernel at any size, and smaller maps would no longer be the per-slice forward bit for bit)
        if not var.is_cuda or _cfg.off("red_encoder") or (var.shape[2] // 4) * (var.shape[3] // 4) < 128 * 128:
            return None
        w1 = ops.derived_weight(self.conv1.conv.weight, "neg", lambda w: -w)
        c1 = ops.conv2d_s2_zs_batched(var, w1, act=1)
        c2 = None if c1 is None else ops.conv2d_s2_zs_batched(c1, self.conv2.conv.weight, act=1)
        c3 = None if c2 is None else ops.conv2d_s2_zs_batched(c2, self.conv3.conv.weight, act=1)
        return None if c3 is None else (c1, c2, c3)

    def forward(self, cost, state1, state2, state3, state4, regress=None, enc=None):
        """enc: (c1, c2, c3) of this slice from encode_all(), or None (the encoder runs here)."""
        w1 = ops.derived_weight(self.conv1.conv.weight, "neg", lambda w: -w)
        if enc is not None and cost.is_cuda and not _cfg.off("red_streams"):
            # encoder precomputed: the three side levels start at once, level 4 and the decoder follow on the caller's stream
            c1, c2, c3 = enc
            main = torch.cuda.current_stream(cost.device)
            side = ops.side_streams(cost.device, 3, "red")
            e0 = main.record_event()                                                  # (the states of the previous slice exist)
            news = []
            for st, cell, x, h, neg in ((side[2], self.conv_gru1, cost, state1, True), (side[0], self.conv_gru2, c1, state2, False),
                                        (side[1], self.conv_gru3, c2, state3, False)):
                with torch.cuda.stream(st):
                    st.wait_event(e0)
                    hn, _ = cell(x, h, negate_x=True) if neg else cell(x, h)
                    news.append((hn, st.record_event()))
                ops.hand_over(hn, main)
            (state1, d1), (state2, d2), (state3, d3) = news
            state4, _ = self.conv_gru4(c3, state4)
            main.wait_event(d3)
            up33 = self.upconv3(state4, skip=state3)
            main.wait_event(d2)
            up22 = self.upconv2(up33, skip=state2)
            main.wait_event(d1)
            return self._tail(up22, state1, regress), state1, state2, state3, state4
        if enc is not None:
            c1, c2, c3 = enc
            state4, _ = self.conv_gru4(c3, state4)
            state3, _ = self.conv_gru3(c2, state3)
            up33 = self.upconv3(state4, skip=state3)
            state2, _ = self.conv_gru2(c1, state2)
            up22 = self.upconv2(up33, skip=state2)
            state1, _ = self.conv_gru1(cost, state1, negate_x=True)
            return self._tail(up22, state1, regress), state1, state2, state3, state4
        if cost.is_cuda and not _cfg.off("red_streams"):
            # The four recurrent cells of a slice depend on the encoder's maps only (msrednet.py:352-367), and at the first two
            # stages their kernels are tens of workgroups each: level 1 starts on a side stream beside the encoder, levels 2 / 3
            # on two more as soon as their encoder map exists, level 4 follows the encoder on the caller's stream; the decoder
            # joins them.  Same kernels on the same operands; every tensor that crosses streams is ordered by an event, and a
            # block freed on a side stream is reused there only behind the next slice's fork, i.e. behind this slice's decoder.
            main = torch.cuda.current_stream(cost.device)
            side = ops.side_streams(cost.device, 3, "red")   # three per caller stream: conv-GRU levels 2, 3 and 1 of a slice
            e0 = main.record_event()                                                  # (the cost slice and the states exist)
            with torch.cuda.stream(side[2]):                                          # level 1 -- the largest -- beside the encoder
                side[2].wait_event(e0)
                state1, _ = self.conv_gru1(cost, state1, negate_x=True)               # conv_gru1(-cost)
                d1 = side[2].record_event()
            ops.hand_over(state1, main)
            c1 = ops.conv2d_k3(cost, w1, None, None, None, act=1, stride=2)           # conv1(-cost)
            e1 = main.record_event()
            with torch.cuda.stream(side[0]):
                side[0].wait_event(e1)
                state2, _ = self.conv_gru2(c1, state2)
                d2 = side[0].record_event()
            ops.hand_over(state2, main)
            c2 = self.conv2(c1)
            e2 = main.record_event()
            with torch.cuda.stream(side[1]):
                side[1].wait_event(e2)
                state3, _ = self.conv_gru3(c2, state3)
                d3 = side[1].record_event()
            ops.hand_over(state3, main)
            c3 = self.conv3(c2)
            state4, _ = self.conv_gru4(c3, state4)
            main.wait_event(d3)
            up33 = self.upconv3(state4, skip=state3)                                  # relu(convT) + reg_cost3
            main.wait_event(d2)
            up22 = self.upconv2(up33, skip=state2)
            main.wait_event(d1)
            return self._tail(up22, state1, regress), state1, state2, state3, state4
        c1 = ops.conv2d_k3(cost, w1, None, None, None, act=1, stride=2)             # conv1(-cost)
        c2 = self.conv2(c1)
        c3 = self.conv3(c2)
        state4, _ = self.conv_gru4(c3, state4)
        state3, _ = self.conv_gru3(c2, state3)
        up33 = self.upconv3(state4, skip=state3)                                      # relu(convT) + reg_cost3
        state2, _ = self.conv_gru2(c1, state2)
        up22 = self.upconv2(up33, skip=state2)
        state1, _ = self.conv_gru1(cost, state1, negate_x=True)                       # conv_gru1(-cost)
        return self._tail(up22, state1, regress), state1, state2, state3, state4


class InferDepthNet(nn.Module):
    """msrednet.py:373-438: plane-by-plane variance cost, recurrent regularisation, online regression."""

    def forward(self, features, proj_matrices, depth_values, num_depth, cost_regularization):
        V = len(features)
        assert V == proj_matrices.shape[1], "Different number of images and projection matrices"
        assert depth_values.shape[1] == num_depth, "depth_values.shape[1]:{}  num_depth:{}".format(
            depth_values.shape[1], num_depth)
        B, C, h, w = features[0].shape
        if h % 8 or w % 8:
            raise ValueError("feature maps must be divisible by 8 (three stride-2 levels), got %dx%d" % (h, w))
        depths, confs = [], []
        for b in range(B):
            feats = [f[b].contiguous() for f in features]
            p34 = ops.compose_projections(proj_matrices[b].contiguous())
            dvb = depth_values[b].contiguous()
            dev = feats[0].device
            loop = RedLoopGraph.get(cost_regularization, C, h, w, num_depth, dvb)
            if loop is not None and loop.usable():   # the slice loop as one captured HIP graph (RedLoopGraph)
                ops.variance_volume(feats, p34, dvb, out=loop.buffers().var, plane_major=True)
                dep, conf = ops.online_regress_finalize(*loop.run(dvb))
                depths.append(dep)
                confs.append(conf)
                continue
            states = [torch.zeros((8 << i, h >> i, w >> i), dtype=torch.float32, device=dev) for i in range(4)]
            max_p = torch.zeros((h, w), dtype=torch.float32, device=dev)
            sum_d = torch.zeros_like(max_p)
            sum_p = torch.zeros_like(max_p)
            # The reference warps plane by plane (msrednet.py:400-414) because it holds one slice at a time; the
            # whole variance volume of a stage is at most 2.6 GB here, so it is swept in ONE fused launch and the
            # recurrent regulariser then walks its depth slices.
            var = ops.variance_volume(feats, p34, dvb, plane_major=True)  # [D,C,h,w]: a slice is one contiguous block
            enc = cost_regularization.encode_all(var)   # (the encoder of every slice in three batched launches; None: inside the slices)
            for d in range(num_depth):
                if dvb.dim() == 1:   # [D] uniform planes: a [1,1] map (the update resamples it to the image, a constant)
                    dplane = dvb[d:d + 1].view(1, 1)
                else:                # [D,h,w] per-pixel hypotheses
                    dplane = dvb[d]
                _, *states = cost_regularization(var[d], *states, regress=(dplane, max_p, sum_d, sum_p),
                                                 enc=None if enc is None else tuple(c[d] for c in enc))
            dep, conf = ops.online_regress_finalize(max_p, sum_d, sum_p)
            depths.append(dep)
            confs.append(conf)
        return {"depth": torch.stack(depths), "photometric_confidence": torch.stack(confs)}


class RedLoopGraph(object):
    """The slice loop of one RED-Net cascade stage (msrednet.py:400-437: D times the four-level conv-GRU encoder-decoder and the
    online regression update, ~23 launches per slice) captured once as a HIP graph and replayed per reference view.

    Why: a RED-Net view is ~2 000 launches of 10 - 200 us kernels on four streams per slice.  On ONE stream the card paces the loop
    (launch loop and graph take the same time); what the host cannot do is issue four streams' launches and event operations fast
    enough -- the four-stream slices are 6 ms per view faster captured than launched.  Captured, the host issues one
    graph launch per stage.  What is captured is the forward as it runs eagerly, including the four-stream form of a slice
    (slice_RED_Regularization.forward: ops.side_streams / events / ops.hand_over -- tensors that cross streams are recorded, which
    keeps the capture's allocator from handing their blocks to another stream before the capture ends), the GroupNorm slot arenas
    (created and zeroed INSIDE the capture, per captured stream: every replay zeroes them again) and the zeroing of the states and
    accumulators.  Static inputs: the variance volume [D,C,h,w] (the sweep writes it in place) and the hypotheses (copied in).
    Same kernels, same operands as the eager loop; equal to it to the order of the fp64 atomics of the GroupNorm statistics
    (tests/test_parity_gpu.py::test_msrednet_loop_graph_is_the_eager_loop).  Same protocol as adamvs.SliceLoopGraph: first call of
    a shape eager, second captures (after one eager slice that prepares what first-use code would), main thread only, a failed
    capture raises (D3D_KERNELS_OFF=red_graph).

    Measured and not kept (profiles/r05_red_pipeline_ab.txt): the D slices as six chains across the slices (encoder of slice d + 2,
    the four levels' cells of d + 1, the decoder of d in flight together, 2-dependency nodes only) -- stage 1 14.7 -> 17.0 ms, stages
    2 / 3 unchanged.  Stage 1 replays 1 584 nodes in 14.7 ms, 9 us per node whatever the branches: the node rate bounds it, not the
    slices' dependency chain; stages 2 / 3 are bound by their kernels' durations.  The lighter variant -- only the decoder of slice d
    on its own stream beside the encoder and cells of slice d + 1 -- is slower as well (58.3 -> 63.3 ms per view): more branches in
    flight cost this graph executor more than the overlap returns.  With the encoder hoisted out of the loop (encode_all) the four
    conv-GRU levels are four independent recurrences; as four free-running chains they take 62.3 ms captured and 55.5 ms as launches
    against 53.2 ms for the fork-join slices captured: the card is saturated by the kernels themselves (their durations add up to the
    view), not waiting on the slices' dependency chain."""

    _cache = {}
    _lock = __import__("threading").Lock()
    MAX_GRAPHS = 12

    @classmethod
    def get(cls, cr, C, h, w, D, dvb):
        import threading

        if (not dvb.is_cuda or ops.conv_precision() != "h16" or _cfg.off("red_graph") or D < 4 or torch.cuda.is_current_stream_capturing()
                or threading.current_thread() is not threading.main_thread()):
            return None
        dev = dvb.device
        key = (id(cr), C, h, w, D, tuple(dvb.shape), dev.index, torch.cuda.current_stream(dev).cuda_stream, ops.h16_dtype(),
               tuple(sorted(_cfg.switches.items())))
        with cls._lock:
            for k in [k for k, v in cls._cache.items() if v.cr() is None]:
                del cls._cache[k]
            g = cls._cache.get(key)
            if g is None or g.cr() is not cr:
                if len(cls._cache) >= cls.MAX_GRAPHS:
                    cls._cache.pop(next(iter(cls._cache)))
                g = cls._cache[key] = cls(cr, C, h, w, D, dvb)
        return g

    def __init__(self, cr, C, h, w, D, dvb):
        import weakref

        self.cr = weakref.ref(cr)
        self.C, self.h, self.w, self.D, self.dev = C, h, w, D, dvb.device
        self.dv_shape = tuple(dvb.shape)
        self.calls, self.graph, self.counts, self.wkey, self.failed, self.var = 0, None, None, None, None, None

    def usable(self):
        k = tuple((p.data_ptr(), p._version) for p in self.cr().parameters())
        if k != self.wkey:
            self.wkey, self.calls, self.graph = k, 0, None
        self.calls += 1
        return self.calls >= 2 and self.failed is None

    def buffers(self):
        if self.var is None:
            f32, dev = torch.float32, self.dev
            self.var = torch.empty((self.D, self.C, self.h, self.w), dtype=f32, device=dev)
            self.dv = torch.empty(self.dv_shape, dtype=f32, device=dev)
            self.max_p = torch.empty((self.h, self.w), dtype=f32, device=dev)
            self.sum_d, self.sum_p = torch.empty_like(self.max_p), torch.empty_like(self.max_p)
            self.stream = torch.cuda.Stream(dev)
            with torch.cuda.stream(self.stream):     # the captured forward's side streams exist before the capture begins
                self.side = ops.side_streams(dev, 3, "red")
        return self

    def _dplane(self, d):
        return self.dv[d:d + 1].view(1, 1) if self.dv.dim() == 1 else self.dv[d]

    def _zero_states(self):
        return [torch.zeros((8 << i, self.h >> i, self.w >> i), dtype=torch.float32, device=self.dev) for i in range(4)]

    def _capture(self):
        cr = self.cr()
        before = dict(ops.dispatch_counts)
        states = self._zero_states()          # one eager slice first: packed weights, LDS attributes, slot arenas of the eager streams
        enc = cr.encode_all(self.var)
        _, *states = cr(self.var[0], *states, regress=(self._dplane(0), self.max_p, self.sum_d, self.sum_p),
                        enc=None if enc is None else tuple(c[0] for c in enc))
        del states, enc
        ops.dispatch_counts.clear()
        ops.dispatch_counts.update(before)
        graph = torch.cuda.CUDAGraph()
        self.stream.wait_stream(torch.cuda.current_stream(self.dev))
        for st in [self.stream] + list(self.side):   # the captured streams' GroupNorm slot arenas are born (and zeroed) in the capture
            ops._gn_arenas.pop((self.dev.index, st.cuda_stream), None)
        with torch.cuda.graph(graph, stream=self.stream, capture_error_mode="thread_local"):
            states = self._zero_states()
            self.max_p.zero_(); self.sum_d.zero_(); self.sum_p.zero_()
            enc = cr.encode_all(self.var)   # (inside the capture: three nodes in front of the slices)
            for d in range(self.D):
                _, *states = cr(self.var[d], *states, regress=(self._dplane(d), self.max_p, self.sum_d, self.sum_p),
                                enc=None if enc is None else tuple(c[d] for c in enc))
            del states, enc
        after = dict(ops.dispatch_counts)
        self.counts = {k: v - before.get(k, 0) for k, v in after.items() if v != before.get(k, 0)}
        for k, v in self.counts.items():
            ops.dispatch_counts[k] -= v
        self.graph = graph

    def run(self, dvb):
        if self.graph is None:
            with RedLoopGraph._lock:
                try:
                    self._capture()
                except Exception as e:
                    self.failed = repr(e)[:300]
                    raise RuntimeError("capturing RED-Net's slice loop as a HIP graph failed (%s); run with D3D_KERNELS_OFF=red_graph"
                                       % self.failed) from e
                torch.cuda.current_stream(self.dev).wait_stream(self.stream)
        self.dv.copy_(dvb)
        self.graph.replay()
        for k, v in self.counts.items():
            ops.dispatch_counts[k] += v
        return self.max_p, self.sum_d, self.sum_p


class Infer_CascadeREDNet(nn.Module):
    """msrednet.py:442-527."""

    def __init__(self, num_depth=384, ndepths=[48, 32, 8], depth_intervals_ratio=[4, 2, 1], share_cr=False,
                 cr_base_chs=[8, 8, 8]):
        super().__init__()
        assert len(ndepths) == len(depth_intervals_ratio)
        self.num_depth, self.share_cr, self.ndepths = num_depth, share_cr, list(ndepths)
        self.depth_intervals_ratio, self.cr_base_chs = list(depth_intervals_ratio), list(cr_base_chs)
        self.num_stage = len(ndepths)
        self.stage_infos = {"stage1": {"scale": 4.0}, "stage2": {"scale": 2.0}, "stage3": {"scale": 1.0}}
        self.feature = FeatureNet_mvsnet(base_channels=8, stride=4, num_stage=self.num_stage, arch_mode="unet")
        if share_cr:
            # the reference passes the channel LIST here (msrednet.py:467), which cannot construct; share_cr
            # is therefore only meaningful with equal channels -- use the stage-1 width like cas_mvsnet.py
            self.cost_regularization = slice_RED_Regularization(self.feature.out_channels[0], 8)
        else:
            self.cost_regularization = nn.ModuleList(
                [slice_RED_Regularization(self.feature.out_channels[i], self.cr_base_chs[i])
                 for i in range(self.num_stage)])
        self.DepthNet = InferDepthNet()

    feature_cache = None  # dataset.FeatureCache shared across reference views (set by the harness); see image_keys

    def forward(self, imgs, proj_matrices, depth_values, image_keys=None):
        """image_keys (optional, with self.feature_cache set): one hashable key per view; the feature pyramid of a
        key seen before is reused instead of recomputed, and imgs may then be a list whose cached entries are None."""
        if self.training:
            raise RuntimeError("inference only: call .eval()")
        dmin, dmax = ops.depth_range_host(depth_values)   # (host numbers: no device access when the caller noted them)
        depth_interval = (dmax - dmin) / self.num_depth
        features = extract_features(self.feature, imgs, image_keys, self.feature_cache)
        V = len(features)
        B, _, img_h, img_w = features[0]["stage3"].shape  # the finest level has the image's size
        outputs = {}
        depth = None
        for s in range(self.num_stage):
            key = "stage%d" % (s + 1)
            feats = [f[key] for f in features]
            scale = int(self.stage_infos[key]["scale"])
            h, w = img_h // scale, img_w // scale
            D = self.ndepths[s]
            if depth is None:
                dv = plane_depths(depth_values, D)                              # [B,D]: constant planes
            else:
                # msrednet.py:497-516: depth -> full res (bilinear), hypotheses at full res, trilinear resample
                dvs = []
                for b in range(B):
                    cur = ops.resize_bilinear(depth[b:b + 1].contiguous(), img_h, img_w)[0]
                    full = ops.depth_range_samples(cur, D, self.depth_intervals_ratio[s] * depth_interval)
                    dvs.append(full if (h, w) == (img_h, img_w) else ops.resize_bilinear(full, h, w))
                dv = torch.stack(dvs)
            cr = self.cost_regularization if self.share_cr else self.cost_regularization[s]
            with torch.no_grad():
                out = self.DepthNet(feats, proj_matrices[key], depth_values=dv, num_depth=D, cost_regularization=cr)
            depth = out["depth"]
            outputs[key] = out
            outputs.update(out)
        return outputs
