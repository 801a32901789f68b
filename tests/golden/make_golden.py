#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ by RUNNING THE REFERENCE on CPU.

Run only in the build container, where the reference checkout is mounted read-only at
/root/reference (it does not exist on the GPU box):

    python tests/golden/make_golden.py

The reference's model files import torch only, so they are imported in
place (nothing is copied; sys.dont_write_bytecode keeps the read-only tree clean).  The
.npz files hold data only: seeded inputs, the reference's outputs, and -- instead of
checkpoints -- the seed from which deep3d_aerial_amd.synthetic.fill_state_dict_ rebuilds
the exact weights for any module with the same state_dict keys.

Infer_AdaMVSNet hard-codes .cuda() on its state tensors; on this CPU-only
container torch.Tensor.cuda is shimmed to the identity for the duration of the run.
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("D3D_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(REF, "mvs", "mvs_cas"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from deep3d_aerial_amd import synthetic as S  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self  # F6 shim (CPU run)
torch.set_num_threads(8)

from models import module as RM  # noqa: E402
from models import adamvs as RA  # noqa: E402
from models import cas_mvsnet as RC  # noqa: E402
from models import msrednet as RR  # noqa: E402
from models import ucsnet as RU  # noqa: E402

T = torch.from_numpy


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %-28s %8.1f KiB" % (name + ".npz", os.path.getsize(path) / 1024))


def composed(proj):
    """module.py:528 exactly: matmul(src_proj, inverse(ref_proj)) -> [V-1,3,4]."""
    P = T(proj)
    out = [torch.matmul(P[i:i + 1], torch.inverse(P[0:1]))[0, :3, :4].numpy() for i in range(1, P.shape[0])]
    return np.stack(out).astype(np.float32)


def hand_projs(h, w):
    """identity / pure x-shift / oblique with partial out-of-frame -- SURVEY.md 8c (i)."""
    K = np.array([[1.5 * w, 0, (w - 1) / 2], [0, 1.5 * w, (h - 1) / 2], [0, 0, 1]], np.float64)
    ref = np.eye(4)
    ref[:3, :3] = K
    out = [ref]
    # identity
    out.append(ref.copy())
    # pure x translation
    P = np.eye(4)
    P[:3, :3] = K
    P[:3, 3] = K @ np.array([40.0, 0, 0])
    out.append(P)
    # oblique rotation + translation, partly out of frame
    R = S._rot(0.12, -0.2, 0.3)
    P = np.eye(4)
    P[:3, :3] = K @ R
    P[:3, 3] = K @ np.array([-60.0, 35.0, 20.0])
    out.append(P)
    return np.stack(out).astype(np.float32)


# --------------------------------------------------------------------------------------
def gen_warp():
    rng = np.random.default_rng(100)
    out = {}
    n = 0
    for (C, h, w, D) in [(4, 12, 10, 1), (4, 12, 10, 5), (32, 24, 20, 5)]:
        projs = hand_projs(h, w)
        scene, dv = S.make_scene(3, h, w, 8, sweep_px=6.0, seed=n)
        all_projs = np.concatenate([projs, scene[1:]], 0)  # index 0 = ref of hand set
        src = rng.standard_normal((C, h, w), dtype=np.float32)
        for depth_kind in ("vec", "map"):
            if depth_kind == "vec":
                depth = S.uniform_depths([400, 800], max(D, 2))[:D].copy()
                dt = T(depth)[None]
            else:
                depth = (600 + 150 * rng.standard_normal((D, h, w))).astype(np.float32)
                dt = T(depth)[None]
            for pi in range(1, all_projs.shape[0]):
                ref_p = all_projs[0] if pi < projs.shape[0] else scene[0]
                src_p = all_projs[pi]
                y = RM.homo_warping_float(T(src)[None], T(src_p)[None], T(ref_p)[None], dt)[0].numpy()
                pc = composed(np.stack([ref_p, src_p]))[0]
                k = "c%d_" % n
                out[k + "src"] = src
                out[k + "ref_proj"] = ref_p
                out[k + "src_proj"] = src_p
                out[k + "proj34"] = pc
                out[k + "depth"] = depth
                out[k + "out"] = y
                n += 1
    out["n_cases"] = np.array(n)
    save("ops_warp", **out)


def gen_aggregate():
    """variance (cas_mvsnet.py:45-60), pair channel-mean (adamvs.py:469-474), weighted corr (adamvs.py:492-509)."""
    rng = np.random.default_rng(200)
    out = {}
    n = 0
    for (V, C, h, w, D, depth_kind) in [(2, 4, 12, 10, 3, "vec"), (3, 8, 16, 12, 4, "map"), (5, 32, 24, 20, 6, "vec"),
                                        (5, 16, 20, 24, 5, "map")]:
        proj, dv = S.make_scene(V, h, w, D, sweep_px=5.0, seed=10 + n, yaw_deg=4.0)
        feats = S.make_features(V, C, h, w, seed=20 + n)
        if depth_kind == "vec":
            depth = S.uniform_depths(dv, D)
            dt = T(depth)[None]
        else:
            depth = (600 + 120 * rng.standard_normal((D, h, w))).astype(np.float32)
            dt = T(depth)[None]
        ft = [T(feats[v])[None] for v in range(V)]
        pt = [T(proj[v])[None] for v in range(V)]
        # variance, eval-mode arithmetic of cas_mvsnet.DepthNet
        ref_volume = ft[0].unsqueeze(2).repeat(1, 1, D, 1, 1)
        vs = ref_volume
        vq = ref_volume ** 2
        pair = []
        warped_all = []
        for i in range(1, V):
            wv = RM.homo_warping_float(ft[i], pt[i], pt[0], dt)
            warped_all.append(wv.clone())
            pair.append((ft[0].unsqueeze(2) * wv).mean(dim=1)[0].numpy())
            vs = vs + wv
            vq = vq + wv.pow(2)
        var = vq.div_(V).sub_(vs.div_(V).pow_(2))[0].numpy()
        # weighted correlation with random positive weights, one plane at a time as adamvs does
        vw = rng.uniform(0.02, 1.0, (V - 1, h, w)).astype(np.float32)
        sim = []
        for d in range(D):
            ssum = 0
            wsum = 1e-5
            refv = ft[0].unsqueeze(2)
            for i in range(1, V):
                wv = warped_all[i - 1][:, :, d:d + 1]
                w2 = wv * refv
                vwi = T(vw[i - 1])[None, None]
                ssum = ssum + w2 * vwi.unsqueeze(1)
                wsum = wsum + vwi.unsqueeze(1)
            sim.append((ssum / wsum)[0, :, 0].numpy())
        sim = np.stack(sim, 1)
        k = "c%d_" % n
        out[k + "feats"] = feats
        out[k + "proj"] = proj
        out[k + "proj34"] = composed(proj)
        out[k + "depth"] = depth
        out[k + "variance"] = var
        out[k + "pair_mean"] = np.stack(pair)
        out[k + "weights"] = vw
        out[k + "weighted"] = sim
        n += 1
    out["n_cases"] = np.array(n)
    save("ops_aggregate", **out)


def gen_regress():
    """softmax/soft-argmin/conf4 (cas_mvsnet.py:69-76), online exp-sum (adamvs.py:514-529),
    depth hypotheses (module.py:616-650), bilinear x2 (adamvs.py:519-520)."""
    rng = np.random.default_rng(300)
    out = {}
    n = 0
    for (D, h, w, kind) in [(8, 6, 5, "flat"), (48, 8, 6, "peaked"), (32, 7, 9, "edge")]:
        cost = rng.standard_normal((D, h, w)).astype(np.float32)
        if kind == "peaked":
            idx = rng.integers(0, D, (h, w))
            for y in range(h):
                for x in range(w):
                    cost[idx[y, x], y, x] += 12.0
        if kind == "edge":
            cost[0, :, : w // 2] += 15.0
            cost[D - 1, :, w // 2:] += 15.0
        depth_map = np.sort(500 + 80 * rng.standard_normal((D, h, w)), 0).astype(np.float32)
        for dk in ("vec", "map"):
            dvals = S.uniform_depths([400, 800], D) if dk == "vec" else depth_map
            prob = F.softmax(T(cost)[None], dim=1)
            dt = T(dvals)[None]
            depth = RM.depth_regression(prob, depth_values=dt)
            s4 = 4 * F.avg_pool3d(F.pad(prob.unsqueeze(1), pad=(0, 0, 0, 0, 1, 2)), (4, 1, 1), stride=1,
                                  padding=0).squeeze(1)
            di = RM.depth_regression(prob, depth_values=torch.arange(D, dtype=torch.float)).long().clamp(0, D - 1)
            conf = torch.gather(s4, 1, di.unsqueeze(1)).squeeze(1)
            k = "sa%d_" % n
            out[k + "cost"] = cost
            out[k + "depth_values"] = dvals
            out[k + "depth"] = depth[0].numpy()
            out[k + "conf"] = conf[0].numpy()
            n += 1
    out["n_softargmin"] = np.array(n)

    # online regression: 8 planes, with and without the x2 depth upsample, incl. a peaked case
    m = 0
    for (D, h, w, up, peaked) in [(8, 6, 5, False, False), (8, 6, 5, True, True), (5, 4, 7, True, False)]:
        H, W = (2 * h, 2 * w) if up else (h, w)
        reg = rng.standard_normal((D, H, W)).astype(np.float32)
        if peaked:
            reg[3] += 6.0
        dpl = np.sort(500 + 80 * rng.standard_normal((D, h, w)), 0).astype(np.float32)
        exp_sum = torch.zeros(1, 1, H, W)
        depth_image = torch.zeros(1, 1, H, W)
        max_prob = torch.zeros(1, 1, H, W)
        ups = []
        for d in range(D):
            prob = T(reg[d])[None, None].exp()
            flag = (max_prob < prob).float()
            new_max = flag * prob + (1 - flag) * max_prob
            dv = T(dpl[d])[None, None]
            if up:
                dv = F.interpolate(dv, [H, W], mode="bilinear", align_corners=False)
            ups.append(dv[0, 0].numpy())
            depth_image = dv * prob + depth_image
            max_prob = new_max
            exp_sum = exp_sum + prob
        fes = exp_sum + 1e-10
        k = "on%d_" % m
        out[k + "reg"] = reg
        out[k + "dplanes"] = dpl
        out[k + "dplanes_up"] = np.stack(ups)
        out[k + "up"] = np.array(int(up))
        out[k + "depth"] = (depth_image / fes)[0, 0].numpy()
        out[k + "conf"] = (max_prob / fes)[0, 0].numpy()
        m += 1
    out["n_online"] = np.array(m)

    # depth hypotheses, both branches
    s0 = RM.get_depth_range_samples(T(np.array([[400.0, 500.0]], np.float32)), 4, 0.0, "cpu", torch.float32,
                                    [1, 3, 2])
    out["dr0_cur"] = np.array([400.0, 500.0], np.float32)
    out["dr0_out"] = s0[0].numpy()
    cur = (600 + 50 * rng.standard_normal((5, 4))).astype(np.float32)
    s1 = RM.get_depth_range_samples(T(cur)[None], 8, 2.5 * 1.0416666, "cpu", torch.float32, [1, 5, 4])
    out["dr1_cur"] = cur
    out["dr1_interval"] = np.array(2.5 * 1.0416666, np.float32)
    out["dr1_out"] = s1[0].numpy()
    save("ops_regress", **out)


def gen_gru():
    """SliceCostRegNetRED one step + rollout (adamvs.py:403-427, module.py:5-51)."""
    out = {}
    n = 0
    for (C, h, w, up, steps) in [(8, 8, 12, True, 1), (16, 8, 8, False, 4), (32, 12, 8, True, 8)]:
        rng = np.random.default_rng(400 + n)
        net = RA.SliceCostRegNetRED(C, up, 8).eval()
        S.fill_state_dict_(net.state_dict(), 4000 + n)
        costs = rng.standard_normal((steps, C, h, w)).astype(np.float32)
        s1 = torch.zeros(1, 8, h, w)
        s2 = torch.zeros(1, 16, h // 2, w // 2)
        regs = []
        with torch.no_grad():
            for t in range(steps):
                r, s1, s2 = net(T(costs[t])[None], s1, s2)
                regs.append(r[0].numpy())
        k = "c%d_" % n
        out[k + "costs"] = costs
        out[k + "up"] = np.array(int(up))
        out[k + "seed"] = np.array(4000 + n)
        out[k + "regs"] = np.stack(regs)
        out[k + "state1"] = s1[0].numpy()
        out[k + "state2"] = s2[0].numpy()
        n += 1
    out["n_cases"] = np.array(n)
    save("ops_gru", **out)


def gen_gru2():
    """slice_RED_Regularization rollouts (msrednet.py:337-370) with GroupNorm conv-GRUs (module.py:53-99)."""
    out = {}
    n = 0
    for (C, h, w, steps) in [(8, 8, 16, 1), (32, 16, 8, 5), (16, 8, 8, 3)]:
        rng = np.random.default_rng(450 + n)
        net = RR.slice_RED_Regularization(C, 8).eval()
        S.fill_state_dict_(net.state_dict(), 4500 + n)
        costs = np.abs(rng.standard_normal((steps, C, h, w))).astype(np.float32)
        st = [torch.zeros(1, 8 << i, h >> i, w >> i) for i in range(4)]
        regs = []
        with torch.no_grad():
            for t in range(steps):
                r, *st = net(T(costs[t])[None], *st)
                regs.append(r[0].numpy())
        k = "c%d_" % n
        out[k + "costs"] = costs
        out[k + "seed"] = np.array(4500 + n)
        out[k + "regs"] = np.stack(regs)
        for i in range(4):
            out[k + "state%d" % (i + 1)] = st[i][0].numpy()
        n += 1
    out["n_cases"] = np.array(n)
    # one bare cell, non-zero initial state
    rng = np.random.default_rng(460)
    cell = RM.ConvGRUCell2(8, 8, 3).eval()
    S.fill_state_dict_(cell.state_dict(), 4600)
    x = rng.standard_normal((8, 6, 10)).astype(np.float32)
    h0 = rng.standard_normal((8, 6, 10)).astype(np.float32)
    with torch.no_grad():
        h1, _ = cell(T(x)[None], T(h0)[None])
    out["cell_x"], out["cell_h0"], out["cell_h1"], out["cell_seed"] = x, h0, h1[0].numpy(), np.array(4600)
    save("ops_gru2", **out)


def gen_costreg3d():
    """CostRegNet 3D UNet, eval-mode BN (cas_mvsnet.py:81-121)."""
    out = {}
    n = 0
    for (C, D, h, w) in [(8, 8, 16, 16), (32, 16, 16, 16), (16, 8, 8, 24)]:
        rng = np.random.default_rng(500 + n)
        net = RC.CostRegNet(C, 8).eval()
        S.fill_state_dict_(net.state_dict(), 5000 + n)
        x = np.abs(rng.standard_normal((C, D, h, w))).astype(np.float32)
        with torch.no_grad():
            y = net(T(x)[None])[0].numpy()
        k = "c%d_" % n
        out[k + "x"] = x
        out[k + "seed"] = np.array(5000 + n)
        out[k + "y"] = y
        n += 1
    out["n_cases"] = np.array(n)
    save("ops_costreg3d", **out)


def gen_pairnet():
    """CostRegNet2D pair-visibility UNet (adamvs.py:198-238) + softmax/max/regression (adamvs.py:477-486)."""
    out = {}
    rng = np.random.default_rng(600)
    D, h, w = 48, 16, 24
    net = RA.CostRegNet2D(D, 8).eval()
    S.fill_state_dict_(net.state_dict(), 6000)
    x = (0.3 * rng.standard_normal((D, h, w))).astype(np.float32)
    dvals = np.tile(S.uniform_depths([400, 800], D)[:, None, None], (1, h, w)).astype(np.float32)
    with torch.no_grad():
        score = net(T(x)[None])
        prob = F.softmax(score, dim=1)
        conf, _ = prob.max(1)
        est = RM.depth_regression(prob, depth_values=T(dvals)[None])
    out["x"] = x
    out["seed"] = np.array(6000)
    out["depth_values"] = dvals
    out["score"] = score[0].numpy()
    out["view_weight"] = conf[0].numpy()
    out["pair_depth"] = est[0].numpy()
    save("ops_pairnet", **out)


def model_inputs(V, H, W, num_depth, seed):
    return S.model_inputs(V, H, W, num_depth, seed)   # (lives in the package: the large-fixture tests regenerate inputs from the seed)


PEAKED_GAIN = {"casmvsnet": 20.0, "adamvs": 20.0, "msrednet": 5.0}   # (the slice families regress with exp() and no
                                                                        # max-subtraction, SURVEY.md F10: larger gains overflow)


def gen_models_peaked():
    """The "peaked" variants of the V = 5 model fixtures: same inputs protocol, same seeded weights, but the last (logit) layer
    of every regulariser is scaled by PEAKED_GAIN (synthetic.sharpen_state_dict_), so that the distribution over the depth
    planes is peaked and the regressed depth follows the arg-max plane instead of sitting at the middle of a flat range
    (hard parts: random-weight volumes are a weak test of arg-max-sensitive outputs)."""
    gen_models(peaked=True)


def gen_models(only=None, peaked=False):
    for tag, ctor, V, nd, seed in [
        ("model_casmvsnet_v5_peaked", lambda nd: RC.Infer_CascadeMVSNet(num_depth=nd), 5, 384, 7102),
        ("model_adamvs_v5_peaked", lambda nd: RA.Infer_AdaMVSNet(num_depth=nd), 5, 384, 7104),
        ("model_msrednet_v5_peaked", lambda nd: RR.Infer_CascadeREDNet(num_depth=nd), 5, 384, 7106),
    ] if peaked else [
        ("model_casmvsnet_v3", lambda nd: RC.Infer_CascadeMVSNet(num_depth=nd), 3, 64, 7001),
        ("model_casmvsnet_v5", lambda nd: RC.Infer_CascadeMVSNet(num_depth=nd), 5, 384, 7002),
        ("model_adamvs_v3", lambda nd: RA.Infer_AdaMVSNet(num_depth=nd), 3, 64, 7003),
        ("model_adamvs_v5", lambda nd: RA.Infer_AdaMVSNet(num_depth=nd), 5, 384, 7004),
        ("model_msrednet_v3", lambda nd: RR.Infer_CascadeREDNet(num_depth=nd), 3, 64, 7005),
        ("model_msrednet_v5", lambda nd: RR.Infer_CascadeREDNet(num_depth=nd), 5, 384, 7006),
    ]:
        if only and tag not in only:
            continue
        H, W = 64, 96
        net = ctor(nd).eval()
        S.fill_state_dict_(net.state_dict(), seed)
        gain = PEAKED_GAIN[tag.split("_")[1]] if peaked else 1.0
        if peaked:
            assert S.sharpen_state_dict_(net.state_dict(), gain) > 0
        imgs, pm, dv = model_inputs(V, H, W, nd, seed)
        with torch.no_grad():
            o = net(T(imgs), {k: T(v) for k, v in pm.items()}, T(dv))
        if peaked:
            c1 = o["stage1"]["photometric_confidence"].numpy()
            print("  %s: logit gain %g, stage-1 confidence median %.3f (flat: %.3f), depth std %.1f" % (
                tag, gain, float(np.median(c1)), (4.0 if "casmvsnet" in tag else 1.0) / 48.0, float(o["depth"].std())))
            assert np.isfinite(o["depth"].numpy()).all()
        out = {"imgs": imgs, "depth_values": dv, "seed": np.array(seed), "num_depth": np.array(nd), "logit_gain": np.array(gain),
               "n_state_keys": np.array(len(net.state_dict())),
               "state_keys": np.array(list(net.state_dict().keys())),
               "state_shapes": np.array([",".join(map(str, v.shape)) for v in net.state_dict().values()])}
        for k, v in pm.items():
            out["proj_" + k] = v
        out["depth"] = o["depth"][0].numpy()
        out["photometric_confidence"] = o["photometric_confidence"][0].numpy()
        for s in ("stage1", "stage2", "stage3"):
            out[s + "_depth"] = o[s]["depth"][0].numpy()
            out[s + "_conf"] = o[s]["photometric_confidence"][0].numpy()
        if "adamvs" in tag:
            pcs = o["stage1"]["pair_confidence"]
            out["stage1_view_weights"] = np.stack([pcs[i][0, 0].numpy() for i in range(V - 1)])
            out["stage1_pair_depths"] = np.stack([p[0].numpy() for p in o["stage1"]["pair_result"]])
        save(tag, **out)


def gen_models_large():
    """Model fixtures at 256 x 384, V = 5, num_depth = 384 -- a size at which the PRODUCTION kernels are the ones selected
    (stage-3 maps of 98 304 pixels: the 2-D tile kernels, the fused conv-GRU cell, the window sweep, the CL8 volume), flat and
    peaked.  Only the reference's OUTPUTS are stored; the tests regenerate the inputs with synthetic.model_inputs."""
    H, W, V, nd = 256, 384, 5, 384
    for tag, ctor, seed in [("casmvsnet", lambda: RC.Infer_CascadeMVSNet(num_depth=nd), 7202),
                            ("adamvs", lambda: RA.Infer_AdaMVSNet(num_depth=nd), 7204),
                            ("msrednet", lambda: RR.Infer_CascadeREDNet(num_depth=nd), 7206)]:
        for peaked in (False, True):
            net = ctor().eval()
            S.fill_state_dict_(net.state_dict(), seed)
            # (the slice families regress with exp() and no max-subtraction, SURVEY.md F10: at this size the small fixtures' gain of
            #  20 overflows AdaMVS's exp-sum)
            gain = {"casmvsnet": 20.0, "adamvs": 8.0, "msrednet": 4.0}[tag] if peaked else 1.0
            if peaked:
                assert S.sharpen_state_dict_(net.state_dict(), gain) > 0
            imgs, pm, dv = model_inputs(V, H, W, nd, seed)
            with torch.no_grad():
                o = net(T(imgs), {k: T(v) for k, v in pm.items()}, T(dv))
            assert np.isfinite(o["depth"].numpy()).all()
            out = {"seed": np.array(seed), "num_depth": np.array(nd), "V": np.array(V), "H": np.array(H), "W": np.array(W),
                   "logit_gain": np.array(gain), "imgs_checksum": np.array(float(np.abs(imgs.astype(np.float64)).sum()))}
            for st in ("stage1", "stage2", "stage3"):
                out[st + "_depth"] = o[st]["depth"][0].numpy()
                out[st + "_conf"] = o[st]["photometric_confidence"][0].numpy()
            name = "model_%s_v5_256%s" % (tag, "_peaked" if peaked else "")
            print("  %s: stage-1 confidence median %.3f, depth std %.1f" % (name, float(np.median(out["stage1_conf"])), float(o["depth"].std())))
            save(name, **out)


def gen_ucsnet():
    """UCS-Net (ucsnet.py): uncertainty_aware_samples both branches (30-53), the variance tail of compute_depth (137-151),
    and Infer_UCSNet end to end.  The class is built with its own constructor arguments; `num_depth`, which forward reads
    but nothing sets, is assigned as an attribute -- the reference's code is otherwise run as it is."""
    out = {}
    rng = np.random.default_rng(900)
    dv2 = np.array([[400.0, 800.0]], np.float32)
    s1 = RU.uncertainty_aware_samples(T(dv2), None, 48, torch.device("cpu"), torch.float32, [1, 6, 10])
    out["s1_depth_values"], out["s1_samples"] = dv2[0], s1[0].numpy()
    cur = (600 + 50 * rng.standard_normal((1, 1, 12, 20))).astype(np.float32)
    var = np.abs(8 * rng.standard_normal((1, 1, 12, 20))).astype(np.float32)
    s2 = RU.uncertainty_aware_samples(T(cur), T(var), 8, torch.device("cpu"), torch.float32, [1, 12, 20])
    out["s2_cur"], out["s2_var"], out["s2_samples"] = cur[0, 0], var[0, 0], s2[0].numpy()
    # tail of compute_depth on a given probability-volume pre-activation
    D, h, w = 16, 10, 14
    pre = (2.0 * rng.standard_normal((1, D, h, w))).astype(np.float32)
    pre[0, :, 0, 0] = -30.0; pre[0, 3, 0, 0] = 30.0      # a peaked column
    samps = np.sort(rng.uniform(400, 800, (1, D, h, w)).astype(np.float32), 1)
    prob = F.softmax(T(pre), dim=1)
    depth = RM.depth_regression(prob, depth_values=T(samps))
    samp_variance = (T(samps) - depth.unsqueeze(1)) ** 2
    exp_variance = 1.5 * torch.sum(samp_variance * prob, dim=1, keepdim=False) ** 0.5
    out["cd_pre"], out["cd_samps"], out["cd_depth"], out["cd_variance"] = pre[0], samps[0], depth[0].numpy(), exp_variance[0].numpy()
    save("ops_ucsnet", **out)

    for tag, V, nd, seed in [("model_ucsnet_v3", 3, 64, 7011), ("model_ucsnet_v5", 5, 384, 7012)]:
        H, W = 64, 96
        net = RU.Infer_UCSNet().eval()
        net.num_depth = nd
        S.fill_state_dict_(net.state_dict(), seed)
        imgs, pm, dv = model_inputs(V, H, W, nd, seed)
        with torch.no_grad():
            o = net(T(imgs), {k: T(v) for k, v in pm.items()}, T(dv))
        res = {"imgs": imgs, "depth_values": dv, "seed": np.array(seed), "num_depth": np.array(nd),
               "n_state_keys": np.array(len(net.state_dict())),
               "state_keys": np.array(list(net.state_dict().keys())),
               "state_shapes": np.array([",".join(map(str, v.shape)) for v in net.state_dict().values()])}
        for k, v in pm.items():
            res["proj_" + k] = v
        res["depth"] = o["depth"][0].numpy()
        res["photometric_confidence"] = o["photometric_confidence"][0].numpy()
        res["variance"] = o["variance"][0].numpy()
        for st in ("stage1", "stage2", "stage3"):
            res[st + "_depth"] = o[st]["depth"][0].numpy()
            res[st + "_conf"] = o[st]["photometric_confidence"][0].numpy()
            res[st + "_variance"] = o[st]["variance"][0].numpy()
        save(tag, **res)


if __name__ == "__main__":
    which = sys.argv[1:] or ["warp", "aggregate", "regress", "gru", "gru2", "costreg3d", "pairnet", "models", "models_peaked", "ucsnet",
                             "models_large"]
    for wname in which:
        if wname.startswith("model_"):
            gen_models(only=[wname])
        else:
            globals()["gen_" + wname]()
