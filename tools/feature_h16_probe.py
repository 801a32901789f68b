"""Round 5: do the feature pyramids have to stay fp32 now that the library's 16-bit format is IEEE half?  The six 256 x 384 model
fixtures (flat / peaked, the reference's outputs) in h16 mode with the pyramids in fp32 (default) and following the mode
(D3D_FEATURE_PRECISION=follow): per-stage depth error in stage-3 intervals and the final depth's relative L1 (budgets of
tests/test_parity_gpu.py: 0.25 intervals, 1e-3).  Run from the repository root on the GPU box."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import load_golden, rel_l1
from deep3d_aerial_amd import config, ops, synthetic as S
from deep3d_aerial_amd.adamvs import Infer_AdaMVSNet
from deep3d_aerial_amd.cas_mvsnet import Infer_CascadeMVSNet
from deep3d_aerial_amd.msrednet import Infer_CascadeREDNet
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
host = lambda t: t.detach().cpu().numpy()
print("library 16-bit format:", ops.h16_dtype())
for name in ("casmvsnet", "adamvs", "msrednet"):
    for peaked in (False, True):
        g = load_golden("model_%s_v5_256%s" % (name, "_peaked" if peaked else ""))
        V, H, W, nd, seed = (int(g[k]) for k in ("V", "H", "W", "num_depth", "seed"))
        imgs, pm, dv = S.model_inputs(V, H, W, nd, seed)
        net = {"casmvsnet": Infer_CascadeMVSNet, "adamvs": Infer_AdaMVSNet, "msrednet": Infer_CascadeREDNet}[name](num_depth=nd)
        S.fill_state_dict_(net.state_dict(), seed)
        if peaked:
            S.sharpen_state_dict_(net.state_dict(), float(g["logit_gain"]))
        net = net.cuda().eval()
        interval = float(dv[0, -1] - dv[0, 0]) / nd
        for feat in ("fp32", "follow"):
            config.switches["D3D_FEATURE_PRECISION"] = feat
            ops.set_conv_precision("h16")
            try:
                with torch.no_grad():
                    out = net(dev(imgs), {k: dev(v) for k, v in pm.items()}, dev(dv))
            finally:
                ops.set_conv_precision(None)
                config.switches["D3D_FEATURE_PRECISION"] = "fp32"
            errs = [float((np.abs(host(out[s]["depth"][0]) - g[s + "_depth"]) / interval).mean()) for s in ("stage1", "stage2", "stage3")]
            cerr = [float(np.abs(host(out[s]["photometric_confidence"][0]) - g[s + "_conf"]).mean()) for s in ("stage1", "stage2", "stage3")]
            print("%-10s %-6s features %-6s: depth error %.3f %.3f %.3f intervals, confidence %.1e %.1e %.1e, final rel-L1 %.2e" % (
                name, "peaked" if peaked else "flat", feat, *errs, *cerr, rel_l1(host(out["depth"][0]), g["stage3_depth"])), flush=True)
