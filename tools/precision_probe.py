import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import load_golden, rel_l1
from deep3d_aerial_amd import config, ops, synthetic as S
from deep3d_aerial_amd.adamvs import Infer_AdaMVSNet
from deep3d_aerial_amd.cas_mvsnet import Infer_CascadeMVSNet
from deep3d_aerial_amd.msrednet import Infer_CascadeREDNet
from deep3d_aerial_amd.ucsnet import Infer_UCSNet
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for tag in ["model_casmvsnet_v3", "model_casmvsnet_v5", "model_adamvs_v3", "model_adamvs_v5", "model_msrednet_v3", "model_msrednet_v5", "model_ucsnet_v5"]:
    g = load_golden(tag)
    ctor = {"casmvsnet": Infer_CascadeMVSNet, "adamvs": Infer_AdaMVSNet, "msrednet": Infer_CascadeREDNet, "ucsnet": Infer_UCSNet}[tag.split("_")[1]]
    net = ctor(num_depth=int(g["num_depth"]))
    S.fill_state_dict_(net.state_dict(), int(g["seed"]))
    net = net.cuda().eval()
    pm = {s: dev(g["proj_" + s]) for s in ("stage1", "stage2", "stage3")}
    res = []
    for feat in ("fp32", "follow"):
        config.switches["D3D_FEATURE_PRECISION"] = feat
        ops.set_conv_precision("h16")
        with torch.no_grad():
            out = net(dev(g["imgs"]), pm, dev(g["depth_values"]))
        ops.set_conv_precision(None)
        res.append(rel_l1(out["depth"][0].cpu().numpy(), g["depth"]))
    print("%-22s depth rel-L1 vs reference: bf16 regulariser %.2e | + bf16 feature nets %.2e" % (tag, res[0], res[1]), flush=True)
