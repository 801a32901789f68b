"""Full-size CasMVSNet in bf16 mode: compare every channel-last variance volume with the planar kernel's (debug)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import predict, ops, synthetic as S
net = predict.build_model("casmvsnet", 384)
S.fill_state_dict_(net.state_dict(), 21)
net = net.cuda().eval()
s = predict.SyntheticBlock(1, 5, 2752, 1856, 384, seed=9)[0]
imgs = torch.from_numpy(s["imgs"])[None].cuda()
pm = {k: torch.from_numpy(v)[None].cuda() for k, v in s["proj_matrices"].items()}
dv = torch.from_numpy(s["depth_values"])[None].cuda()
orig = ops.variance_volume_cl
def checked(f, p, d):
    got = orig(f, p, d)
    planar = ops.variance_volume(f, p, d)
    want = planar.to(ops.h16_dtype()).permute(1, 2, 3, 0).contiguous()
    bad = got.view(torch.int16) != want.view(torch.int16)
    nanp = ~torch.isfinite(planar)
    print("variance volume", tuple(got.shape), "mismatching:", int(bad.sum()), "non-finite CL:", int((~torch.isfinite(got.float())).sum()),
          "non-finite planar:", int(nanp.sum()), "non-finite depth:", int((~torch.isfinite(d)).sum()), flush=True)
    if int(bad.sum()):
        idx = bad.nonzero()
        print("  d", int(idx[:, 0].min()), int(idx[:, 0].max()), "y", int(idx[:, 1].min()), int(idx[:, 1].max()), "x", int(idx[:, 2].min()),
              int(idx[:, 2].max()), "first", idx[:8].tolist())
        i0 = idx[0]
        print("  got", got[i0[0], i0[1], i0[2]].float().tolist(), "\n  want", want[i0[0], i0[1], i0[2]].float().tolist())
    return got
ops.variance_volume_cl = checked
ops.set_conv_precision("h16")
with torch.no_grad():
    o = net(imgs, pm, dv)
for st in ("stage1", "stage2", "stage3"):
    print(st, "non-finite depth:", int((~torch.isfinite(o[st]["depth"])).sum()))
