"""Per-layer timing of the regulariser convolutions at the full-size cascade shapes (GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep3d_aerial_amd import ops  # noqa: E402

H, W = 1856, 2752


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def layer3d(tag, Ci, Co, D, h, w, stride=1, transposed=False):
    x = torch.randn(Ci, D, h, w, device="cuda")
    if transposed:
        wt = torch.randn(Ci, Co, 3, 3, 3, device="cuda") * 0.1
        fn = lambda: ops.convtranspose3d_k3s2(x, wt, relu=True)
        outv = 8 * D * h * w
        flop = 2 * 27 * Ci * Co * D * h * w
    else:
        wt = torch.randn(Co, Ci, 3, 3, 3, device="cuda") * 0.1
        fn = lambda: ops.conv3d_k3(x, wt, relu=True, stride=stride)
        outv = D * h * w // stride ** 3
        flop = 2 * 27 * Ci * Co * outv
    ms = timeit(fn)
    gb = 4 * (Ci * D * h * w + Co * outv) / 1e9
    print("%-28s Ci=%2d Co=%2d %3dx%4dx%4d s%d%s  %8.3f ms  %7.2f TFLOP/s  %7.1f GB/s(min traffic)" % (
        tag, Ci, Co, D, h, w, stride, "T" if transposed else " ", ms, flop / ms / 1e9, gb / ms * 1e3), flush=True)
    return ms


def layer2d(tag, Ci, Co, h, w, stride=1, transposed=False, Ci1=0):
    x = torch.randn(Ci, h, w, device="cuda")
    x2 = torch.randn(Ci1, h, w, device="cuda") if Ci1 else None
    if transposed:
        wt = torch.randn(Ci, Co, 3, 3, device="cuda") * 0.1
        fn = lambda: ops.convtranspose2d_k3s2(x, wt, act=1)
        outv = 4 * h * w
        flop = 2 * 9 * Ci * Co * h * w
    else:
        wt = torch.randn(Co, Ci + Ci1, 3, 3, device="cuda") * 0.1
        fn = lambda: ops.conv2d_k3(x, wt, act=1, stride=stride, x2=x2)
        outv = h * w // stride ** 2
        flop = 2 * 9 * (Ci + Ci1) * Co * outv
    ms = timeit(fn, 20)
    gb = 4 * ((Ci + Ci1) * h * w + Co * outv) / 1e9
    print("%-28s Ci=%2d Co=%2d     %4dx%4d s%d%s  %8.3f ms  %7.2f TFLOP/s  %7.1f GB/s(min traffic)" % (
        tag, Ci + Ci1, Co, h, w, stride, "T" if transposed else " ", ms, flop / ms / 1e9, gb / ms * 1e3), flush=True)
    return ms


def layer3d_cl(tag, Ci, Co, D, h, w, stride=1, transposed=False, in_cl=True, out_cl=True, skip=False):
    """The same layer on channel-last bf16 activations (bf16 mode of CostRegNet); traffic = one read of the input (and of the
    skip) and one write of the output in their formats."""
    x = torch.randn(Ci, D, h, w, device="cuda")
    if in_cl:
        x = ops.to_cl(x)
    if transposed:
        wt = torch.randn(Ci, Co, 3, 3, 3, device="cuda") * 0.1
        outv = 8 * D * h * w
        sk = torch.randn(2 * D, 2 * h, 2 * w, Co, device="cuda").to(ops.h16_dtype()) if skip else None
        fn = lambda: ops.convtranspose3d_k3s2_cl(x, wt, skip=sk, relu=True)
        flop = 2 * 27 * Ci * Co * D * h * w
    else:
        wt = torch.randn(Co, Ci, 3, 3, 3, device="cuda") * 0.1
        outv = D * h * w // stride ** 3
        sk = None
        fn = lambda: ops.conv3d_k3_cl(x, wt, relu=True, stride=stride, out_cl=out_cl)
        flop = 2 * 27 * Ci * Co * outv
    ms = timeit(fn)
    gb = ((2 if in_cl else 4) * Ci * D * h * w + (2 if out_cl else 4) * Co * outv * (2 if skip else 1)) / 1e9
    print("%-28s Ci=%2d Co=%2d %3dx%4dx%4d s%d%s  %8.3f ms  %7.2f TFLOP/s  %7.1f GB/s(min traffic, %s -> %s)" % (
        tag, Ci, Co, D, h, w, stride, "T" if transposed else " ", ms, flop / ms / 1e9, gb / ms * 1e3,
        "CL" if in_cl else "planar", "CL" if out_cl else "planar"), flush=True)
    return ms


def costreg3d_cl(tag, C, D, h, w):
    t = 0
    t += layer3d_cl(tag + " conv0", C, 8, D, h, w, in_cl=False)
    t += layer3d_cl(tag + " conv1", 8, 16, D, h, w, 2)
    t += layer3d_cl(tag + " conv2", 16, 16, D // 2, h // 2, w // 2)
    t += layer3d_cl(tag + " conv3", 16, 32, D // 2, h // 2, w // 2, 2)
    t += layer3d_cl(tag + " conv4", 32, 32, D // 4, h // 4, w // 4)
    t += layer3d_cl(tag + " conv5", 32, 64, D // 4, h // 4, w // 4, 2)
    t += layer3d_cl(tag + " conv6", 64, 64, D // 8, h // 8, w // 8)
    t += layer3d_cl(tag + " conv7T", 64, 32, D // 8, h // 8, w // 8, transposed=True, skip=True)
    t += layer3d_cl(tag + " conv9T", 32, 16, D // 4, h // 4, w // 4, transposed=True, skip=True)
    t += layer3d_cl(tag + " conv11T", 16, 8, D // 2, h // 2, w // 2, transposed=True, skip=True)
    t += layer3d_cl(tag + " prob", 8, 1, D, h, w, out_cl=False)
    print("%s total %.2f ms (channel-last bf16 activations)" % (tag, t), flush=True)


def costreg3d(tag, C, D, h, w):
    t = 0
    t += layer3d(tag + " conv0", C, 8, D, h, w)
    t += layer3d(tag + " conv1", 8, 16, D, h, w, 2)
    t += layer3d(tag + " conv2", 16, 16, D // 2, h // 2, w // 2)
    t += layer3d(tag + " conv3", 16, 32, D // 2, h // 2, w // 2, 2)
    t += layer3d(tag + " conv4", 32, 32, D // 4, h // 4, w // 4)
    t += layer3d(tag + " conv5", 32, 64, D // 4, h // 4, w // 4, 2)
    t += layer3d(tag + " conv6", 64, 64, D // 8, h // 8, w // 8)
    t += layer3d(tag + " conv7T", 64, 32, D // 8, h // 8, w // 8, transposed=True)
    t += layer3d(tag + " conv9T", 32, 16, D // 4, h // 4, w // 4, transposed=True)
    t += layer3d(tag + " conv11T", 16, 8, D // 2, h // 2, w // 2, transposed=True)
    t += layer3d(tag + " prob", 8, 1, D, h, w)
    print("%s total %.2f ms" % (tag, t), flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "3d"):
        costreg3d("cas s3", 8, 8, H, W)
        costreg3d("cas s2", 16, 32, H // 2, W // 2)
        costreg3d("cas s1", 32, 48, H // 4, W // 4)
    if which in ("all", "3dcl"):
        ops.set_conv_precision("h16")
        costreg3d_cl("cas s3", 8, 8, H, W)
        costreg3d_cl("cas s2", 16, 32, H // 2, W // 2)
        costreg3d_cl("cas s1", 32, 48, H // 4, W // 4)
        ops.set_conv_precision(None)
    if which in ("all", "2d"):
        # AdaMVS slice regulariser (per depth plane) at stage 3 resolution and the pair UNet at 1/4
        h, w = H, W
        t = 0
        t += layer2d("gru s3 conv_gru1 gates", 8, 16, h, w, Ci1=8)
        t += layer2d("gru s3 conv_gru1 cand", 8, 8, h, w, Ci1=8)
        t += layer2d("gru s3 conv1 s2", 8, 16, h, w, 2)
        t += layer2d("gru s3 conv_gru2 gates", 16, 32, h // 2, w // 2, Ci1=16)
        t += layer2d("gru s3 conv_gru2 cand", 16, 16, h // 2, w // 2, Ci1=16)
        t += layer2d("gru s3 upconv T", 16, 8, h // 2, w // 2, transposed=True)
        t += layer2d("gru s3 out", 8, 1, h, w)
        print("gru s3 slice total %.2f ms" % t, flush=True)
        h, w = H // 4, W // 4
        layer2d("pair conv0", 48, 8, h, w)
        layer2d("pair conv1 s2", 8, 16, h, w, 2)
        layer2d("pair prob", 8, 1, h, w)
