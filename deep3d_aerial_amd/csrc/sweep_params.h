// Kernel-argument block shared by the direct and tiled plane-sweep kernels.
#pragma once
#include "../../include/deep3d_planesweep.h"

namespace d3d {

enum { MODE_WARP = 0, MODE_VARIANCE = 1, MODE_WEIGHTED = 2, MODE_PAIR = 3 };

struct SweepParams {
    const float* feats[D3D_MAX_VIEWS];  // [0] = reference, [1..n_src] = sources, each [C,h,w]
    const float* proj34;                // device [n_src,12]
    const float* depth;                 // [D] or [D,h,w]
    const float* weights;               // [n_src,h,w] (MODE_WEIGHTED)
    float* out;                         // [C,D,h,w] or [D,h,w] (MODE_PAIR)
    int n_src, C, D, h, w;
    int depth_mode;
    int d_chunk;
};

}  // namespace d3d
