// Kernel-argument block shared by the direct and tiled plane-sweep kernels.
#pragma once
#include <cstddef>

#include "../../include/deep3d_planesweep.h"

namespace d3d {

enum { MODE_WARP = 0, MODE_VARIANCE = 1, MODE_WEIGHTED = 2, MODE_PAIR = 3 };

struct SweepParams {
    const float* feats[D3D_MAX_VIEWS];  // [0] = reference, [1..n_src] = sources, each [C,h,w] (fp16 tensors when elem_bytes == 2)
    const float* proj34;                // device [n_src,12]
    const float* depth;                 // [D] or [D,h,w]
    const float* weights;               // [n_src,h,w] (MODE_WEIGHTED)
    float* out;                         // [C,D,h,w] or [D,h,w] (MODE_PAIR); fp16 when elem_bytes == 2
    void* workspace;                    // caller-owned scratch (d3d_sweep_workspace_bytes), or null
    size_t workspace_bytes;
    int n_src, C, D, h, w;
    int depth_mode;
    int d_chunk;
    int elem_bytes;                     // 4 (fp32 tensors) | 2 (fp16 storage, fp32 arithmetic)
    int plane_major;                    // 0: out [C,D,h,w] | 1: out [D,C,h,w] (one contiguous [C,h,w] slice per plane)
    int out_cl;                         // MODE_VARIANCE, ring / window kernels: 1 = channel-last bf16 volume [D,h,w,C];
                                        // 2 = the same in planes of 8-channel groups [D,C/8,h,w,8] ("CL8": every 16-byte store a whole cell)
};

}  // namespace d3d
