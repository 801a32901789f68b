// GroupNorm(1, C) statistics accumulated where the tensor is produced (module.py:62-67 of the reference: mean and biased variance over
// ALL elements of a channel group; ConvGRUCell2 normalises the reset half and the update half of the gate convolution and the
// candidate convolution -- one or two groups of consecutive channels per layer).  The convolution's epilogue adds every value it
// stores to per-lane fp64 partial sums (sum, sum of squares: the operands and the arithmetic of gn_stats_kernel, regress.hip, which
// read the stored tensor back); gn_flush folds a workgroup's partials and adds them to the layer's fp64 pairs with one atomic per
// value.  The pairs are zeroed by the caller before the launch (stream order).
#pragma once
#include "common.h"

namespace d3d {

struct GnAcc {
    double s[2], q[2];   // per channel group
};

__device__ __forceinline__ void gn_zero(GnAcc& a) { a.s[0] = a.s[1] = a.q[0] = a.q[1] = 0.0; }

template <typename V4>   // four fp32 values (the kernels' own 4-vector types)
__device__ __forceinline__ void gn_add(GnAcc& a, bool second, const V4& y) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double v = (double)y[k];
        // (v * v is exact in fp64 -- 24 x 24 significand bits -- so the fused form rounds once, as the product-then-sum does)
        if (second) { a.s[1] += v; a.q[1] = __builtin_fma(v, v, a.q[1]); }
        else        { a.s[0] += v; a.q[0] = __builtin_fma(v, v, a.q[0]); }
    }
}

// every thread of the workgroup calls it once (after its last gn_add); scratch: 4 doubles per wave of LDS nobody else touches
// between the call's two barriers.  stats: [ngroups][2] = (sum, sum of squares).
__device__ __forceinline__ void gn_flush(const GnAcc& a, double* __restrict__ stats, int ngroups, double* scratch, int tid, int nwaves) {
    double v[4] = {a.s[0], a.q[0], a.s[1], a.q[1]};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[j] += __shfl_down(v[j], o);
    __syncthreads();   // (the scratch may alias a buffer the workgroup was reading)
    if ((tid & 63) == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) scratch[(tid >> 6) * 4 + j] = v[j];
    }
    __syncthreads();
    if (tid < 2 * ngroups) {
        double t = 0.0;
        for (int wv = 0; wv < nwaves; ++wv) t += scratch[wv * 4 + tid];
        atomicAdd(stats + tid, t);
    }
}

}  // namespace d3d
