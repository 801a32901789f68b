// LDS-tiled plane-sweep kernels (placeholder until the tiled path lands: always defers to
// the direct kernel).
#include "common.h"
#include "sweep_params.h"

namespace d3d {
int launch_tiled(int, const SweepParams&, hipStream_t) { return D3D_ERR_UNSUPPORTED; }
}  // namespace d3d
