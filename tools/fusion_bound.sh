#!/bin/bash
# sweep -> conv0 fusion: an upper bound of what it could gain, from two timing-only builds of the library
# (-DD3D_X_SWEEP_NOSTORE: the window kernel computes the channel-last volume and does not write it; -DD3D_X_CONV0_NOLOAD: the
# channel-last layers do not read their input) against the production build.  The fused kernel would save AT MOST
#   (sweep - sweep without its store) + (conv0 - conv0 without its input loads)
# per stage, before it pays for the halo voxels it has to sweep again (conv0 is 3 x 3 x 3).  Run on the GPU box:
#   tools/fusion_bound.sh > gpurun_out/r05_fusion_bound.txt
# The variant library is built in the container:  make -C deep3d_aerial_amd/csrc OBJDIR=$PWD/build_ab/x_nostore \
#   OUT=$PWD/build_ab/x_nostore/libdeep3d_planesweep.so CXXFLAGS="<Makefile's> -DD3D_X_SWEEP_NOSTORE -DD3D_X_CONV0_NOLOAD"
cd ${GRAFT_REPO_ROOT:-.}
X=$PWD/build_ab/x_nostore/libdeep3d_planesweep.so
for lib in "" $X; do
    echo "## library: ${lib:-production}  (d3d_build_flags: '$(D3D_LIBRARY=$lib python3 -c 'from deep3d_aerial_amd import _lib; print(_lib.load().d3d_build_flags().decode())')')"
    echo "# channel-last variance sweeps of the three cascade stages (tools/stage_sweep_bench.py)"
    D3D_LIBRARY=$lib python3 tools/stage_sweep_bench.py window 2>&1 | grep -v amdgpu.ids
    echo "# CostRegNet layers (tools/regnet_layers.py)"
    D3D_LIBRARY=$lib python3 tools/regnet_layers.py 5 2>&1 | grep -v amdgpu.ids
done
