# builds a variant of the tiled kernel with given NSUB/NLOADW/PFD and benches it
v() { nsub=$1; nl=$2; pfd=$3; cg=$4; pipe=${5:-1}
  sed -e "s/constexpr int NSUB = [0-9]*;/constexpr int NSUB = $nsub;/" -e "s/constexpr int NLOADW = [0-9]*;/constexpr int NLOADW = $nl;/" -e "s/constexpr int PFD = [0-9]*;/constexpr int PFD = $pfd;/" -e "s/constexpr int LDS_PIPE = [0-9]*;/constexpr int LDS_PIPE = $pipe;/" deep3d_aerial_amd/csrc/planesweep_tiled.hip > deep3d_aerial_amd/csrc/planesweep_tiled_v.hip
  (cd deep3d_aerial_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -c -o /tmp/v.o planesweep_tiled_v.hip 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o libdeep3d_planesweep.so planesweep.o /tmp/v.o regress.o conv.o conv_mfma.o conv_stream.o)
  rm -f deep3d_aerial_amd/csrc/planesweep_tiled_v.hip
  D3D_TILED_STATS=1 D3D_TILED_CG=$cg python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "d3d tiled" | head -2
  D3D_TILED_CG=$cg python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NSUB=$nsub NLOADW=$nl PFD=$pfd CG=$cg', d['value'], 'Mvox/s', d['ms_per_step'], 'ms', d['roofline']['frac'])"
}
cp deep3d_aerial_amd/csrc/libdeep3d_planesweep.so /tmp/keep.so
for spec in "$@"; do v $spec; done
cp /tmp/keep.so deep3d_aerial_amd/csrc/libdeep3d_planesweep.so
