# usage: tools/build_variant.sh NAME "-DFLAGS"  -> epsm_mitsuba3_amd/libepsm_NAME.so (A/B builds: EPSM_LIB_NAME=libepsm_NAME.so)
# The flags apply to the two units that contain the per-path code (epsm_grad_scatter, epsm_grad).
set -e
cd "$(dirname "$0")/../epsm_mitsuba3_amd/csrc"
make -s
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=fast -fno-slp-vectorize -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-sched-strategy=iterative-maxocc $2 -c -o build/epsm_grad_scatter_$1.o epsm_grad_scatter.hip &
/opt/rocm/bin/hipcc $F $2 -c -o build/epsm_grad_$1.o epsm_grad.hip &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libepsm_$1.so build/epsm_grad_$1.o build/epsm_tangent.o build/epsm_scatter.o build/epsm_grad_scatter_$1.o build/epsm_trace.o build/epsm_matcher.o
