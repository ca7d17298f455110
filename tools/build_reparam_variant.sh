# usage: tools/build_reparam_variant.sh NAME "FLAGS"  -> epsm_mitsuba3_amd/libepsm_NAME.so with epsm_trace_reparam.hip rebuilt with extra FLAGS
set -e
cd "$(dirname "$0")/../epsm_mitsuba3_amd/csrc"
make -s
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=fast -fno-slp-vectorize -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $F $2 -c -o build/rp_$1.o epsm_trace_reparam.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libepsm_$1.so build/epsm_grad.o build/epsm_tangent.o build/epsm_scatter.o build/epsm_grad_scatter.o build/epsm_backward_cp.o build/epsm_matcher.o build/epsm_trace.o build/epsm_trace_probe.o build/rp_$1.o
