# usage: tools/build_cp_variant.sh NAME "-DFLAGS"  -> epsm_mitsuba3_amd/libepsm_NAME.so (A/B builds of the constraint-parallel
# backward kernel: EPSM_LIB_NAME=libepsm_NAME.so selects it).  The flags apply to epsm_backward_cp.hip only; prints the
# resource usage of the packed manifold instantiation (headline kernel).
set -e
cd "$(dirname "$0")/../epsm_mitsuba3_amd/csrc"
make -s
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=fast -fno-slp-vectorize -Wall -Wno-unused-function -Wno-pass-failed"
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-sched-strategy=iterative-maxocc $2 -Rpass-analysis=kernel-resource-usage -c -o build/epsm_backward_cp_$1.o epsm_backward_cp.hip 2> build/epsm_backward_cp_$1.txt || { cat build/epsm_backward_cp_$1.txt | grep -E "error" -A3; exit 1; }
grep -A9 "Function Name: _ZN12_GLOBAL__N_123epsm_backward_cp_kernelILi0ELi2ELb1ELb0ELi2048ELb${DROP:-1}E" build/epsm_backward_cp_$1.txt | grep -E "VGPRs:|Spill|Scratch|Occupancy|LDS" | sed 's/.*remark: [^ ]* //; s/\[-Rpass.*//' | tr '\n' ' '; echo " <- $1"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libepsm_$1.so build/epsm_grad.o build/epsm_tangent.o build/epsm_scatter.o build/epsm_grad_scatter.o build/epsm_backward_cp_$1.o build/epsm_trace*.o build/epsm_matcher.o
