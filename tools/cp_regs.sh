# usage: tools/cp_regs.sh NAME "-DFLAGS" [kernel-mangled-substring]  -- compile epsm_backward_cp.hip only and print the resource
# usage of the headline instantiation (packed log, manifold, fixed-point rows, windows of 2048) -- no library is linked.
set -e
cd "$(dirname "$0")/../epsm_mitsuba3_amd/csrc"
mkdir -p build/regs/$1
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=fast -fno-slp-vectorize -Wall -Wno-unused-function -Wno-pass-failed"
K=${3:-ILi0ELi2ELb1ELb0ELi2048ELb${DROP:-1}E}
/opt/rocm/bin/hipcc $F -mllvm -amdgpu-sched-strategy=${SCHED:-iterative-maxocc} $2 -Rpass-analysis=kernel-resource-usage -save-temps=obj -c -o build/regs/$1/cp.o epsm_backward_cp.hip 2> build/regs/$1/cp.txt || { grep -E "error" -A3 build/regs/$1/cp.txt; exit 1; }
echo "[$1 $2]" $(grep -A10 "Function Name: _ZN12_GLOBAL__N_123epsm_backward_cp_kernel$K" build/regs/$1/cp.txt | grep -E "SGPRs|VGPRs|Spill|Scratch|Occupancy|LDS" | sed 's/.*remark: [^ ]* //; s/\[-Rpass.*//' | tr '\n' ' ')
