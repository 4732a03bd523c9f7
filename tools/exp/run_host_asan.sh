#!/bin/bash
# run_host_asan.sh: the C++ host program built with AddressSanitizer and UBSan on the HOST side only
# (g++ -fsanitize=address,undefined -static-libasan; liblsk.so itself is the ordinary build), over the three
# solvers, CSR and COO, eager and with recorded traces.  Leak checking is off (the HIP runtime keeps its own).
R=${GRAFT_REPO_ROOT:-/root/repo}
B=$R/tools/exp/BenchmarkStencil_asan
[ -x $B ] || g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -static-libasan -static-libubsan \
  -Wall -Wextra -I $R/include -I $R/host_cpp $R/host_cpp/benchmark_stencil.cpp -L $R/legionsolvers_amd -llsk \
  -Wl,-rpath,'$ORIGIN/../../legionsolvers_amd' -o $B || exit 1
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
for solver in 1 2 3; do
  for coo in 0 1; do
    for graph in 0 1; do
      echo "== solver $solver coo $coo graph $graph"
      timeout -k 10 120 $B -dim 4 -nx 11 -ny 12 -nz 13 -vp 3 -solver $solver -it 8 -pt 2 -restart 4 -history 1 -coo $coo -graph $graph | tail -3 || exit 1
    done
  done
done
timeout -k 10 120 $B -dim 4 -nx 12 -ny 11 -nz 10 -vp 3 -it 4 -systems 1 -adjoint 1 | tail -2 || exit 1
timeout -k 10 120 $B -dim 4 -nx 10 -ny 9 -nz 8 -vp 3 -solver 2 -it 6 -history 1 -coo 1 -binned 1 -bincols 64 -binrows 128 | tail -2 || exit 1
timeout -k 10 120 $B -dim 4 -nx 10 -ny 9 -nz 8 -vp 1 -solver 3 -it 6 -restart 5 -cgs 1 -history 1 | tail -2 || exit 1
# the workspace placement (a copy of 1.5 GB of values gets it: blocks of a third of the device held while the workspace is allocated)
timeout -k 10 200 $B -dim 4 -nx 192 -ny 192 -nz 192 -vp 1 -solver 1 -it 12 -pt 1 -systems 1 2>&1 | grep -v amdgpu | tail -4 || exit 1
timeout -k 10 200 $B -dim 4 -nx 192 -ny 192 -nz 192 -vp 1 -solver 1 -it 12 -pt 1 2>&1 | grep -v amdgpu | tail -4 || exit 1
echo "host sanitizers: clean"
