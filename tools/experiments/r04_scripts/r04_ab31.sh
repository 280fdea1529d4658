#!/bin/bash
# lazy2: the code window's two v_readlane issued one code ahead, in front of the previous code's vector block (J1) against the
# shipped loop (J0); unmodelled-method GPU tests first, then 256 distinct x 4 MiB alternated, then the profile of the winner
mkdir -p gpurun_out/r04
cp build/ab/libJ1.so zpaqsharp_amd/libzpaqhip.so
timeout -k 10 400 python -m pytest tests/test_methods.py -m gpu -x -q > gpurun_out/r04/ab31_tests.log 2>&1; tail -1 gpurun_out/r04/ab31_tests.log
grep -q passed gpurun_out/r04/ab31_tests.log && ! grep -q failed gpurun_out/r04/ab31_tests.log || exit 1
for v in J0 J1 J0 J1; do
  cp build/ab/lib$v.so zpaqsharp_amd/libzpaqhip.so
  timeout -k 10 200 python tools/method_rate.py --methods "x2,1,4,0,3,22" 2>/dev/null | sed "s/^/$v: /"
done | tee gpurun_out/r04/ab31.txt
cp build/ab/libJ1.so zpaqsharp_amd/libzpaqhip.so
