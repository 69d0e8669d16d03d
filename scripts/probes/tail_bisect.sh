cd ${GRAFT_REPO_ROOT:-$(pwd)}
echo "== default"; python tests/fuzz/fuzz_q8.py 2 1 2>&1 | grep -v amdgpu.ids | cut -c1-160
echo "== NO_PARK"; MCCONV_NO_PARK=1 python tests/fuzz/fuzz_q8.py 2 1 2>&1 | grep -v amdgpu.ids | cut -c1-160
echo "== BAR_IO=0"; MCCONV_BAR_IO=0 python tests/fuzz/fuzz_q8.py 2 1 2>&1 | grep -v amdgpu.ids | cut -c1-160
echo "== fft0"; MCCONV_LIB=build_ab/lib_fft0.so python tests/fuzz/fuzz_q8.py 2 1 2>&1 | grep -v amdgpu.ids | cut -c1-160
true
