#!/bin/bash
# round 4, JACK path records in one GPU call: the suite + smoke on the final library, the driver-style bench line, the 256-frame tail
# against the round-3 form (build_ab/lib_fft0.so = -DMCCONV_LAB -DMC_TAIL_FFT0, with the early dry-mix load) back to back and spaced,
# the in-kernel stamps (build_ab/lib_trace_td.so), the longer periods, the packed multiply-add probes
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/full_suite.txt 2>&1
rc=$?; tail -2 gpurun_out/full_suite.txt; [ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r4_bench_cfg3.json 2> gpurun_out/r4_bench_cfg3.err; tail -c 200 gpurun_out/r4_bench_cfg3.json; echo
{
  echo "== spaced calls (500 us idle), median / p90 / mean per run of 3000, alternating libraries"
  CALLS=3000 bash scripts/gpu_jack_p50.sh default build_ab/lib_fft0.so
  echo "== back to back and spaced, alternating (scripts/gpu_jack_ab.sh)"
  bash scripts/gpu_jack_ab.sh default MCCONV_LIB=build_ab/lib_fft0.so MCCONV_NO_PARK=1
  echo "== 512- and 1024-frame periods"
  PERIODS="512 1024" bash scripts/gpu_jack_ab.sh default
  echo "== shipped operating point (Q8 regime), 256-frame periods"
  JACK_SHIPPED=1 bash scripts/gpu_jack_ab.sh default MCCONV_LIB=build_ab/lib_fft0.so
  echo "== in-kernel stamps (MC_JACK_TRACE build)"
  LIBS="trace_td trace_td" bash scripts/gpu_tail_trace.sh | grep -o "\[trace.*"
  echo "== packed multiply-add probes"
  ./build_ab/pkfma_probe; ./build_ab/pkfma_tile_probe
} > gpurun_out/r4_jack_tail.txt 2>&1
tail -5 gpurun_out/r4_jack_tail.txt
