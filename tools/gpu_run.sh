#!/bin/bash
# tools/gpu_run.sh -- every GPU-side experiment of this repository behind ONE entry point (round 5; it
# replaces the 59 one-off gpu_r0*.sh scripts of rounds 2-4, which live on in the git history).
#
#   gpurun --timeout 900 -- tools/gpu_run.sh <experiment> [args]
#   gpurun --timeout 1500 -- tools/gpu_run.sh multi '<experiment> [args]' '<experiment> [args]' ...
#
# Outputs go to gpurun_out/<tag>.*; what is to be judged is copied to profiles/ by hand.
#
#   tests [pytest args]            pytest -m gpu (default: the whole GPU suite, -x -q)
#   smoke                          __graft_entry__.smoke()
#   bench <tag> [bench args]       one bench.py line -> gpurun_out/<tag>.json (stderr -> <tag>.err)
#   ab <tag> <envA> <envB> [bench args]
#                                  interleaved A/B of two ENVIRONMENTS ("X=1 Y=2", "-" = none), three rounds,
#                                  the device leg's frames/s of each run -> gpurun_out/<tag>.txt
#   libab <tag> [bench args]       the in-tree library against every build/variants/*.so (UIS_LIB_PATH), three rounds
#   timing <tag> [bench args]      stage clocks from the diagnostic build build/variants/timing.so
#                                  (python -c "from uisrnn_amd import build; build.build(defines=['UIS_RESIDENT_TIMING'], output='build/variants/timing.so')")
#   usweep <tag> <Us> [envs...]    frames/s of U utterances x 500 frames (device leg) for every U of the comma list under
#                                  every environment ("name:X=1 Y=2"; default "default:")  -> gpurun_out/<tag>.json
#   prof <tag> [bench args]        rocprofv3 --kernel-trace --stats of a bench command -> gpurun_out/<tag>_kernel_stats.csv
#   pmc <tag> [bench args]         the PMC passes (MFMA ops / busy, waits, FETCH/WRITE), one counter group per pass
#   fuzz <seconds> [seed]          tools/fuzz_gpu.py
#   stress                         tools/stress_resident.py 100; tools/stress_persistent.py
#   stream <tag>                   tools/stream_latency.py -> gpurun_out/<tag>.json
#   resources                      (CPU) tools/resource_table.py: registers / scratch / spills of every kernel
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
BENCH_QUIET="--no_cpu_baseline --no_host_buffers --no_extra_configs"

value_of() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline'].get('frac'))" "$1" 2>/dev/null; }

run_one() {
  local exp=$1; shift
  case "$exp" in
    tests)
      if [ $# -eq 0 ]; then set -- -x -q; fi
      timeout ${TEST_TIMEOUT:-1500} python -m pytest tests -m gpu "$@" > gpurun_out/pytest.log 2>&1
      echo "pytest rc=$?" >> gpurun_out/pytest.log
      tail -15 gpurun_out/pytest.log ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ;;
    bench)
      local tag=$1; shift
      timeout ${BENCH_TIMEOUT:-600} python bench.py "$@" > gpurun_out/$tag.json 2> gpurun_out/$tag.err
      echo "$tag rc=$? $(value_of gpurun_out/$tag.json)"
      grep -i "error\|Traceback" gpurun_out/$tag.err | head -5 ;;
    ab)
      local tag=$1 enva=$2 envb=$3; shift 3
      : > gpurun_out/$tag.txt
      for rep in 1 2 3; do
        for which in A B; do
          local e=$enva; [ $which = B ] && e=$envb
          [ "$e" = "-" ] && e=""
          local v=$(env $e timeout 300 python bench.py --timed device --steps ${STEPS:-5} --warmup 1 $BENCH_QUIET "$@" 2>/dev/null | tail -1 > gpurun_out/.ab.json; value_of gpurun_out/.ab.json)
          echo "rep=$rep $which [$e] $v" >> gpurun_out/$tag.txt
        done
      done
      sort -k2,2 -s gpurun_out/$tag.txt ;;
    libab)
      local tag=$1; shift
      : > gpurun_out/$tag.txt
      for rep in 1 2 3; do
        for lib in uisrnn_amd/libuisrnn_hip.so build/variants/*.so; do
          local v=$(UIS_LIB_PATH=$PWD/$lib timeout 300 python bench.py --timed device --steps ${STEPS:-5} --warmup 1 $BENCH_QUIET "$@" 2>/dev/null | tail -1 > gpurun_out/.ab.json; value_of gpurun_out/.ab.json)
          echo "rep=$rep $lib $v" >> gpurun_out/$tag.txt
        done
      done
      sort -k2,2 -s gpurun_out/$tag.txt ;;
    timing)
      local tag=$1; shift
      UIS_LIB_PATH=$PWD/build/variants/timing.so timeout 300 python bench.py --timed device --steps 3 --warmup 1 $BENCH_QUIET "$@" \
        > gpurun_out/$tag.json 2> gpurun_out/$tag.err
      grep "timing\]" gpurun_out/$tag.err | tail -${TIMING_LINES:-8} > gpurun_out/$tag.txt
      cat gpurun_out/$tag.txt; value_of gpurun_out/$tag.json ;;
    usweep)
      local tag=$1 us=$2; shift 2
      if [ $# -eq 0 ]; then set -- "default:"; fi
      python - "$tag" "$us" "$@" <<'PY'
import json, os, subprocess, sys
tag, us, envs = sys.argv[1], [int(u) for u in sys.argv[2].split(',')], sys.argv[3:]
frames = os.environ.get('SWEEP_FRAMES', '500')
out = []
for u in us:
  for spec in envs:
    name, _, assigns = spec.partition(':')
    e = dict(os.environ)
    for a in assigns.split():
      k, _, v = a.partition('=')
      e[k] = v
    r = subprocess.run([sys.executable, 'bench.py', '--utterances', str(u), '--frames', frames, '--timed', 'device', '--steps', '3', '--warmup', '1',
                        '--no_cpu_baseline', '--no_host_buffers', '--no_extra_configs'] + os.environ.get('SWEEP_ARGS', '').split(),
                       capture_output=True, text=True, env=e)
    try:
      d = json.loads(r.stdout.strip().splitlines()[-1])
      rec = {'utterances': u, 'env': name, 'frames_per_s': d['value'], 'ms_per_pass': d['ms_per_step'], 'kernel': d['roofline']['kernel'],
             'us_per_decode_step': round(d['roofline']['avg_launch_us'] / (int(frames) * 2), 2), 'frac': d['roofline']['frac'],
             'effective_frac': d['roofline'].get('effective', {}).get('frac')}
    except Exception as ex:  # pylint: disable=broad-except
      rec = {'utterances': u, 'env': name, 'error': str(ex), 'stderr': r.stderr[-400:]}
    out.append(rec)
    print(rec, flush=True)
json.dump({'workload': 'U utterances x %s frames x 256-dim, beam 10, test_iteration 2, device leg' % frames, 'sweep': out},
          open('gpurun_out/%s.json' % tag, 'w'), indent=1)
PY
      ;;
    prof)
      local tag=$1; shift
      rm -rf gpurun_out/prof_$tag
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_$tag -o $tag --output-format csv -- \
         python $REPO/bench.py --steps 5 --warmup 1 --no_cpu_baseline --no_extra_configs "$@" > $REPO/gpurun_out/${tag}_bench_line_under_rocprof.json 2> $REPO/gpurun_out/${tag}_prof.err)
      find gpurun_out/prof_$tag -name "*kernel_stats.csv" -exec cp {} gpurun_out/${tag}_kernel_stats.csv \;
      head -6 gpurun_out/${tag}_kernel_stats.csv ;;
    pmc)
      local tag=$1; shift
      : > gpurun_out/${tag}_pmc.txt
      for grp in "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE" \
                 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
        rm -rf gpurun_out/pmc_tmp
        (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $REPO/gpurun_out/pmc_tmp -o p --output-format csv -- \
           python $REPO/bench.py --timed device --steps 3 --warmup 1 $BENCH_QUIET "$@" > /dev/null 2>> $REPO/gpurun_out/${tag}_pmc.err)
        python tools/pmc_summary.py gpurun_out/pmc_tmp >> gpurun_out/${tag}_pmc.txt 2>&1
      done
      cat gpurun_out/${tag}_pmc.txt | head -40 ;;
    fuzz)
      timeout $(( ${1:-120} + 120 )) python tools/fuzz_gpu.py ${1:-120} ${2:-2027} > gpurun_out/fuzz.txt 2>&1
      tail -3 gpurun_out/fuzz.txt ;;
    stress)
      timeout 400 python tools/stress_resident.py 100 > gpurun_out/stress.txt 2>&1
      timeout 300 python tools/stress_persistent.py >> gpurun_out/stress.txt 2>&1
      tail -4 gpurun_out/stress.txt ;;
    stream)
      local tag=$1; shift
      timeout 400 python tools/stream_latency.py "$@" > gpurun_out/$tag.json 2> gpurun_out/$tag.err
      tail -20 gpurun_out/$tag.json ;;
    resources)
      python tools/resource_table.py "$@" ;;
    *)
      echo "unknown experiment: $exp" >&2; return 2 ;;
  esac
}

if [ "$1" = "multi" ]; then
  shift
  for line in "$@"; do
    echo "=== $line"
    eval "run_one $line"
  done
else
  run_one "$@"
fi
