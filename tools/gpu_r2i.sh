#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_register.py tests/test_gpu_stream_driver.py tests/test_gpu_headline_parity.py -m gpu -q -x --timeout 900 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
run() { local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 200 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  echo "$name rc=$?"; python - $O/bench_$name.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("  scans/s %.0f  ms %.4f knn us %.1f  frac %.3f"%(d["value"],d["ms_per_step"],d["roofline"]["avg_launch_ms"]*1e3,d["roofline"]["frac"]))
except Exception as e: print("  parse failed",e)
PY
}
run default
run nomerge LII_NO_MERGE=1
run default2
prof() { local name=$1; shift
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o t -- python bench.py --steps 100 --warmup 10 --prime 20 --no-cpu-baseline --profile-every 0 > $O/prof_$name.log 2>&1
  echo "prof $name rc=$?"
  python tools/summarize_profile.py $O/prof_$name $O/summary_$name.md "$name" > /dev/null 2>&1; python tools/timeline.py $O/prof_$name $O/timeline_$name.md "$name" > /dev/null 2>&1
  rm -rf $O/prof_$name
  sed -n 5,40p $O/timeline_$name.md
}
prof default
