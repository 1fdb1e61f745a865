O=gpurun_out/r06_b; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1; tail -5 $O/gputests.log
for spec in "W=0" "W=8" "W=16" "W=16 FB=0"; do
  W=$(echo $spec | sed 's/.*W=\([0-9]*\).*/\1/'); FB=1; case "$spec" in *FB=0*) FB=0;; esac
  export UVO_LAT_OCT_WIDE=$W UVO_LAT_FAST_BLUR=$FB
  for i in 1 2; do python tools/latency.py 2>>$O/lat.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$spec', 'host', d['host_ms_median'], 'topup', d['topup_host_ms_median'], 'device', d['device_ms_median'], d['kernel_us'])" | tee -a $O/latency_ab.txt; done
done
unset UVO_LAT_OCT_WIDE UVO_LAT_FAST_BLUR
python bench.py > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-300; tail -3 $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("step_spread", d["step_spread"]); r=d["roofline"]; print({k:r[k] for k in ("frac","frac_live","frac_alone","avg_launch_ms","avg_launch_ms_alone")})
h=d["host_to_host"]; print("h2h", h["value"], h.get("link_GBps"), h.get("h2h_frac"))
c=d["sub_records"]["configs[3] per-GPU share"]; print("c3", c["value"], c["ms_per_step"], c["whole_path_frac"], c["verified_frames"], c["step_spread"])
print({k:v for k,v in d["sub_records"].items() if k.startswith("configs[1]")})
PY
