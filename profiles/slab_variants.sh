run() { # label, env, args
  out=$(env $2 python3 bench.py --slab 1/4 --steps 40 --warmup 5 --no-cpu-baseline $3 2>gpurun_out/slabvar.err | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); s=d['slab']; print(s['ms_per_step'], s['integrate_ms'], s['ms_per_step_without_halo'], s.get('frame_gather_ms_to_self'), s.get('schedule'))
except Exception as e: print('ERR', e)")
  echo "$1: $out"; tail -2 gpurun_out/slabvar.err | grep -i -E "error|Traceback|rgbdr" | head -3
}
run "torch redundant" "X=1" ""
run "managed redundant" "X=1" "--managed"
run "managed shard seq" "X=1" "--managed --shard"
run "managed shard pipe" "X=1" "--managed --shard --pipeline"
run "managed shard pipe cu16" "RGBDR_CU_SPLIT=16" "--managed --shard --pipeline"
run "managed shard pipe cu32" "RGBDR_CU_SPLIT=32" "--managed --shard --pipeline"
run "managed redundant pipe cu32" "RGBDR_CU_SPLIT=32" "--managed --pipeline"
run "managed shard seq cu32" "RGBDR_CU_SPLIT=32" "--managed --shard"
