#!/bin/bash
# Interleaved A/B of library variants on ONE box:  scripts/ab_libs.sh REPS name1 name2 ...
#   "main" = the in-tree build, NAME = elasticreconstruction_amd/_ab/liber_hip_NAME.so; NAME@Q runs with GPU_MAX_HW_QUEUES=Q
#   AB_ALONE=1: also the extra pass that runs k_integrate alone (one launch at a time)
reps=$1; shift
alone="--no-alone"; [ -n "$AB_ALONE" ] && alone=""
for i in $(seq $reps); do
  for vq in "$@"; do
    v="${vq%@*}"
    if [ "$vq" != "$v" ]; then export GPU_MAX_HW_QUEUES="${vq#*@}"; else unset GPU_MAX_HW_QUEUES; fi
    if [ "$v" = main ]; then unset ER_HIP_LIB; else export ER_HIP_LIB=$PWD/elasticreconstruction_amd/_ab/liber_hip_$v.so; fi
    python bench.py --steps 20 --warmup 2 --cpu-sample 0 --icp-pairs 0 --no-streamed $alone --other-configs 0 --min-seconds 0.3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; a=r.get('kernel_alone')
print('%-10s value %8.0f fps  ms/step %.3f  k_integrate %.3f ms  frac %.3f%s' % ('$vq', d['value'], d['ms_per_step'], r['avg_launch_ms'], r['frac'], ('  alone %.3f ms' % a['avg_launch_ms']) if a else ''))"
  done
done
unset GPU_MAX_HW_QUEUES ER_HIP_LIB
