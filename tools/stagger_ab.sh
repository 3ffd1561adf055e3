#!/bin/bash
# A/B of FSDP_STAGGER (fsdp_lib.hip launch_pass: the k-th kernel boundary of a pass releases the next pass's first kernel):
# the 20-step bench line and the 100-step line per mode, three runs each.
for k in 0 1 2 3 4; do
  for rep in 1 2 3; do
    v20=$(FSDP_STAGGER=$k timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-latency --stream-batches 0 --no-skidpad 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['value']/1e6,3))")
    v100=$(FSDP_STAGGER=$k timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-latency --stream-batches 0 --no-skidpad 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['value']/1e6,3))")
    echo "FSDP_STAGGER=$k run $rep: 20 steps $v20 M frames/s, 100 steps $v100 M"
  done
done
