#!/bin/bash
# Steps in flight (pass-split programs): bench lines of scene-test 320x240 for --streams s
# --in-flight n.  Usage (GPU box): bash tools/in_flight_sweep.sh > gpurun_out/in_flight.log
set -e
for s in 1 2; do for n in 1 2 3 4; do
    echo "== scene-test 320x240 --streams $s --in-flight $n"
    timeout -k 10 200 python bench.py --scene scene-test --size 320x240 --passes-per-step 16 --no-cpu-baseline --streams $s --in-flight $n 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))"
done; done
for n in 2 3; do
    echo "== scene-mesh --streams 1 --in-flight $n"
    timeout -k 10 300 python bench.py --scene scene-mesh --passes-per-step 16 --steps 6 --no-cpu-baseline --streams 1 --in-flight $n 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3))"
done
