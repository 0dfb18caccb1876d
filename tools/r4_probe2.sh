#!/bin/bash
# round-4 probe 2: engine hand-over probe, sleep calibration, schedule fuzzing with and without the copy synchronisation, checker again
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r4p2; mkdir -p $O
timeout 300 python tests/aids/engine_handover_probe.py > $O/engine_probe.txt 2>&1; cat $O/engine_probe.txt | tail -3
python - > $O/sleep_calib.txt 2>&1 <<'PY'
import torch, time
torch.cuda.synchronize()
for c in (100000, 1000000):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda._sleep(c); e1.record(); torch.cuda.synchronize()
    print(c, "cycles =", e0.elapsed_time(e1), "ms")
PY
cat $O/sleep_calib.txt
( SSCG_DBG_NO_COPY_SYNC=1 timeout 900 python tests/aids/fuzz_step.py 6 3 64 2 ) > $O/fuzz_nosync.txt 2>&1; grep -v "^\[W\|Warning" $O/fuzz_nosync.txt | tail -10
( timeout 900 python tests/aids/fuzz_step.py 6 3 64 2 ) > $O/fuzz_fixed.txt 2>&1; grep -v "^\[W\|Warning" $O/fuzz_fixed.txt | tail -10
( GPU_MAX_HW_QUEUES=2 SSCG_SIDE_LANES=3 timeout 900 python tests/aids/fuzz_step.py 4 3 64 2 ) > $O/fuzz_fixed_q2_l3.txt 2>&1; tail -6 $O/fuzz_fixed_q2_l3.txt
( timeout 600 python tests/aids/racecheck_step.py 3 64 2 1 ) > $O/rc_lanes2.txt 2>&1; grep "racecheck:\|^\s*\[" $O/rc_lanes2.txt | cut -c1-250
