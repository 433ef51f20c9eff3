#!/bin/bash
# round 5: events without system-scope fences (tree) against the same sources with HIP's default events (tools/_ab/lib_sysfence.so),
# one box; then what the library's own timers cost the timed call, and what the factor transfers cost
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
echo "== c3 driver command"; bash tools/ab_libs.sh r05/ab_events 2 "--steps 20 --warmup 5" tree sysfence
echo "== c3 31 steps";       bash tools/ab_libs.sh r05/ab_events 1 "" tree sysfence
echo "== c1";                bash tools/ab_libs.sh r05/ab_events 1 "--workload c1" tree sysfence
echo "== c1 121 steps";      bash tools/ab_libs.sh r05/ab_events 1 "--workload c1 --steps 121" tree sysfence
echo "== c2";                bash tools/ab_libs.sh r05/ab_events 1 "--workload c2" tree sysfence
echo "== c5";                bash tools/ab_libs.sh r05/ab_events 1 "--workload c5" tree sysfence
echo "== timers on / off (tree)"; python tools/profile_cost.py c3
echo "== transfers"; python tools/xfer_probe.py
echo EVENTS_DONE
