#!/bin/bash
# bash tools/with_heartbeat.sh <command...>: run a long, quiet command while printing a line every 45 s (gpurun kills silent runs)
( while sleep 45; do echo "tick $(date +%T)"; done ) &
HB=$!
"$@"
rc=$?
kill $HB
exit $rc
