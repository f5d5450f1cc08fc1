#!/bin/bash
# sample the shader / memory clocks while a command runs:  tools/clocks.sh <logfile> <command...>
LOG=$1; shift
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ' '; echo; sleep 0.05; done ) > $LOG 2>&1 &
SPID=$!
"$@"
RC=$?
kill $SPID 2>/dev/null
exit $RC
