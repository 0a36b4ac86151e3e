#!/bin/bash
# gpurun, waiting for a free GPU slot: exit code 3 ("no box or slot free, nothing charged") is retried every two
# minutes (up to 40 times); any other outcome is returned as it is.  usage: tools/gpurun_wait.sh <timeout s> '<command>'
t=$1; shift
for i in $(seq 1 40); do
    /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
    rc=$?
    if [ $rc -ne 3 ]; then exit $rc; fi
    sleep 120
done
exit 3
