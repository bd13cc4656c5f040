#!/bin/bash
# gpurun with a wait for a free slot (exit code 3 = no box/slot, nothing ran, nothing charged).  usage: grun.sh TIMEOUT 'command'
T=$1; shift
for i in 1 2 3 4 5 6 7 8 9 10; do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
