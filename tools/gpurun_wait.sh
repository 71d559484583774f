#!/bin/bash
# usage: tools/gpurun_wait.sh TIMEOUT 'command'   -- retries only while no GPU slot is free (exit code 3: nothing ran)
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
