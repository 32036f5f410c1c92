#!/bin/bash
# Calls gpurun until it gets a slot (exit code 3 = none free right now).  Usage: tools/gpu_retry.sh <timeout> <logfile> <command...>
t=$1; log=$2; shift 2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@" > $log 2>&1
  rc=$?
  if [ $rc != 3 ] && ! grep -q "status=transient" $log; then exit $rc; fi
  sleep 45
done
exit 3
