#!/bin/bash
# Run GPU steps one after another on the gpurun box.  An ordinary failure (test assertion, rc < 124) does not stop
# the chain; a timeout or a kill (rc >= 124) does: no further GPU step is started after one was killed.
#   step <seconds> <logfile> <command...>
mkdir -p gpurun_out
step() {
  local secs=$1 log=$2; shift 2
  echo "=== $(date +%T) $* (limit ${secs}s) -> $log"
  timeout -k 10 "$secs" "$@" > "gpurun_out/$log" 2>&1
  local rc=$?
  echo "=== rc=$rc"; tail -n 3 "gpurun_out/$log"
  if [ $rc -ge 124 ]; then echo "killed/timeout: stopping the chain"; exit $rc; fi
  return 0
}
