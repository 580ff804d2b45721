#!/usr/bin/env bash
# Runs a list of GPU steps one after another on the gpurun box, each under its own timeout.
# An ordinary failure (non-zero exit) is logged and the session goes on; a step that times out or is
# killed (124 / 137) ends the session -- no further GPU step is started after a hang.
# usage: tools/gpu_session.sh "name|timeout_s|command" ...
mkdir -p gpurun_out
status=0
for spec in "$@"; do
  name="${spec%%|*}"; rest="${spec#*|}"; tmo="${rest%%|*}"; cmd="${rest#*|}"
  echo "=== [$name] (timeout ${tmo}s): $cmd"
  start=$(date +%s)
  timeout -k 10 "$tmo" bash -o pipefail -c "$cmd" > "gpurun_out/${name}.log" 2>&1
  rc=$?
  echo "=== [$name] exit $rc after $(( $(date +%s) - start ))s"; tail -n 15 "gpurun_out/${name}.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "=== [$name] timed out / killed: stopping the session"; exit $rc; fi
  [ $rc -ne 0 ] && status=$rc
done
exit $status
