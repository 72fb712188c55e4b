#!/bin/bash
# build-container helper: gpurun with a retry while no slot / box is free (exit code 3: nothing ran, nothing charged)
T=${GPU_TIMEOUT:-900}
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 60
done
exit 3
