#!/bin/bash
# one gpurun call of round 3: runs the commands of the step file given as $1 (one command per line), each under its own
# timeout, logs to gpurun_out/<tag>_<i>.log
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=$1; shift
i=0
while IFS= read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  echo "=== [$i] $line" | tee -a gpurun_out/${TAG}_summary.txt
  ( eval "timeout 2400 env $line" ) > gpurun_out/${TAG}_$i.log 2>&1
  echo "exit $?" | tee -a gpurun_out/${TAG}_summary.txt
  tail -n 25 gpurun_out/${TAG}_$i.log
done < "$1"
