#!/bin/bash
# debugging aid: run a pytest node, and if it is still alive after $2 seconds print the native stacks of all its threads
export EXON_HIP_INFLATE_PAR=${EXON_HIP_INFLATE_PAR:-1}
timeout -s KILL 150 python -m pytest "$1" -x -q -m gpu > gpurun_out/hang.log 2>&1 &
sleep ${2:-60}
PID=$(pgrep -n -f "pytest $1" || true)
PID=$(ps -eo pid,comm,args | awk '$2 ~ /python/ && $0 ~ /pytest/ {print $1}' | tail -1)
echo "pid $PID"
[ -n "$PID" ] && timeout 60 /opt/rocm/bin/rocgdb -p $PID -batch -ex "set pagination off" -ex "thread apply all bt 14" 2>&1 | grep -v "^\[New\|^warning\|No symbol table" | cut -c1-200 | tail -120
