#!/bin/bash
# usage: clock_probe.sh <label> <command...> : polls sclk / power while the command runs
label=$1; shift
"$@" > /tmp/clock_cmd.log 2>&1 &
pid=$!
sleep 1.5
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Power|Graphics Package" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 0.4
done
wait $pid
echo "== $label: $(tail -1 /tmp/clock_cmd.log)"
