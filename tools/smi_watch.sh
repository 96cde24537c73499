# sample clocks / power while a command runs: bash tools/smi_watch.sh <out.txt> <cmd...>
OUT=$1; shift
( while true; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction|hotspot)" | tr '\n' ';' ; echo; sleep 1; done ) > $OUT 2>&1 &
W=$!
"$@"
RC=$?
kill $W
exit $RC
