#!/bin/bash
# A/B two builds of the library in ONE gpurun call (boxes differ by ~10 % in clocks, so numbers from different calls
# do not compare).  usage: tools/ab.sh /path/A.so /path/B.so [bench args]
A=$1; B=$2; shift 2
ARGS="${*:---steps 1000 --warmup 100 --no-cpu-baseline --no-interactive}"
for round in 1 2; do
  for lib in "$A" "$B"; do
    SF_LIBRARY_PATH=$lib python bench.py $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']/1e6,1), 'M steps/s')"
  done
done
