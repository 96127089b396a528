#!/bin/bash
# round 6: the legs the dense walk changes, one line each (tools/leg_prof.py); $1 = output file
out=${1:-gpurun_out/r6_legs.txt}
: > $out
for leg in poisson3 noisy poisson1.5 poisson10 synth oddsize; do
  python3 tools/leg_prof.py $leg free 10 >> $out 2>&1 || echo "$leg FAILED" >> $out
done
cat $out
