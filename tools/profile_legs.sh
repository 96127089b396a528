#!/bin/bash
# usage (GPU box, repo root): tools/profile_legs.sh <outdir> [<leg:mode> ...]
# Per benchmark leg (tools/leg_prof.py): rocprofv3 --kernel-trace --stats, then FETCH_SIZE and WRITE_SIZE in separate --pmc passes
# (no trace domains next to --pmc), the program directly behind `--`.  tools/legs_json.py turns the directory into one json.
out=$1; shift
legs=${@:-synth:free synth:idx synth:enc noisy:free noisy:idx poisson3:free poisson3:idx midsize:free midsize:idx midsize_p3:free oddsize:free c4:free c4:idx c4:enc mid2048:free mid2048_p3:free}
mkdir -p $out
export TMPDIR=/tmp
for lm in $legs; do
  leg=${lm%%:*}; mode=${lm##*:}
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t_${leg}_${mode} -- python3 tools/leg_prof.py $leg $mode 10 > $out/${leg}_${mode}.log 2>&1 || { echo "$lm stats failed"; tail -3 $out/${leg}_${mode}.log; exit 1; }
  cp $(ls $out/t_${leg}_${mode}/*/*kernel_stats.csv | head -1) $out/${leg}_${mode}_kernel_stats.csv; rm -rf $out/t_${leg}_${mode}
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 240 rocprofv3 --pmc $c --output-format csv -d $out/p_${leg}_${mode}_$c -- python3 tools/leg_prof.py $leg $mode 3 > $out/${leg}_${mode}_$c.log 2>&1 || { echo "$lm $c failed"; tail -3 $out/${leg}_${mode}_$c.log; exit 1; }
    cp $(ls $out/p_${leg}_${mode}_$c/*/*counter_collection.csv | head -1) $out/${leg}_${mode}_$c.csv; rm -rf $out/p_${leg}_${mode}_$c
  done
  echo "== $lm: $(grep 'ms per call' $out/${leg}_${mode}.log)"
done
