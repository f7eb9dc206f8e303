#!/bin/bash
# After a commit that REMOVES an experiment's code from the library: write the diff that puts it back.
#   tools/park_experiment.sh experiments/r06_NAME.diff      (applies to HEAD = the removal commit: `git apply experiments/r06_NAME.diff`)
set -eu
out="$1"
{
  echo "# Applies to commit $(git rev-parse --short HEAD) (\"$(git log -1 --format=%s)\"): git apply $out"
  echo "# Restores what that commit removed (the state of $(git rev-parse --short HEAD~1))."
  git diff HEAD HEAD~1 -- dartray_amd include tests bench.py tools
} > "$out"
wc -l "$out"
