#!/bin/bash
# Run every built experiment binary (tools/exp/exp_*), each under its own time limit.
cd "$(dirname "$0")"
for b in $(ls exp_* | grep -v '\.'); do
  timeout -k 5 60 ./$b "$@" || { echo "$b failed ($?)"; exit 1; }
done
