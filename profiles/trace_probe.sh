#!/bin/bash
# kernel trace of one probe script: bash profiles/trace_probe.sh <tag> <script.py> [args...]   -> gpurun_out/trace_<tag>/
TAG=$1; shift
OUT=$PWD/gpurun_out/trace_$TAG
rm -rf $OUT && mkdir -p $OUT
ROOT=$PWD
SCRIPT=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $SCRIPT "$@" > $OUT/stdout.txt 2>&1
