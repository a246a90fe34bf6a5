#!/bin/bash
# tools/r06_net_trace.sh -- the stamped account of one wait in the lock-step network (tools/net_bench.hip trace=1).  GPU box.
set -u
cd "$(dirname "$0")/.."
NB="timeout 120 tools/net_bench"
{
$NB trace=1 steps=120 stamp=0 -- trace=1 steps=120 stamp=0 coupled=0 -- trace=1 steps=120 stamp=0 coupled=0 far_store=0 -- trace=1 steps=120 stamp=0 coupled=0 nbrs=2 -- trace=1 steps=120 stamp=0 jitter=0 -- trace=1 steps=120 stamp=0 coupled=0 jitter=0
} > gpurun_out/r06_net_trace.txt 2>&1
cat gpurun_out/r06_net_trace.txt
