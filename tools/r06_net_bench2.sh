#!/bin/bash
# second pass of tools/r06_net_bench.sh: is the cost of the write-through store a property of where the two copies lie?
set -u
cd "$(dirname "$0")/.."
NB="timeout 120 tools/net_bench"
{
echo "== (10) distance between a record's two copies: a multiple of 4 KB (as in passes 1-9), + 512 B, + 1 KB, + 2 KB, + 128 B (uncoupled)"
$NB coupled=0 -- coupled=0 spad=512 -- coupled=0 spad=1024 -- coupled=0 spad=2048 -- coupled=0 spad=128 -- coupled=0 spad=256
$NB coupled=0 far_store=0 -- coupled=0 far_store=0 spad=512
echo "== (11) the same, coupled"
$NB spad=512 -- spad=512 far_store=0 -- spad=512 poll=1
echo "== (12) residency and unaligned record runs: 36x28 patches (126 per XCD, 3.9 per CU), nine 16-byte records per patch in 144-byte runs"
$NB px=36 py=28 coupled=0 spad=512 -- px=36 py=28 spad=512 -- coupled=0 spad=512 pubs=9 -- px=36 py=28 coupled=0 spad=512 far_store=0
} > gpurun_out/r06_net_bench2.txt 2>&1
cat gpurun_out/r06_net_bench2.txt
