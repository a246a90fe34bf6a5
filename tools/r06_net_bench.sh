#!/bin/bash
# tools/r06_net_bench.sh -- the hand-off of the lock-step network, contributor by contributor (tools/net_bench.hip; GPU box).
# Output: gpurun_out/r06_net_bench.txt
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
[ -x tools/net_bench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/net_bench tools/net_bench.hip 2>/dev/null
NB="timeout 120 tools/net_bench"
{
echo "== (1) the network as the product runs it (coupled, producer-kept records, write-through store first), then uncoupled; with equal arithmetic everywhere"
$NB coupled -- coupled=0 -- coupled jitter=0 -- coupled=0 jitter=0
echo "== (2) no arithmetic"
$NB chain=0 jitter=0 -- coupled=0 chain=0 jitter=0
echo "== (3) how many producers a wait ends on (uncoupled): left+right, 4, 8 neighbours"
$NB coupled=0 nbrs=2 -- coupled=0 nbrs=4 -- coupled=0 nbrs=8 -- coupled=0 nbrs=2 jitter=0 -- coupled=0 nbrs=8 jitter=0
echo "== (4) records per producer (uncoupled, 8 neighbours): 1+1, 3+2 (default), 6+4"
$NB coupled=0 redge=1 rcorner=1 -- coupled=0 redge=6 rcorner=4 pubs=16
echo "== (5) the stores (uncoupled): no write-through copy where no other XCD reads; same-XCD store first"
$NB coupled=0 far_store=0 -- coupled=0 order=1 -- coupled=0 far_store=0 jitter=0 -- coupled=0 far_store=0 chain=0 jitter=0
echo "== (6) consumer-contiguous mailbox for the same-XCD copies (uncoupled, then coupled)"
$NB coupled=0 layout=1 -- coupled=0 layout=1 far_store=0 -- coupled=0 layout=1 far_store=0 jitter=0 -- layout=1 -- layout=1 far_store=0
echo "== (7) the poll (uncoupled): un-narrowed, one s_sleep between rounds, blocking VGPR loads, a wait after every load; the same without the write-through store"
$NB coupled=0 narrow=0 -- coupled=0 gap=1 -- coupled=0 poll=2 -- coupled=0 poll=3
$NB coupled=0 far_store=0 narrow=0 -- coupled=0 far_store=0 gap=1 -- coupled=0 far_store=0 poll=2 -- coupled=0 far_store=0 poll=3
echo "== (8) coupled: write-through only where another XCD reads; near and far lanes by two load instructions; both"
$NB far_store=0 -- poll=1 -- far_store=0 poll=1 -- far_store=0 order=1 -- far_store=0 jitter=0
echo "== (9) patches per XCD: 32x24 (96/XCD), 16x16 (32/XCD: one per CU), 8x8"
$NB px=32 py=24 -- px=16 py=16 -- px=8 py=8 -- px=16 py=16 coupled=0 -- px=8 py=8 coupled=0 -- px=16 py=16 coupled=0 far_store=0
} > gpurun_out/r06_net_bench.txt 2>&1
tail -n 150 gpurun_out/r06_net_bench.txt
