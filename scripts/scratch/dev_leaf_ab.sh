#!/bin/bash
# builds and runs the standalone leaf harness (stamps and production variants)
set -e
cd $GRAFT_REPO_ROOT
F="--offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Igaussian_processes_amd/csrc"
/opt/rocm/bin/hipcc $F scripts/scratch/dev_leaf_time.hip -o /tmp/leaf_time 2>&1 | grep -v warning | head -5 || true
/opt/rocm/bin/hipcc $F -DNO_STAMPS scripts/scratch/dev_leaf_time.hip -o /tmp/leaf_time_ns 2>&1 | grep -v warning | head -5 || true
echo "== production (no stamps)"; timeout -k 5 60 /tmp/leaf_time_ns; timeout -k 5 60 /tmp/leaf_time_ns | head -1
echo "== with stamps"; timeout -k 5 60 /tmp/leaf_time
/opt/rocm/bin/hipcc $F -DMIN_STAMPS scripts/scratch/dev_leaf_time.hip -o /tmp/leaf_time_min 2>&1 | grep -v warning | head -5 || true
echo "== arrivals at B1 only"; timeout -k 5 60 /tmp/leaf_time_min | grep -v "rows-read"
