#!/bin/bash
# developer helper: ISA of astar_tile.hip -> /tmp/astar_tile.s, the per-phase instruction mix and the kernel's resources
cd /root/repo/ros_navigation_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -mllvm -amdgpu-atomic-optimizer-strategy=None -S --cuda-device-only astar_tile.hip -o /tmp/astar_tile.s -I../../include -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|tsa_search_kernelILi8ELb0E" -A12 | grep -E "error|SGPRs Spill|VGPRs:|Scratch|TotalSGPRs" | head -5
python3 /root/repo/scripts/isa_regions.py /tmp/astar_tile.s
