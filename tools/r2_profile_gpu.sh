#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/profile_models.py vrcnet > gpurun_out/r2k_profile_vrcnet.txt 2>&1
grep -v "^\[W\|amdgpu.ids" gpurun_out/r2k_profile_vrcnet.txt | cut -c1-80,150-215 | head -40
