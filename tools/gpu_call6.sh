#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/host_profile.py > gpurun_out/r03j_host_profile.txt 2>&1
head -70 gpurun_out/r03j_host_profile.txt
