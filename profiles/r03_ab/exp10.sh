#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/exp10_args.txt bash profiles/ab_args.sh "--no-profile" "--batch 128 --no-profile" "--batch 112 --no-profile" "--batch 80 --no-profile" "--contexts 2 --batch 128 --no-profile" "--contexts 4 --batch 96 --no-profile" "--no-profile"
