#!/bin/bash
# round-4 parity campaigns on the head build (logs -> gpurun_out/r04_campaign_*.log, copied to profiles/):
# the default kernel as in round 2 (culling pass forced on where it applies), the reference-shaped kernel on a sample.
cd /root/repo
P=tools/parity_campaign.py
python $P --poses 2000 --seed 402 --cull 1 > gpurun_out/r04_campaign.log 2>&1; tail -2 gpurun_out/r04_campaign.log
python $P --poses 600 --w 1280 --h 720 --seed 403 > gpurun_out/r04_campaign_720p.log 2>&1; tail -2 gpurun_out/r04_campaign_720p.log
python $P --poses 150 --w 1920 --h 1080 --seed 407 --cull 1 > gpurun_out/r04_campaign_1080p.log 2>&1; tail -2 gpurun_out/r04_campaign_1080p.log
python $P --count --poses 400 --w 320 --h 180 --seed 404 > gpurun_out/r04_campaign_count.log 2>&1; tail -2 gpurun_out/r04_campaign_count.log
python $P --secondary --poses 600 --w 480 --h 270 --seed 405 --cull 1 > gpurun_out/r04_campaign_secondary.log 2>&1; tail -2 gpurun_out/r04_campaign_secondary.log
python $P --deep --poses 500 --seed 406 > gpurun_out/r04_campaign_deep.log 2>&1; tail -2 gpurun_out/r04_campaign_deep.log
python $P --variant 0 --poses 300 --seed 441 > gpurun_out/r04_campaign_restart.log 2>&1; tail -2 gpurun_out/r04_campaign_restart.log
