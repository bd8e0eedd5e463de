#!/bin/bash
# parity campaigns of the counting (adaptive) mode on the build that merges its queue before the flush (round 3):
# full-size frames, deep trees, fs_main's ray set (primary + shadow) -- counters AND records against the oracle.
cd /root/repo
P=tools/parity_campaign.py
python $P --count --poses 50 --w 1920 --h 1080 --seed 504 > gpurun_out/r03_campaign_count_1080p.log 2>&1; tail -n 2 gpurun_out/r03_campaign_count_1080p.log
python $P --count --deep --poses 300 --seed 505 > gpurun_out/r03_campaign_deep_count.log 2>&1; tail -n 2 gpurun_out/r03_campaign_deep_count.log
python $P --count --secondary --poses 50 --w 1280 --h 720 --seed 506 > gpurun_out/r03_campaign_secondary_count_720p.log 2>&1; tail -n 2 gpurun_out/r03_campaign_secondary_count_720p.log
