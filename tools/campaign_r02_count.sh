#!/bin/bash
# parity campaigns of the queued counting path (round 2, second build): hit counters compared after every frame
cd /root/repo
python tools/parity_campaign.py --count --poses 700 --w 320 --h 180 --seed 404 > gpurun_out/r02b_campaign_count.log 2>&1; tail -2 gpurun_out/r02b_campaign_count.log
python tools/parity_campaign.py --secondary --count --poses 300 --w 320 --h 180 --seed 405 > gpurun_out/r02b_campaign_secondary_count.log 2>&1; tail -2 gpurun_out/r02b_campaign_secondary_count.log
python tools/parity_campaign.py --deep --count --poses 60 --w 320 --h 180 --seed 406 > gpurun_out/r02b_campaign_deep_count.log 2>&1; tail -2 gpurun_out/r02b_campaign_deep_count.log
python tools/parity_campaign.py --count --poses 6 --w 1920 --h 1080 --seed 407 > gpurun_out/r02b_campaign_count_1080p.log 2>&1; tail -2 gpurun_out/r02b_campaign_count_1080p.log
