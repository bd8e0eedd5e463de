#!/bin/bash
# longer parity campaigns on the final round-2 build (about 15 minutes on the GPU box)
cd /root/repo
python tools/parity_campaign.py --poses 10000 --seed 302 --cull 1 > gpurun_out/r02c_campaign_long.log 2>&1; tail -2 gpurun_out/r02c_campaign_long.log
python tools/parity_campaign.py --poses 1500 --w 1920 --h 1080 --seed 303 > gpurun_out/r02c_campaign_long_1080p.log 2>&1; tail -2 gpurun_out/r02c_campaign_long_1080p.log
python tools/parity_campaign.py --count --poses 1500 --w 320 --h 180 --seed 304 > gpurun_out/r02c_campaign_long_count.log 2>&1; tail -2 gpurun_out/r02c_campaign_long_count.log
python tools/parity_campaign.py --secondary --poses 2500 --w 480 --h 270 --seed 305 --cull 1 > gpurun_out/r02c_campaign_long_secondary.log 2>&1; tail -2 gpurun_out/r02c_campaign_long_secondary.log
python tools/parity_campaign.py --secondary --count --poses 400 --w 320 --h 180 --seed 306 > gpurun_out/r02c_campaign_long_secondary_count.log 2>&1; tail -2 gpurun_out/r02c_campaign_long_secondary_count.log
python tools/parity_campaign.py --deep --poses 2000 --seed 307 > gpurun_out/r02c_campaign_long_deep.log 2>&1; tail -2 gpurun_out/r02c_campaign_long_deep.log
