#!/bin/bash
# round-2 parity campaigns on the final build (logs -> gpurun_out/r02c_campaign_*.log, copied to profiles/)
cd /root/repo
python tools/parity_campaign.py --poses 2000 --seed 202 --cull 1 > gpurun_out/r02c_campaign.log 2>&1; tail -2 gpurun_out/r02c_campaign.log
python tools/parity_campaign.py --poses 600 --w 1280 --h 720 --seed 203 > gpurun_out/r02c_campaign_720p.log 2>&1; tail -2 gpurun_out/r02c_campaign_720p.log
python tools/parity_campaign.py --count --poses 400 --w 320 --h 180 --seed 204 > gpurun_out/r02c_campaign_count.log 2>&1; tail -2 gpurun_out/r02c_campaign_count.log
python tools/parity_campaign.py --secondary --poses 600 --w 480 --h 270 --seed 205 --cull 1 > gpurun_out/r02c_campaign_secondary.log 2>&1; tail -2 gpurun_out/r02c_campaign_secondary.log
python tools/parity_campaign.py --deep --poses 500 --seed 206 > gpurun_out/r02c_campaign_deep.log 2>&1; tail -2 gpurun_out/r02c_campaign_deep.log
