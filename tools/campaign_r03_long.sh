#!/bin/bash
# longer parity campaigns on the final round-3 build, in two halves of under 15 minutes each: tools/campaign_r03_long.sh a|b
cd /root/repo
P=tools/parity_campaign.py
if [ "${1:-a}" = "a" ]; then
python $P --poses 10000 --seed 402 --cull 1 > gpurun_out/r03_campaign_long.log 2>&1; tail -n 2 gpurun_out/r03_campaign_long.log
python $P --poses 1500 --w 1920 --h 1080 --seed 403 > gpurun_out/r03_campaign_long_1080p.log 2>&1; tail -n 2 gpurun_out/r03_campaign_long_1080p.log
else
python $P --count --poses 1500 --w 320 --h 180 --seed 404 > gpurun_out/r03_campaign_long_count.log 2>&1; tail -n 2 gpurun_out/r03_campaign_long_count.log
python $P --secondary --poses 2500 --w 480 --h 270 --seed 405 --cull 1 > gpurun_out/r03_campaign_long_secondary.log 2>&1; tail -n 2 gpurun_out/r03_campaign_long_secondary.log
python $P --secondary --count --poses 400 --w 320 --h 180 --seed 406 > gpurun_out/r03_campaign_long_secondary_count.log 2>&1; tail -n 2 gpurun_out/r03_campaign_long_secondary_count.log
python $P --deep --poses 2000 --seed 407 > gpurun_out/r03_campaign_long_deep.log 2>&1; tail -n 2 gpurun_out/r03_campaign_long_deep.log
python $P --variant 2 --poses 2000 --seed 412 --cull 1 > gpurun_out/r03_campaign_long_variant2.log 2>&1; tail -n 2 gpurun_out/r03_campaign_long_variant2.log
python $P --variant 3 --poses 2000 --seed 413 --cull 1 > gpurun_out/r03_campaign_long_variant3.log 2>&1; tail -n 2 gpurun_out/r03_campaign_long_variant3.log
fi
