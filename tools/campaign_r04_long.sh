#!/bin/bash
# longer parity campaigns on the final round-4 build (the f32 step, 23-bit path codes, 7 waves per SIMD), in three parts of under 15 minutes
# each: tools/campaign_r04_long.sh a|b|c      (logs -> gpurun_out/r04_campaign_long_*.log, copied to profiles/)
cd /root/repo
P=tools/parity_campaign.py
case "${1:-a}" in
a)
python $P --poses 10000 --seed 502 --cull 1 > gpurun_out/r04_campaign_long.log 2>&1; tail -n 2 gpurun_out/r04_campaign_long.log
python $P --poses 1500 --w 1920 --h 1080 --seed 503 > gpurun_out/r04_campaign_long_1080p.log 2>&1; tail -n 2 gpurun_out/r04_campaign_long_1080p.log
;;
b)
python $P --count --poses 1500 --w 320 --h 180 --seed 504 > gpurun_out/r04_campaign_long_count.log 2>&1; tail -n 2 gpurun_out/r04_campaign_long_count.log
python $P --secondary --poses 2500 --w 480 --h 270 --seed 505 --cull 1 > gpurun_out/r04_campaign_long_secondary.log 2>&1; tail -n 2 gpurun_out/r04_campaign_long_secondary.log
python $P --secondary --count --poses 400 --w 320 --h 180 --seed 506 > gpurun_out/r04_campaign_long_secondary_count.log 2>&1; tail -n 2 gpurun_out/r04_campaign_long_secondary_count.log
python $P --deep --poses 2000 --seed 507 > gpurun_out/r04_campaign_long_deep.log 2>&1; tail -n 2 gpurun_out/r04_campaign_long_deep.log
;;
c)
python $P --count --poses 50 --w 1920 --h 1080 --seed 604 > gpurun_out/r04_campaign_count_1080p.log 2>&1; tail -n 2 gpurun_out/r04_campaign_count_1080p.log
python $P --count --deep --poses 300 --seed 605 > gpurun_out/r04_campaign_deep_count.log 2>&1; tail -n 2 gpurun_out/r04_campaign_deep_count.log
python $P --count --secondary --poses 50 --w 1280 --h 720 --seed 606 > gpurun_out/r04_campaign_secondary_count_720p.log 2>&1; tail -n 2 gpurun_out/r04_campaign_secondary_count_720p.log
python $P --poses 40 --w 3840 --h 2160 --seed 607 > gpurun_out/r04_campaign_4k.log 2>&1; tail -n 2 gpurun_out/r04_campaign_4k.log
;;
esac
