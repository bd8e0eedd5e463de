#!/bin/bash
# round-3 parity campaigns on the head build (logs -> gpurun_out/r03_campaign_*.log, copied to profiles/):
# the default kernel as in round 2 (culling pass forced on where it applies), then the two child-mask-table variants.
cd /root/repo
P=tools/parity_campaign.py
python $P --poses 2000 --seed 302 --cull 1 > gpurun_out/r03_campaign.log 2>&1; tail -2 gpurun_out/r03_campaign.log
python $P --poses 600 --w 1280 --h 720 --seed 303 > gpurun_out/r03_campaign_720p.log 2>&1; tail -2 gpurun_out/r03_campaign_720p.log
python $P --poses 150 --w 1920 --h 1080 --seed 307 --cull 1 > gpurun_out/r03_campaign_1080p.log 2>&1; tail -2 gpurun_out/r03_campaign_1080p.log
python $P --count --poses 400 --w 320 --h 180 --seed 304 > gpurun_out/r03_campaign_count.log 2>&1; tail -2 gpurun_out/r03_campaign_count.log
python $P --secondary --poses 600 --w 480 --h 270 --seed 305 --cull 1 > gpurun_out/r03_campaign_secondary.log 2>&1; tail -2 gpurun_out/r03_campaign_secondary.log
python $P --deep --poses 500 --seed 306 > gpurun_out/r03_campaign_deep.log 2>&1; tail -2 gpurun_out/r03_campaign_deep.log
for v in 2 3; do
  python $P --variant $v --poses 1000 --seed 31$v --cull 1 > gpurun_out/r03_campaign_variant$v.log 2>&1; tail -2 gpurun_out/r03_campaign_variant$v.log
  python $P --variant $v --secondary --poses 200 --w 480 --h 270 --seed 32$v > gpurun_out/r03_campaign_variant${v}_secondary.log 2>&1; tail -2 gpurun_out/r03_campaign_variant${v}_secondary.log
  python $P --variant $v --deep --poses 150 --seed 33$v > gpurun_out/r03_campaign_variant${v}_deep.log 2>&1; tail -2 gpurun_out/r03_campaign_variant${v}_deep.log
done
