#!/bin/bash
# more parity campaigns on the final round-4 kernel (hand-written walk loop), other seeds and frame sizes than tools/campaign_r04_long.sh:
# tools/campaign_r04_extra.sh a|b [seed offset] [log tag]     (logs -> gpurun_out/<tag>_campaign_*.log, copied to profiles/; default tag r04x)
cd /root/repo
P=tools/parity_campaign.py
O=${2:-0}
T=${3:-r04x}
case "${1:-a}" in
a)
python $P --poses 6000 --seed $((812 + O)) > gpurun_out/${T}_campaign.log 2>&1; tail -n 2 gpurun_out/${T}_campaign.log
python $P --poses 1200 --w 1280 --h 720 --seed $((813 + O)) --cull 1 > gpurun_out/${T}_campaign_720p.log 2>&1; tail -n 2 gpurun_out/${T}_campaign_720p.log
python $P --poses 600 --w 1920 --h 1080 --seed $((814 + O)) --cull 1 > gpurun_out/${T}_campaign_1080p.log 2>&1; tail -n 2 gpurun_out/${T}_campaign_1080p.log
python $P --variant 0 --poses 600 --seed $((815 + O)) > gpurun_out/${T}_campaign_restart.log 2>&1; tail -n 2 gpurun_out/${T}_campaign_restart.log
;;
b)
python $P --count --poses 1000 --w 320 --h 180 --seed $((816 + O)) > gpurun_out/${T}_campaign_count.log 2>&1; tail -n 2 gpurun_out/${T}_campaign_count.log
python $P --count --poses 150 --w 1280 --h 720 --seed $((817 + O)) > gpurun_out/${T}_campaign_count_720p.log 2>&1; tail -n 2 gpurun_out/${T}_campaign_count_720p.log
python $P --secondary --poses 1500 --w 480 --h 270 --seed $((818 + O)) > gpurun_out/${T}_campaign_secondary.log 2>&1; tail -n 2 gpurun_out/${T}_campaign_secondary.log
python $P --secondary --count --poses 300 --w 320 --h 180 --seed $((819 + O)) > gpurun_out/${T}_campaign_secondary_count.log 2>&1; tail -n 2 gpurun_out/${T}_campaign_secondary_count.log
python $P --deep --poses 1500 --seed $((820 + O)) > gpurun_out/${T}_campaign_deep.log 2>&1; tail -n 2 gpurun_out/${T}_campaign_deep.log
python $P --count --deep --poses 200 --seed $((821 + O)) > gpurun_out/${T}_campaign_deep_count.log 2>&1; tail -n 2 gpurun_out/${T}_campaign_deep_count.log
;;
esac
