#!/bin/bash
# more parity campaigns on the round-5 kernel, other seeds and frame sizes than tools/campaign_r05.sh: tools/campaign_r05_extra.sh a|b [seed offset] [tag]
# (logs -> gpurun_out/<tag>_campaign_*.log, copied to profiles/; default tag r05x)
cd /root/repo
P=tools/parity_campaign.py
O=${2:-0}
T=${3:-r05x}
run() { name=$1; shift; python $P "$@" > gpurun_out/${T}_campaign_$name.log 2>&1; echo "$name: $(tail -n 1 gpurun_out/${T}_campaign_$name.log)"; }
case "${1:-a}" in
a)
run 640x360 --poses 6000 --seed $((1212 + O))
run 720p --poses 1200 --w 1280 --h 720 --seed $((1213 + O)) --cull 1
run 1080p --poses 600 --w 1920 --h 1080 --seed $((1214 + O)) --cull 1
run restart --variant 0 --poses 600 --seed $((1215 + O))
;;
b)
run count --count --poses 1000 --w 320 --h 180 --seed $((1216 + O))
run count_720p --count --poses 150 --w 1280 --h 720 --seed $((1217 + O))
run secondary --secondary --poses 1500 --w 480 --h 270 --seed $((1218 + O))
run secondary_count --secondary --count --poses 300 --w 320 --h 180 --seed $((1219 + O))
run deep --deep --poses 1500 --seed $((1220 + O))
run deep_count --count --deep --poses 200 --seed $((1221 + O))
;;
esac
