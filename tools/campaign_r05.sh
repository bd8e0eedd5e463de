#!/bin/bash
# round-5 parity campaigns on the head build (column-major schedule ranks, list shares, look-up-table walk, streaming record stores,
# refill_min 32), new seeds: tools/campaign_r05.sh a|b|c|d [tag]   (logs -> gpurun_out/<tag>_campaign_*.log, copied to profiles/; default r05)
# Every part prints a line per log and keeps writing its log, so that a long part does not look hung.
cd /root/repo
P=tools/parity_campaign.py
T=${2:-r05}
run() { name=$1; shift; python $P "$@" > gpurun_out/${T}_campaign_$name.log 2>&1; echo "$name: $(tail -n 1 gpurun_out/${T}_campaign_$name.log)"; }
case "${1:-a}" in
a)  # short set: every mode once
run 640x360 --poses 2000 --seed 902 --cull 1
run 720p --poses 600 --w 1280 --h 720 --seed 903
run 1080p --poses 150 --w 1920 --h 1080 --seed 907 --cull 1
run count --count --poses 400 --w 320 --h 180 --seed 904
run secondary --secondary --poses 600 --w 480 --h 270 --seed 905 --cull 1
run deep --deep --poses 500 --seed 906
run restart --variant 0 --poses 300 --seed 941
;;
b)  # long: plain frames
run long_640x360 --poses 10000 --seed 1002 --cull 1
run long_1080p --poses 1500 --w 1920 --h 1080 --seed 1003
;;
c)  # long: counting, secondary, deep
run long_count --count --poses 1500 --w 320 --h 180 --seed 1004
run long_secondary --secondary --poses 2500 --w 480 --h 270 --seed 1005 --cull 1
run long_secondary_count --secondary --count --poses 400 --w 320 --h 180 --seed 1006
run long_deep --deep --poses 2000 --seed 1007
;;
d)  # large frames
run count_1080p --count --poses 50 --w 1920 --h 1080 --seed 1104
run deep_count --count --deep --poses 300 --seed 1105
run secondary_count_720p --count --secondary --poses 50 --w 1280 --h 720 --seed 1106
run 4k --poses 40 --w 3840 --h 2160 --seed 1107
;;
esac
