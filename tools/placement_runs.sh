# several plans, interleaved, in several processes (the placement of the buffers differs from process to process): "pass 1 / both passes" medians in ms
for i in 1 2 3 4; do tools/symv2_probe.out 65536 f64 3 1:2048@0.368,512@0.717,64 4501:2048@0.368,512@0.717,64 3201:2048@0.368,512@0.717,64 7201:2048@0.368,512@0.717,64 | awk '/pass1/{printf "%s=%s/%s ", $1, $8, $14} END{print ""}'; done
