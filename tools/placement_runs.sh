# "pass 1 / both passes" medians in ms: rim tasks of full strips on the lean loop (default) or on the per-element path (SYMV2_NO_FULL=1); one shard, then one 8-way shard
S="4501:2048@0.368,512@0.717,64"
for i in 1 2 3; do tools/symv2_probe.out 65536 f64 3 $S $S | awk '/pass1/{printf "%s/%s ", $8, $14} END{printf " | "}'; SYMV2_NO_FULL=1 tools/symv2_probe.out 65536 f64 3 $S $S | awk '/pass1/{printf "%s/%s ", $8, $14} END{print " (no full)"}'; done
S="4501:128@0.92,32"
for i in 1 2 3; do SYMV2_SHARD=3/8 tools/symv2_probe.out 65536 f64 5 $S $S | awk '/pass1/{printf "%s/%s ", $8, $14} END{printf " | "}'; SYMV2_NO_FULL=1 SYMV2_SHARD=3/8 tools/symv2_probe.out 65536 f64 5 $S $S | awk '/pass1/{printf "%s/%s ", $8, $14} END{print " (no full)"}'; done
