# final kernel and schedule, three processes each: pass-1 medians (ms) with the partial stores and with them compiled out (wrong results: cost of the stores)
S="4501:2048@0.368,512@0.717,64"
for i in 1 2 3; do tools/symv2_probe.out 65536 f64 3 $S $S | awk '/pass1/{printf "%s ", $8} END{printf " | "}'; tools/symv2_probe_nostore.out 65536 f64 3 $S $S | awk '/pass1/{printf "%s ", $8} END{print " (no partial stores)"}'; done
