# the same plan five times in one process (five sets of partial buffers), five processes: pass-1 medians in ms
S="1:256@0.65,32"
for i in 1 2 3 4 5; do tools/symv2_probe_nostore.out 65536 f64 1 $S $S $S $S $S | awk '/pass1/{printf "%s ", $8; e=$NF} END{print " err " e}'; done
