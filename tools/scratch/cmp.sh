P=tools/symv2_probe.out
for i in 1 2 3; do
SYMV2_GEN=1 $P 65536 f64 4 1:64@1.0 1:256@0.65,32 | tail -1
$P 65536 f64 4 1:64@1.0 1:256@0.65,32 | tail -1
done
