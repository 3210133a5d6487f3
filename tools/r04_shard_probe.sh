# ONE row shard of a P-way split of N=65536 on its own (tools/symv2_probe.out with SYMV2_SHARD=q/P); the library's own schedule is the first spec of each line
for q in 0 5; do SYMV2_SHARD=$q/8 tools/symv2_probe.out 65536 f64 7 1:128@0.92,32 1:64@1.0 1:256@0.92,64 1:512@0.92,64; done
SYMV2_SHARD=1/4 tools/symv2_probe.out 65536 f64 7 1:256@0.92,64 1:128@0.92,32 1:512@0.92,128
SYMV2_SHARD=1/2 tools/symv2_probe.out 65536 f64 7 1:512@0.92,128 1:256@0.92,64 1:1024@0.92,256
