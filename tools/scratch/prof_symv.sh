cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04b_prof_symlib -- python3 tools/symmetric_probe.py f64 65536 > gpurun_out/r04b_prof_symlib.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04b_prof_symprobe -- tools/symv2_probe.out 65536 f64 4 256:8:2:256@0.65,32 > gpurun_out/r04b_prof_symprobe.txt 2>&1
for d in gpurun_out/r04b_prof_symlib gpurun_out/r04b_prof_symprobe; do f=$(find $d -name "*kernel_stats.csv" | head -1); echo "== $f"; head -8 "$f"; done
find gpurun_out/r04b_prof_symlib gpurun_out/r04b_prof_symprobe -name "*kernel_trace.csv" -delete
