# per-kernel times of the symmetric product at a small N (rocprofv3 --kernel-trace --stats around tools/symmetric_probe.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r04f_prof_small
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04f_prof_small -- python3 tools/symmetric_probe.py f64 10000 > gpurun_out/r04f_prof_small.txt 2>&1
f=$(find gpurun_out/r04f_prof_small -name "*kernel_stats.csv" | head -1); cut -c1-60,200- "$f" | head -8; awk -F'","' 'NR>1{print substr($1,1,50), $2, $4, $6, $7}' "$f" | head -8
find gpurun_out/r04f_prof_small -name "*kernel_trace.csv" -delete
