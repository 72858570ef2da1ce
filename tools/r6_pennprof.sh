R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r06
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/penn -o run -- python3 $R/tools/config_sweep.py "penn_mvf.yml exactly" > $out/penn.log 2>&1
cp $out/penn/run_kernel_stats.csv $out/penn_mvf_kernel_stats.csv
rm -rf $out/penn
head -30 $out/penn_mvf_kernel_stats.csv | cut -c1-200
