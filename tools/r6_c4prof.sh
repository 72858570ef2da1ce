R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r06
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for f in 0 1; do
  export MVF_FP8_LN_FOLD=$f
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/c4f$f -o run -- python3 $R/tools/config4_roofline.py --dtype fp8 --steps 4 --serial > $out/c4f$f.log 2>&1
  cp $out/c4f$f/run_kernel_stats.csv $out/config4_fp8_fold${f}_serial_kernel_stats.csv
  rm -rf $out/c4f$f
done
head -9 $out/config4_fp8_fold0_serial_kernel_stats.csv | cut -c1-150
head -9 $out/config4_fp8_fold1_serial_kernel_stats.csv | cut -c1-150
