# HBM-side bytes per launch of the fused qkv + attention kernel (vit_qkv_attn.hip) at BASELINE configs[1] size: two separate
# rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined with a trace), (2 x FETCH_SIZE + WRITE_SIZE) x 1024 as the
# guide's gfx950 correction prescribes, first (cold) launch dropped.  GPU box: bash tools/pmc_traffic_qkv_attn.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
o=$R/gpurun_out/pmc_qkv_attn
rm -rf $o; mkdir -p $o
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/fetch -o p -- python3 $R/tools/energy_probe.py --kernels qkv_attn --seconds 0.02 > $o/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/write -o p -- python3 $R/tools/energy_probe.py --kernels qkv_attn --seconds 0.02 > $o/write.log 2>&1
python3 - <<PY
import csv, glob, json
def mean(d, c):
    f = glob.glob('$o/' + d + '/**/*counter_collection.csv', recursive=True)[0]
    v = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if r['Counter_Name'] == c and 'vit_qkv_attn_kernel' in r['Kernel_Name']]
    return sum(v[1:]) / max(len(v) - 1, 1), len(v)
fe, n1 = mean('fetch', 'FETCH_SIZE'); wr, n2 = mean('write', 'WRITE_SIZE')
M, D = 256 * 197, 768
algo = M * D * 2 * 2 + 3 * D * D * 2 + M * 12 * 8      # xb read, attention output written, W once, the LayerNorm partial sums
print(json.dumps({'kernel': 'vit_qkv_attn_kernel<false>', 'launches': [n1, n2], 'fetch_size_kb': fe, 'write_size_kb': wr,
                  'hbm_bytes_per_launch': (2 * fe + wr) * 1024, 'algorithmic_bytes_per_launch': algo,
                  'ratio': round((2 * fe + wr) * 1024 / algo, 3),
                  'the_two_launches_it_replaces_algorithmic': M * D * 2 + M * 3 * D * 2 + M * 3 * D * 2 + M * D * 2}, indent=1))
PY
find $o -name "*.db" -delete
