"""Turns the rocprofv3 PMC passes of tools/gemm_bench.py into profiles/pmc_traffic.json (read by bench.py for
`roofline.traffic`).  Usage on the GPU box (each counter in its OWN pass, --pmc never combined with tracing):

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/gemm_bench.py --iters 3 --warm 1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 tools/gemm_bench.py --iters 3 --warm 1
    python3 tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write 4 > gpurun_out/pmc_traffic.json

HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: on gfx950 FETCH_SIZE reports exactly half of a wide
coalesced read stream (MI355X_MICROARCH.md, section HBM); WRITE_SIZE is exact for 16-byte streaming stores."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import GEMM_NAMES, gemm_source_sha  # noqa: E402

ORDER = [('qkv', (0, 2304, 768)), ('proj', (0, 768, 768)), ('fc1', (1, 3072, 768)), ('fc2', (2, 768, 3072))]
FRAMES, TOKENS = 256, 197


def per_shape(directory, counter, per):
    f = glob.glob(os.path.join(directory, '**', '*_counter_collection.csv'), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if 'gemm_tc256_kernel' in r['Kernel_Name'] and r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    assert len(rows) == per * len(ORDER), (len(rows), per)
    out = {}
    for i, (short, key) in enumerate(ORDER):
        vals = [float(r['Counter_Value']) for r in rows[i * per:(i + 1) * per]]
        out[key] = sum(vals[1:]) / max(len(vals) - 1, 1)      # drop the first (cold) launch
    return out


def main():
    fetch_dir, write_dir, per = sys.argv[1], sys.argv[2], int(sys.argv[3])
    fe, wr = per_shape(fetch_dir, 'FETCH_SIZE', per), per_shape(write_dir, 'WRITE_SIZE', per)
    res = {}
    M = FRAMES * TOKENS
    for short, key in ORDER:
        epi, n, k = key
        # resid epilogue (fc2) reads + writes fp32 and reads the bf16 deferred attention-branch output; the others write bf16
        out_bytes = M * n * (4 + 4 + 2 if epi == 2 else 2)
        algo = M * k * 2 + n * k * 2 + out_bytes
        hbm = (2.0 * fe[key] + wr[key]) * 1024.0
        res[GEMM_NAMES[key]] = {'fetch_size_kb': fe[key], 'write_size_kb': wr[key], 'hbm_bytes_per_launch': hbm,
                                'algorithmic_bytes_per_launch': algo, 'ratio': round(hbm / algo, 3), 'M': M}
    # ties the numbers to the kernel source they were measured on: bench.py reports `traffic: null` when the GEMM has changed
    res['_source_sha'] = gemm_source_sha()
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
