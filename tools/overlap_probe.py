"""Diagnostic: does the one-batch backbone lookahead (side stream) shorten the step, per GEMM variant?
Runs bench.py's loop in-process for variant in {0 persistent, 3 one workgroup per tile, 1 128x128 tiles}."""
import json
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for variant in (0, 3, 1):
    for la in (True, False):
        code = ("import sys; sys.argv=['bench.py','--no-cpu-baseline','--steps','20','--warmup','4'%s];"
                "import torch; torch.zeros(1, device='cuda'); from video_rep_learning_amd import _lib; _lib.call('mvf_gemm_tc_select', %d);"
                "import runpy; runpy.run_path('%s/bench.py', run_name='__main__')") % (
                    '' if la else ",'--no-lookahead'", variant, ROOT)
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd=ROOT)
        line = [l for l in r.stdout.splitlines() if l.startswith('{')]
        if not line:
            print('variant', variant, 'lookahead', la, 'FAILED', r.stderr[-500:])
            continue
        j = json.loads(line[-1])
        print('variant %d lookahead %-5s  %.3f ms/step  %.1f clips/s' % (variant, la, j['ms_per_step'], j['value']), flush=True)
