# Socket power and clocks while the product GEMM loops on random / all-zero operands (GPU box; rocm-smi read-only):
#   bash tools/power_probe.sh
cd $GRAFT_REPO_ROOT
rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk|mclk|fclk" | head -8
for mode in random zeros; do
  echo "== $mode operands"
  python - <<PY &
import torch, sys, os
sys.path.insert(0, os.getcwd())
from video_rep_learning_amd import _lib
M = N = 4096; K = 4096
A = torch.randn(M, K, device='cuda').to(torch.bfloat16); W = (torch.randn(N, K, device='cuda') * 0.02).to(torch.bfloat16)
if "$mode" == "zeros":
    A.zero_(); W.zero_()
b = torch.zeros(N, device='cuda'); C = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
_lib.call('mvf_gemm_tc_select', 5)
import time
t0 = time.time(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
while time.time() - t0 < 6.0:
    e0.record()
    for _ in range(2000):
        _lib.call('mvf_gemm_tc', _lib.BF16, 0, A.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), C.data_ptr(), N, None, 0, None, 0, None, None, 197, M, N, K, st)
    e1.record(); torch.cuda.synchronize()
    print('   %.1f TFLOP/s' % (2.0 * M * N * K * 2000 / (e0.elapsed_time(e1) * 1e-3) / 1e12), flush=True)
PY
  pid=$!
  sleep 2.5
  for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk" | tr '\n' ' '; echo; sleep 0.5; done
  wait $pid
done
