"""Diagnostic: backbone forward of 256 frames as concurrent lanes of UNEQUAL sizes.  A lane of 110 frames has
ceil(110*197/256) = 85 row tiles: 255 / 765 / 1020 tiles for N = 768 / 2304 / 3072 -- whole rounds on 256 CUs."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_rep_learning_amd import _lib, ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from vit_streams_probe import random_sd   # noqa

def main():
    dev = torch.device('cuda')
    pk = ops.PackedViT(random_sd(), 12, 768, 12, 16, 224, [5, 7, 9, 11], 'bf16')
    F, np_ = 256, 196
    x = torch.randn(F, 3, 224, 224, device=dev)
    lib = _lib.load()
    splits = [[128, 128]] + [[a, 256 - a] for a in range(132, 200, 4)] + [[128, 128], [146, 110], [110, 110, 36], [110, 73, 73]]
    if len(sys.argv) > 1:
        splits = [[int(v) for v in arg.split(',')] for arg in sys.argv[1:]]
    for sp in splits:
        assert sum(sp) == F
        streams = [torch.cuda.Stream() for _ in sp]
        wss = [torch.empty(lib.mvf_vit_workspace_bytes(pk.code, f, np_ + 1, 768, 16), device=dev, dtype=torch.uint8) for f in sp]
        taps = [torch.empty(F * np_, 768, device=dev, dtype=torch.bfloat16) for _ in range(4)]
        cls = torch.empty(F, 768, device=dev)
        offs = [sum(sp[:i]) for i in range(len(sp))]
        def run():
            for s, (f, o) in enumerate(zip(sp, offs)):
                tab = (ctypes.c_void_p * 4)(*[t.data_ptr() + o * np_ * 768 * 2 for t in taps])
                _lib.call('mvf_vit_fwd', ctypes.byref(pk.struct), pk.code, x.data_ptr() + o * 3 * 224 * 224 * 4, f, tab,
                          cls.data_ptr() + o * 768 * 4, wss[s].data_ptr(), wss[s].numel(), f, 0, ctypes.c_void_p(streams[s].cuda_stream))
        for _ in range(3): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): run()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print('lanes %-16s %.3f ms' % (sp, dt * 1e3), flush=True)
main()
