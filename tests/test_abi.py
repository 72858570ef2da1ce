"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports exactly the
entry points include/mvf_hip.h declares with the argument kinds the ctypes binding assumes.  No compute calls
(there is no GPU here); bad-argument paths that return before any launch ARE exercised."""
import ctypes
import os

import pytest

from abi_util import header_signatures, ROOT
from video_rep_learning_amd import _lib


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        from video_rep_learning_amd.csrc import build
        build.build()
    return _lib.load()


def test_every_declared_symbol_is_exported_and_bound(lib):
    decl = header_signatures()
    assert len(decl) >= 37
    for name, kinds in decl.items():
        assert hasattr(lib, name), 'declared in include/mvf_hip.h but not exported: %s' % name
        assert name in _lib.SIGNATURES, 'no ctypes signature for %s' % name
        assert _lib.SIGNATURES[name] == kinds, (name, _lib.SIGNATURES[name], kinds)
    assert set(_lib.SIGNATURES) == set(decl), set(_lib.SIGNATURES) ^ set(decl)


def test_header_cites_the_reference():
    src = open(os.path.join(ROOT, 'include', 'mvf_hip.h')).read()
    for cite in ('models/transformer.py', 'models/mvformer.py', 'models/utils.py', 'algos/scl.py', 'train.py',
                 'utils/optimizer.py', 'resnet_c2d.py'):
        assert cite in src, cite


def test_argument_errors_are_reported_before_any_launch(lib):
    # NULL pointers / bad shapes are rejected on the host with MVF_ERR_ARG (10001) -- no GPU needed
    assert lib.mvf_gemm_tc(_lib.BF16, 0, None, 0, None, 0, None, None, 0, None, 0, None, 0, None, None, 0, 0, 0, 0, None) == 10001
    assert lib.mvf_scl_fwd(None, None, None, None, None, None, None, None, None, 0, 0, 0, 0, 0.1, 10.0, None) == 10001
    assert lib.mvf_gemm_tc_select(8) == 10001
    assert lib.mvf_gemm_tc_select(0) == 0
    assert lib.mvf_vit_workspace_bytes(_lib.BF16, 256, 197, 768, 16) >= 256 * 197 * 768 * (4 + 2 + 6 + 8)
    with pytest.raises(_lib.MvfError):
        _lib.call('mvf_layernorm_fwd', 0, None, 0, None, None, None, 0, 0, 0, 1e-6, None)


def test_softmax_rowsum_convention_is_the_same_in_library_and_oracle(lib):
    """The 16-bit attention kernels normalise by the sum of the ROUNDED probabilities for some token counts and by the fp32 sum for the
    others; the emulating oracle must follow the library's dispatch, not a copy of it (a host-side query: no launch)."""
    import sys
    sys.path.insert(0, ROOT)
    from oracle import vit as OV
    for dt in (_lib.BF16, _lib.F16):
        for n in range(1, 2049):
            assert bool(lib.mvf_vit_attn_rowsum_rounded(dt, n)) == OV.rowsum_rounded(n), (dt, n)
    assert lib.mvf_vit_attn_rowsum_rounded(_lib.F32, 197) == 0
    # the pre-scaled-q convention of frozen backbones (q rows of the packed qkv weights carry log2(e) / 8): packer, kernels and the
    # emulating oracle follow ONE statement of where it applies
    for dt in (_lib.BF16, _lib.F16, _lib.FP8):
        for n in range(1, 2049):
            assert bool(lib.mvf_vit_attn_q_prescaled(dt, n)) == OV.q_prescaled(n), (dt, n)
    assert lib.mvf_vit_attn_q_prescaled(_lib.F32, 785) == 0 and OV.q_prescaled(785) and not OV.q_prescaled(197)


def test_product_path_refuses_cpu_tensors():
    import torch
    from video_rep_learning_amd import ops
    x = torch.randn(4, 8)
    with pytest.raises(_lib.MvfError):
        ops.linear(x, torch.randn(3, 8))
    with pytest.raises(_lib.MvfError):
        ops.vit_forward(torch.randn(1, 3, 32, 32), None)


def test_struct_layouts_match_the_header(tmp_path):
    """The ctypes mirrors of MvfVitWeights / MvfAugmentParams against the C compiler's view of include/mvf_hip.h (gcc, host
    only: sizes and the offsets of a few members)."""
    import shutil
    import subprocess
    if shutil.which('gcc') is None or not os.path.exists('/opt/rocm/include/hip/hip_runtime_api.h'):
        pytest.skip('needs gcc and the HIP headers')
    probes = [('MvfVitWeights', ['depth', 'taps', 'ln_eps', 'cls_token', 'patch_w', 'ln1_w', 'fc2_b', 'ls2']),
              ('MvfAugmentParams', ['crop_top', 'flip', 'color_op', 'color_factor', 'blur_kx', 'blur_sigma', 'gray', 'mean', 'std']),
              ('MvfDrop', ['p', 'seed', 'offset']),
              ('MvfPackEntry', ['w', 'ld', 'N', 'K', 'w16', 'w16t']),
              ('MvfEncFwd', ['M', 'Mp', 'ln_eps', 'o', 'wo', 'ln1_b', 'drop_attn', 'drop_ffn', 'x1', 'a', 'x2', 'aT', 'wqkv', 'qkv', 'h0T']),
              ('MvfEncBwd', ['M', 'Mp', 'dqkv', 'dqkvT', 'dres', 'dx_out', 'drop_ffn', 'drop_attn', 'w2T', 'a', 'dln1_b', 'goT', 'd_o']),
              ('MvfDwProblem', ['gT', 'xT', 'dw', 'lddw', 'db', 'N', 'K']),
              ('MvfRowLinFwd', ['M', 'Mp', 'X', 'ldx', 'g_ntok', 'g_arg', 'bn_mean', 'bn_eps', 'bn_relu', 'oh_ntok', 'drop_in', 'drop_out', 'w16',
                                'table', 'tab_mod', 'l2norm', 'l2_eps', 'Y', 'xT', 'st_part', 'st_rvar', 'st_momentum']),
              ('MvfRowLinBwd', ['M', 'Mp', 'dY', 'nb_Y', 'nb_s2', 'nb_eps', 'nb_count', 'drop_out', 'drop_in', 'l2norm', 'l2_y', 'l2_eps',
                                'w16t', 'gT', 'oh_ntok', 'X', 'ldx', 'bn_mean', 'bn_eps', 'bn_relu', 'st_part', 'dbeta', 'g_ntok', 'g_arg',
                                'dX', 'lddx'])]
    body = ''.join('printf("%s %%zu", sizeof(%s));%sprintf("\\n");\n' % (
        st, st, ''.join('printf(" %%zu", offsetof(%s, %s));' % (st, f) for f in fs)) for st, fs in probes)
    src = tmp_path / 'layout.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mvf_hip.h"\nint main(void) {\n%s return 0; }\n' % body)
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', '-D__HIP_PLATFORM_AMD__', '-I/opt/rocm/include', '-I' + os.path.join(ROOT, 'include'), str(src), '-o',
                    str(exe)], check=True, capture_output=True)
    lines = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    for (st, fs), line in zip(probes, lines):
        name, size, *offs = line.split()
        cls = getattr(_lib, st)
        assert name == st and int(size) == ctypes.sizeof(cls), (st, size, ctypes.sizeof(cls))
        for f, o in zip(fs, offs):
            assert getattr(cls, f).offset == int(o), (st, f, o, getattr(cls, f).offset)


def test_spare_cu_policy_follows_the_heads_rows_and_the_mode(lib, monkeypatch):
    """ops._spare_cus_for_rows / gemm_spare_mode (host logic only: the setter launches nothing): 32 spare CUs up to 1 024 head rows, the
    row-chain workgroup count rounded up to 8 beyond, 64 at most; none while the backbone runs on its own (evaluation); MVF_GEMM_SPARE pins."""
    from video_rep_learning_amd import ops
    monkeypatch.delenv('MVF_GEMM_SPARE', raising=False)
    seen = []
    monkeypatch.setattr(ops, 'call', lambda name, *a: seen.append((name, a)) or 0)
    monkeypatch.setattr(ops, '_SPARE_SET', [None])
    monkeypatch.setattr(ops, '_SPARE_MODE', [True])
    for rows, want in ((96, 32), (768, 32), (1024, 32), (1536, 48), (1537, 56), (4096, 64)):
        ops._spare_cus_for_rows(rows)
        assert ops._SPARE_SET[0] == want, (rows, ops._SPARE_SET)
    n = len(seen)
    ops._spare_cus_for_rows(4096)                 # unchanged: no call
    assert len(seen) == n
    ops.gemm_spare_mode(False)
    assert ops._SPARE_SET[0] == 0 and seen[-1] == ('mvf_gemm_tc_set_spare', (0,))
    ops._spare_cus_for_rows(768)                  # evaluation: stays off
    assert ops._SPARE_SET[0] == 0
    ops.gemm_spare_mode(True)
    assert ops._SPARE_SET[0] == 32
    monkeypatch.setenv('MVF_GEMM_SPARE', '16')    # pinned by the environment: the policy keeps its hands off
    n = len(seen)
    ops._spare_cus_for_rows(4096)
    ops.gemm_spare_mode(False)
    assert len(seen) == n
    assert lib.mvf_gemm_tc_set_spare(257) == 10001 and lib.mvf_gemm_tc_set_spare(32) == 0
    assert lib.mvf_optim_set_width(300) == 10001 and lib.mvf_optim_set_width(64) == 0
