"""ctypes binding of libmvf_hip.so (the C ABI declared in include/mvf_hip.h).

There is deliberately NO fallback: if the library is missing or a call returns non-zero this raises.
The product path never computes on the CPU."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MVF_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libmvf_hip.so')

F32, BF16, FP8, F16 = 0, 1, 2, 3
EPI_STORE, EPI_GELU, EPI_RESID, EPI_PATCH = 0, 1, 2, 3

_P, _I, _L, _Z, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_size_t, ctypes.c_float

# name -> argument kinds (p pointer, i int, l long, z size_t, f float, u uint64_t); every function returns int
# except mvf_vit_workspace_bytes (size_t).  Kept in the order of include/mvf_hip.h.
SIGNATURES = {
    'mvf_vit_workspace_bytes': 'iiiii',
    'mvf_vit_fwd': 'pipipppziip',
    'mvf_vit_fwd_x': 'pipippppziip',
    'mvf_vit_blocks_fwd': 'pipiiipzip',
    'mvf_prof_enable': 'i',
    'mvf_prof_collect': 'ppppppip',
    'mvf_gemm_tc': 'iipipippipipippiiiip',
    'mvf_gemm_tc_batched': 'ipipipipiiiiiip',
    'mvf_gemm_tc_ln': 'iipipippipipipipipppiiip',
    'mvf_ln_stats_finalize': 'pipiifp',
    'mvf_gemm_tc_ln_part': 'ipipippipifpiiip',
    'mvf_layernorm_add_fwd': 'ipzpzpppziifp',
    'mvf_gemm_tc_resid2': 'pipippipipiiiiip',
    'mvf_gemm_tc_f32': 'pipippipiiip',
    'mvf_gemm_tc_batched_f32': 'pipipiiiiiip',
    'mvf_gelu_bf16': 'ppzp',
    'mvf_gelu_bwd_bf16': 'pppzp',
    'mvf_grad_prep': 'ippppiiiip',
    'mvf_ln_bwd_block': 'ppppppppiifp',
    'mvf_quant_mxfp8': 'ipzpzpiip',
    'mvf_layernorm_mxfp8': 'pzpppzpiifp',
    'mvf_gemm_fp8': 'ipippipppippipipiiiip',
    'mvf_gemm_fp8_ln': 'ipippipppipipipipipippppiiip',
    'mvf_gemm_tc_select': 'i',
    'mvf_gemm_tc_debug_stamps': 'p',
    'mvf_gemm_tc_debug_rowmask': 'i',
    'mvf_gemm_tc_debug_ktile': 'i',
    'mvf_gemm_tc_debug_ablate': 'i',
    'mvf_gemm_tc_set_cus': 'i',
    'mvf_gemm_tc_set_spare': 'i',
    'mvf_gemm_tc_get_wgs': 'p',
    'mvf_gemm_tc_set_ngroup': 'i',
    'mvf_debug_xcc_map': 'piiip',
    'mvf_patchify': 'ippiiiip',
    'mvf_layernorm_fwd': 'ipzpppziifp',
    'mvf_vit_attn_fwd': 'ippiiiiip',
    'mvf_vit_attn_rowsum_rounded': 'ii',
    'mvf_vit_attn_q_prescaled': 'ii',
    'mvf_vit_qkv_attn_fwd': 'ipipppp' + 'pif' + 'piiiip',
    'mvf_cast_f32_bf16': 'ppzp',
    'mvf_cast_f32_f16': 'ppzp',
    'mvf_cast_bf16_f32': 'ppzp',
    'mvf_vit_attn_fwd_lse': 'pppiiiip',
    'mvf_vit_attn_fwd_mxfp8': 'pppiiiip',
    'mvf_vit_attn_bwd': 'ppppppiiiiip',
    'mvf_static_query_fwd': 'ppplpiiip',
    'mvf_static_query_bwd': 'plppplppplIIIp'.replace('I', 'i'),
    'mvf_hgemm': 'pllpllplppllii' + 'iiifiip',
    'mvf_hgemm_ex': 'pllpllplppllii' + 'iiifii' + 'plfuup',
    'mvf_hlinear_bwd': 'plplplplplpiiiip',
    'mvf_colsum': 'pliipip',
    'mvf_colsum_split': 'pliiipp',
    'mvf_relu_bwd': 'pppzp',
    'mvf_transpose_chunks': 'ppiiip',
    'mvf_sum_batches': 'ppizip',
    'mvf_colscale': 'ppppiiip',
    'mvf_gelu_fwd': 'ppzp',
    'mvf_gelu_bwd': 'pppzp',
    'mvf_dropout_add': 'pppzfuup',
    'mvf_ln_fwd': 'ppppppiifp',
    'mvf_ln_bwd': 'ppppppppiiiip',
    'mvf_ln_bwd_res': 'pppppppppiiip',
    'mvf_bn_workspace_floats': 'ii',
    'mvf_bn_stats': 'piippppfpzp',
    'mvf_syncbn_merge': 'piifppppfp',
    'mvf_bn_fwd': 'ppppppiifip',
    'mvf_bn_bwd_reduce': 'ppppppppppiiifipzp',
    'mvf_bn_bwd_apply': 'pppppppppiififp',
    'mvf_concat_onehot': 'ppiiiip',
    'mvf_final_reduce_fwd': 'pppiiiiip',
    'mvf_final_reduce_bwd': 'pppiiiiip',
    'mvf_l2norm_fwd': 'pppiifp',
    'mvf_l2norm_bwd': 'ppppiifp',
    'mvf_tattn_select': 'i',
    'mvf_tattn_fwd': 'ppippiiiip',
    'mvf_tattn_bwd': 'ppippppiiiip',
    'mvf_lstp_fused_fwd': 'piiiiiiipifppp',
    'mvf_lstp_fused_bwd': 'piiiiiiipppfpp',
    'mvf_lstp_select': 'i',
    'mvf_lstp_scores': 'piiiiiiipipp',
    'mvf_lstp_wsum': 'piiiiiiippp',
    'mvf_lstp_softmax_fwd': 'ppppiiifip',
    'mvf_lstp_softmax_bwd': 'pppppiiifp',
    'mvf_lstp_reduce_frames': 'ppiiiip',
    'mvf_token_pool': 'piiiiiipp',
    'mvf_lstp_dx': 'piiiiiippppip',
    'mvf_head_pack_weights': 'pip',
    'mvf_head_chain_debug': 'i',
    'mvf_head_l2_warm': 'pzp',
    'mvf_head_chain_debug_stamps': 'p',
    'mvf_rowlin_debug_stamps': 'pi',
    'mvf_head_pack_elems': 'iii',
    'mvf_enc_layer_fwd': 'pp',
    'mvf_enc_layer_bwd': 'pp',
    'mvf_head_dw': 'piiip',
    'mvf_rowlin_fwd': 'pp',
    'mvf_rowlin_bwd': 'pp',
    'mvf_scl_rows': 'ppppiip',
    'mvf_scl_fwd': 'pppppppppiiiiffp',
    'mvf_scl_bwd': 'pppppppppiiiiiiffp',
    'mvf_scl_select': 'i',
    'mvf_grad_norm': 'pzpppp',
    'mvf_adam_step': 'ppppzfffffifpfip',
    'mvf_optim_set_width': 'i',
    'mvf_augment_workspace_bytes': 'iii',
    'mvf_augment_clips': 'ppiiiiippzp',
}
_KIND = {'p': _P, 'i': _I, 'l': _L, 'z': _Z, 'f': _F, 'u': ctypes.c_uint64}


class MvfVitWeights(ctypes.Structure):
    """Mirror of `struct MvfVitWeights` (include/mvf_hip.h)."""
    _fields_ = ([('depth', _I), ('dim', _I), ('heads', _I), ('patch', _I), ('img', _I), ('n_taps', _I),
                 ('taps', _I * 8), ('ln_eps', _F),
                 ('cls_token', _P), ('pos_embed', _P), ('patch_w', _P), ('patch_b', _P), ('norm_w', _P), ('norm_b', _P)]
                + [(n, ctypes.POINTER(_P)) for n in
                   ('ln1_w', 'ln1_b', 'qkv_w', 'qkv_b', 'proj_w', 'proj_b', 'ln2_w', 'ln2_b', 'fc1_w', 'fc1_b',
                    'fc2_w', 'fc2_b', 'ls1', 'ls2', 'qkv_c', 'fc1_c', 'qkv_s', 'proj_s', 'fc1_s', 'fc2_s')]
                + [('q_prescaled', _I)])


class MvfAugmentParams(ctypes.Structure):
    """Mirror of `struct MvfAugmentParams` (include/mvf_hip.h)."""
    _fields_ = [('crop_top', _I), ('crop_left', _I), ('crop_h', _I), ('crop_w', _I), ('flip', _I), ('n_color', _I),
                ('color_op', _I * 4), ('color_factor', _F * 4), ('blur_kx', _I), ('blur_ky', _I), ('blur_sigma', _F),
                ('gray', _I), ('mean', _F * 3), ('std', _F * 3)]


class MvfDrop(ctypes.Structure):
    """Mirror of `struct MvfDrop` (include/mvf_hip.h)."""
    _fields_ = [('p', _F), ('seed', ctypes.c_uint64), ('offset', ctypes.c_uint64)]


class MvfPackEntry(ctypes.Structure):
    """Mirror of `struct MvfPackEntry`."""
    _fields_ = [('w', _P), ('ld', _L), ('N', _I), ('K', _I), ('w16', _P), ('w16t', _P), ('f16', _I)]


class MvfEncFwd(ctypes.Structure):
    """Mirror of `struct MvfEncFwd`."""
    _fields_ = ([('M', _I), ('D', _I), ('DFF', _I), ('Mp', _I), ('ln_eps', _F)]
                + [(n, _P) for n in ('o', 'x_in', 'wo', 'w1', 'w2', 'bo', 'b1', 'b2', 'ln1_g', 'ln1_b')]
                + [('drop_attn', MvfDrop), ('drop_ffn', MvfDrop)]
                + [(n, _P) for n in ('x1', 'mean1', 'rstd1', 'a', 'x2', 'oT', 'h1T', 'aT', 'wqkv', 'bqkv', 'ln0_g', 'ln0_b',
                                     'qkv', 'mean0', 'rstd0', 'h0T')]
                + [('f16', _I)])


class MvfEncBwd(ctypes.Structure):
    """Mirror of `struct MvfEncBwd`."""
    _fields_ = ([('M', _I), ('D', _I), ('DFF', _I), ('Mp', _I)]
                + [(n, _P) for n in ('dqkv', 'wqkvT', 'x_in', 'mean0', 'rstd0', 'ln0_g', 'dln0_g', 'dln0_b', 'dqkvT', 'dres',
                                     'dx_out')]
                + [('drop_ffn', MvfDrop), ('drop_attn', MvfDrop)]
                + [(n, _P) for n in ('w2T', 'w1T', 'woT', 'a', 'x1', 'mean1', 'rstd1', 'ln1_g', 'dln1_g', 'dln1_b', 'g2T', 'duT',
                                     'goT', 'dx1_out', 'd_o')])


class MvfRowLinFwd(ctypes.Structure):
    """Mirror of `struct MvfRowLinFwd`."""
    _fields_ = [('M', _I), ('Cin', _I), ('N', _I), ('Mp', _I), ('X', _P), ('ldx', _L), ('g_ntok', _I), ('g_T', _I), ('g_mode', _I),
                ('g_arg', _P), ('bn_mean', _P), ('bn_var', _P), ('bn_g', _P), ('bn_b', _P), ('bn_eps', _F), ('bn_relu', _I),
                ('oh_ntok', _I), ('oh_div', _I), ('drop_in', MvfDrop), ('drop_out', MvfDrop), ('w16', _P), ('bias', _P), ('table', _P),
                ('tab_mod', _I), ('l2norm', _I), ('l2_eps', _F), ('Y', _P), ('nrm', _P), ('xT', _P), ('st_part', _P), ('st_mean', _P),
                ('st_var', _P), ('st_rmean', _P), ('st_rvar', _P), ('st_momentum', _F), ('f16', _I)]


class MvfRowLinBwd(ctypes.Structure):
    """Mirror of `struct MvfRowLinBwd`."""
    _fields_ = [('M', _I), ('Cin', _I), ('N', _I), ('Mp', _I), ('dY', _P), ('nb_Y', _P), ('nb_mean', _P), ('nb_var', _P), ('nb_g', _P),
                ('nb_s1', _P), ('nb_s2', _P), ('nb_eps', _F), ('nb_count', _F), ('drop_out', MvfDrop), ('drop_in', MvfDrop),
                ('l2norm', _I), ('l2_y', _P), ('l2_nrm', _P), ('l2_eps', _F), ('w16t', _P), ('gT', _P), ('oh_ntok', _I), ('X', _P),
                ('ldx', _L), ('bn_mean', _P), ('bn_var', _P), ('bn_g', _P), ('bn_b', _P), ('bn_eps', _F), ('bn_relu', _I),
                ('st_part', _P), ('s1', _P), ('s2', _P), ('dgamma', _P), ('dbeta', _P), ('g_ntok', _I), ('g_T', _I), ('g_mode', _I),
                ('g_arg', _P), ('dX', _P), ('lddx', _L)]


class MvfDwProblem(ctypes.Structure):
    """Mirror of `struct MvfDwProblem`."""
    _fields_ = [('gT', _P), ('xT', _P), ('dw', _P), ('lddw', _L), ('db', _P), ('N', _I), ('K', _I)]


class MvfError(RuntimeError):
    pass


_lib = None


def load():
    """Loads (once) and returns the ctypes handle; raises if the HIP library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: the wheel carries its own libamdhip64; if this library pulled in /opt/rocm's copy before torch is
    # imported, the process would hold two HIP runtimes and every launch from here fails with hipErrorNoDevice (100)
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise MvfError('libmvf_hip.so not found at %s -- build it with `python -m video_rep_learning_amd.csrc.build` '
                       '(there is no CPU fallback)' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, sig in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.argtypes = [_KIND[k] for k in sig]
        fn.restype = _Z if name in ('mvf_vit_workspace_bytes', 'mvf_bn_workspace_floats', 'mvf_augment_workspace_bytes',
                                    'mvf_head_pack_elems') else _I
    _lib = lib
    return lib


def call(name, *args):
    """Calls an int-returning entry point and raises MvfError on a non-zero code."""
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise MvfError('%s failed with code %d%s' % (name, rc, ' (bad argument)' if rc == 10001 else ''))


def try_call(name, *args):
    """call() for entry points that may decline a shape: returns False on MVF_ERR_UNSUPPORTED (10002), True on success,
    raises on any other code."""
    rc = getattr(load(), name)(*args)
    if rc == 10002:
        return False
    if rc != 0:
        raise MvfError('%s failed with code %d%s' % (name, rc, ' (bad argument)' if rc == 10001 else ''))
    return True


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  The tensor must be a contiguous CUDA(HIP) tensor."""
    if t is None:
        return None
    if not t.is_cuda:
        raise MvfError('HIP op received a %s tensor: the MV-Former kernels only run on a gfx950 device' % t.device)
    if not t.is_contiguous():
        raise MvfError('HIP op received a non-contiguous tensor')
    return t.data_ptr()
