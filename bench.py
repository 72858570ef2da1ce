#!/usr/bin/env python3
"""Headline benchmark: full MV-Former SCL training steps (frozen ViT-B/16 forward + head forward/backward + SCL loss +
gradient all-reduce + clip + Adam) on synthetic clips of BASELINE.json configs[1]:
PennAction MV-Former, ViT-B/16, 32 frames, batch 4 per GPU (8 clips/GPU/step), bf16 backbone.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Started by a launcher (WORLD_SIZE set: `python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N ...`) this process IS a rank; started bare (`python bench.py --gpus N`) it starts that
launcher itself as a fresh child process BEFORE touching the GPU and relays the child's output and exit code.

Rank 0 prints ONE JSON line.  `value` = clips/s over all N GPUs with the inputs already resident in HBM.
`roofline` = the dominant kernel (bf16 MFMA GEMM of the backbone), timed per launch with HIP events on the launch
stream during extra profiled steps right after the timed region.  `cpu_baseline` = the CPU oracle (a port of the
reference's algorithm; the reference itself cannot run on CPU) on a bounded sample, rank 0, N=1 only.
`parity` (same leg) = the HIP path in fp32 AND in the benchmarked bf16 mode against the oracle on real-size clips of the same
workload: fp32 mode must meet the north-star 1e-3 (the run FAILS otherwise), bf16 mode is compared with the oracle that
rounds to bf16 where the kernels store bf16.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# the host driver only supports dmabuf IPC: without this RCCL's buffer exchange fails with `hipIpcGetMemHandle: invalid argument`
# (exported on the GPU boxes already; set here too so that a bare launcher environment cannot lose it)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
# Hardware queues: HIP deals a process's streams over GPU_MAX_HW_QUEUES (default 4) hardware queues, and two streams on one queue run one
# after the other (profiles/r05/order_probe.txt).  A data-parallel rank owns the caller's stream, the lookahead stream, the second backbone
# lane and RCCL's communication stream.  Round 5 asked for 8 queues here, unmeasured; round 6 measured it on the forced one-rank RCCL step
# (profiles/r06/rccl_forced_1rank.txt, three alternations on one box): plain step 10.67-10.73 ms, with the collectives live and the runtime's
# 4 queues 10.85-10.87, with 8 or 16 queues 11.61-11.66 -- the all-reduce then completes at once (0.19 ms exposed instead of 7 ms of queueing
# behind a backbone lane, which the one-batch lookahead hides anyway) but every step pays 0.76 ms for the extra queues.  So the runtime's
# default stays; MVF_HW_QUEUES=<n> asks for n queues (A/B knob).  Read when the runtime loads, hence before torch is imported.
if os.environ.get('MVF_HW_QUEUES', '0') not in ('', '0'):
    os.environ.setdefault('GPU_MAX_HW_QUEUES', os.environ['MVF_HW_QUEUES'])

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA peak, MI355X_MICROARCH.md (no 2:1 sparsity)
# what a bf16 GEMM K LOOP sustains on random operands under the chip's power management (measured, not a spec figure:
# profiles/r03/power_probe.txt -- six seconds of back-to-back 4096^3 launches of the product kernel: 1 475 TFLOP/s at 1 355 W
# socket power and 1.81 GHz; a four-wave loop with the matrix pipe 82 % busy lands on the same 1 475; all-zero operands: 1 953 at
# 2.39 GHz / 1 050 W).  Reported beside `peak` for context; `frac` stays priced against the nominal peak.
SUSTAINED_BF16_LOOP_TFLOPS = 1475.0
TFLOP_PER_CLIP = 1.1925       # algorithmic work of config #2 per clip (fwd backbone + f/b head)
PEAK_F32_TFLOPS = 157.3       # fp32-input MFMA (= vector) peak, parity mode
# (epilogue kind, N, K) of the ViT-B/16 GEMMs; M = frames * 197 (patch-embed: frames * 196)
GEMM_NAMES = {(0, 2304, 768): 'qkv [M,768]x[2304,768]^T', (2, 768, 768): 'proj+resid [M,768]x[768,768]^T',
              (0, 768, 768): 'proj (bf16 branch output, residual add deferred) [M,768]x[768,768]^T',
              (1, 3072, 768): 'fc1+gelu [M,768]x[3072,768]^T', (2, 768, 3072): 'fc2+resid [M,3072]x[768,3072]^T',
              (3, 768, 768): 'patch-embed+pos [M,768]x[768,768]^T',
              (5, 2304, 768): 'qkv + attention fused [M,768]x[2304,768]^T, softmax(QK^T)V (vit_qkv_attn_kernel)'}
# bf16 mode against the bf16-emulating oracle, (loss, embeddings): about 3x what the driver-style runs measure
# measured over this round's runs: loss 1.5e-4 .. 1.0e-3 (relative, on a loss that the resident batch drives down to 0.09),
# embeddings 5.7e-4 .. 1.5e-3.  The loss gate is relative to max(|loss|, 0.25): a longer run over-fits the resident batch further
# and a relative error on a vanishing loss says nothing about the kernels
BF16_GATES = (3e-3, 5e-3)
PMC_FILE = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')


def tile_rows(M, N, nwg=256):
    """The 256x256-thread kernel's tile rows for an [M, N] output (csrc/gemm_tc256.hip pick_tile_rows): for launches of three
    or more rounds the height of 256 / 240 / 224 / 208 whose rounds x rows x (1 + penalty) is smallest, else 256."""
    forced = int(os.environ.get('MVF_GEMM_BM', '0'))
    if forced in (256, 240, 224, 208):
        return forced
    allowed = int(os.environ.get('MVF_GEMM_BM_SET', '7'))
    nbn = (N + 255) // 256
    rounds = lambda bm: -(-(-(-M // bm) * nbn) // nwg)
    if rounds(256) < 3:
        return 256
    best, cost = 256, rounds(256) * 256 * 1000
    for i, (h, pen) in enumerate(((240, 1020), (224, 1080), (208, 1110))):
        if not (allowed >> i & 1) or (rounds(256) < 8 if h == 240 else nbn > 3):
            continue
        c = rounds(h) * h * pen
        if c < cost:
            best, cost = h, c
    return best


def rocprof_names(groups, dtype, ln_fold, defer=True, frames=256, tokens=197, nwg=256):
    """The same launches under the kernel names rocprofv3 prints, gemm_tc256_kernel<EPI, DBG, LN, FP8, ABL, ADD2, BMT, F16>: one
    kernel per epilogue flavour and tile height.  LN = true for the launches that carry the LN-fold extras: of a shape's 12
    launches per forward, 11 in fold mode 2 for qkv (blocks 1..11 consume the folded norm1) and for fc2 (blocks 0..10 produce
    bf16(x) + the row sums); 12 for fc1 / proj in the modes that fold norm2.  ADD2 = true for fc2 when the attention branch's
    residual add is deferred (then proj is a plain store, <0, ...>).  BMT = the tile rows the launch picks for the shape in the
    one-lane mode these times are taken in, for `nwg` workgroups per persistent launch (mvf_gemm_tc_get_wgs: 256, or 248 under the
    collectives' CU reserve).  Averages are per shape group (HIP events cannot tell two names of one shape apart)."""
    by = {}
    for r in groups:
        e = r['epi']
        if e == 5:        # the fused qkv + attention launch (csrc/vit_qkv_attn.hip), not a gemm_tc256 instantiation
            parts = [('vit_qkv_attn_kernel<%s>' % ('true' if dtype == 'fp16' else 'false'), 1.0)]
        elif dtype not in ('bf16', 'fp16'):
            parts = [('gemm_tc_kernel<float, %d, false>' % e, 1.0)]
        else:
            qkv, fc1 = e == 0 and r['n'] > r['k'], e == 1        # (epi 0 with N == K: the proj GEMM of the deferred residual)
            fc2, proj = e == 2 and r['k'] > r['n'], e == 2 and r['k'] == r['n']
            frac = 0.0
            if (qkv or fc2) and ln_fold in (1, 2):
                frac = 11.0 / 12.0
            if (fc1 or proj) and ln_fold in (1, 3):
                frac = 1.0
            add2 = 'true' if (fc2 and defer and ln_fold in (0, 2)) else 'false'
            bm = tile_rows(frames * (tokens - 1 if e == 3 else tokens), r['n'], nwg)
            f16 = 'true' if dtype == 'fp16' else 'false'
            parts = [('gemm_tc256_kernel<%d, false, true, false, 0, %s, %d, %s>' % (e, add2, bm, f16), frac),
                     ('gemm_tc256_kernel<%d, false, false, false, 0, %s, %d, %s>' % (e, add2, bm, f16), 1.0 - frac)]
        for k, f in parts:
            if f > 0:
                d = by.setdefault(k, {'launches': 0.0, 'ms': 0.0})
                d['launches'] += r['launches'] * f
                d['ms'] += r['ms'] * f
    return {k: {'launches': int(round(d['launches'])), 'avg_us': round(d['ms'] * 1e3 / max(d['launches'], 1e-9), 1)} for k, d in by.items()}


GEMM_SOURCES = ('gemm_tc256.hip', 'gemm_tc_epi.h')


def gemm_wgs():
    """workgroups of a persistent GEMM launch as the library will size it now (CU budget included)"""
    from video_rep_learning_amd import _lib
    n = ctypes.c_int(0)
    _lib.call('mvf_gemm_tc_get_wgs', ctypes.byref(n))
    return n.value


def gemm_source_sha():
    """sha256 (16 hex digits) over the sources of the kernel `roofline` reports: stored in profiles/pmc_traffic.json by
    tools/pmc_summary.py when the PMC passes are collected, compared here when the file is read back."""
    import hashlib
    h = hashlib.sha256()
    for name in GEMM_SOURCES:
        with open(os.path.join(ROOT, 'video_rep_learning_amd', 'csrc', name), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(name):
    """HBM bytes per launch of the named GEMM shape from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json,
    written by tools/pmc_summary.py from separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of tools/gemm_bench.py:
    2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md section HBM).  The counters cannot be read
    inside this process (rocprofv3 has to wrap it), so the file carries the sha of the kernel sources it was measured on:
    None (and a warning) when it was collected on another version of the GEMM, or not at all."""
    try:
        with open(PMC_FILE) as f:
            doc = json.load(f)
        if doc.get('_source_sha') != gemm_source_sha():
            print('bench.py: profiles/pmc_traffic.json was collected on another version of %s (sha %s, now %s): roofline.traffic '
                  'is null until tools/collect_profiles.sh has been re-run' % ('/'.join(GEMM_SOURCES), doc.get('_source_sha'),
                                                                               gemm_source_sha()), file=sys.stderr, flush=True)
            return None
        rec = doc.get(name)
        return None if rec is None else rec['hbm_bytes_per_launch']
    except (OSError, ValueError, KeyError):
        return None


STAGE = {'name': 'start'}
_KEEP_STREAMS = []


def stage(name):
    STAGE['name'] = name


def start_init_watchdog(rank, seconds):
    """A rank that has not finished its first step `seconds` after main() started (a peer that never arrived at the
    rendezvous, a collective that hangs on first use) ends the run with exit code 4 and names the stage it was in, instead of
    hanging until the driver's own limit.  Returns the event that disarms it."""
    done = threading.Event()
    if seconds <= 0:
        return done

    def run():
        if not done.wait(seconds):
            print('bench.py rank %d: no progress %d s after start, last stage: %s -- giving up (exit 4)'
                  % (rank, seconds, STAGE['name']), file=sys.stderr, flush=True)
            os._exit(4)
    threading.Thread(target=run, daemon=True).start()
    return done


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=50)       # >= 20 warm-up + >= 50 timed
    p.add_argument('--warmup', type=int, default=20)
    p.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32', 'fp16'],
                   help='backbone compute dtype: bf16 (BASELINE configs[1]), fp32 (parity mode), fp16 (the reference\'s own autocast dtype)')
    p.add_argument('--no-cpu-baseline', action='store_true', help='skip the CPU legs (cpu_baseline and parity)')
    p.add_argument('--parity-videos', type=int, default=2,
                   help='videos (x 2 views x 32 frames) of the resident batch pushed through the oracle for the parity block')
    p.add_argument('--profile-steps', type=int, default=3)
    p.add_argument('--min-sustain-s', type=float, default=2.0,
                   help='after the K timed steps the same loop keeps running until this many seconds of steps have been timed in all (0 = '
                        'off): `ms_per_step` stays the K requested steps, `ms_per_step_sustained` is the extension alone -- the chip\'s '
                        'power management needs a second or two of load to settle (bursts read 10-13 %% slow or fast, DESIGN 4a)')
    p.add_argument('--no-lookahead', action='store_true', help='run backbone and head strictly in sequence')
    p.add_argument('--serial', action='store_true',
                   help='one kernel at a time for the whole run (one backbone lane, no lookahead): the mode the roofline '
                        'section is always timed in; use it under rocprofv3 --kernel-trace --stats so that the per-kernel '
                        'averages there are the ones of roofline.rocprof_kernels')
    p.add_argument('--init-timeout', type=int, default=120,
                   help='seconds a rank may take from start to the end of its first step before it exits with code 4 and its '
                        'last stage on stderr (0 = no limit)')
    p.add_argument('--test-hooks', action='store_true',
                   help='honour the MVF_BENCH_BACKEND / MVF_BENCH_SHARE_GPU environment hooks (tests only: two ranks on the one '
                        'GPU of a test box over gloo); without this flag their presence is an error, so that a stray variable '
                        'cannot turn an RCCL measurement into a gloo one')
    p.add_argument('--plumbing', action='store_true',
                   help='no GPU, no kernels, no measurement (needs --test-hooks and MVF_BENCH_BACKEND=gloo): rank handling, process '
                        'group, model / optimizer / gradient-bucket construction on the CPU and one gradient all-reduce, then the '
                        'JSON line with value null -- what tests/test_bench_launch.py checks of the N > 1 path without a device')
    return p.parse_args()


def _oracle_env():
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    from oracle import model as OM
    import test_gpu_model as T
    # the GPU box gives one GPU's share of the host (16 hardware threads); more torch threads than that only thrash
    cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1))
    torch.set_num_threads(cores)
    return OM, T


def cpu_baseline(cfg, model):
    """Oracle train step on a bounded sample: ONE video (2 clips x T frames = a quarter of one GPU batch) of the same
    workload, one warm step + three timed ones, ~6 s each on one GPU's share of the host cores."""
    OM, T = _oracle_env()
    vit_cfg, head_cfg, scl_cfg = T.oracle_cfgs(cfg)
    params = T.cpu_params(model)
    t, s = cfg.TRAIN.NUM_FRAMES, cfg.IMAGE_SIZE
    g = torch.Generator().manual_seed(1234)
    nv, iters = 1, 3
    batch = (torch.randn(nv, 2, t, 3, s, s, generator=g), torch.full((nv, 2), 100, dtype=torch.long),
             torch.sort(torch.randint(0, 100, (nv, 2, t), generator=g), dim=-1)[0], torch.ones(nv, 2, t))
    st = {}
    OM.train_step(batch, params, st, vit_cfg, head_cfg, scl_cfg)          # warm (allocator, thread pool)
    t0 = time.time()
    for _ in range(iters):
        OM.train_step(batch, params, st, vit_cfg, head_cfg, scl_cfg)
    dt = (time.time() - t0) / iters
    return {'value': round(2.0 * nv / dt, 4), 'unit': 'clips/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': 'oracle train steps (fp32) on %d video = %d clips x %d frames (a quarter of one GPU batch): 1 warm + %d '
                      'timed, %.1f s per step' % (nv, 2 * nv, t, iters, dt)}


def m0_prefixes(pre):
    return sorted({q[len('embed.'):] if q.startswith('embed.') else q for q in pre})


def parity_block(model, batch, nv, dev, reduced='bf16'):
    """The HIP path vs the oracle on `nv` videos of the resident benchmark batch (real size: ViT-B/16, 32 frames), dropout 0
    (its masks cannot be replayed on the CPU), training-mode loss (BatchNorm batch statistics) and eval-mode per-frame
    embeddings (project=False).  The oracle's backbone runs twice: plain fp32 and bf16-emulating (oracle/vit.py)."""
    OM, T = _oracle_env()
    from video_rep_learning_amd.utils import presets
    from video_rep_learning_amd.models import build_model
    from video_rep_learning_amd.algos import get_algo
    cfg0 = presets.make_cfg(network='TIMM-vit_base_patch16_224.dino', num_frames=32, batch_size=nv, compute_dtype='fp32',
                            dropout=0.0)
    m0 = build_model(cfg0, 0).to(dev)
    m0.load_state_dict(model.state_dict())
    vit_cfg, head_cfg, scl_cfg = T.oracle_cfgs(cfg0)
    videos, seq_lens, steps, masks = [x[:nv] for x in batch]
    vc, sc, stc, mc = videos.cpu(), seq_lens.cpu(), steps.cpu(), masks.cpu()
    b, v, t = vc.shape[:3]
    params = T.cpu_params(m0)
    t0 = time.time()
    ref = {}
    # the benchmarked run's head dtype (MI355X.HEAD_DTYPE: bf16 beside a bf16 backbone, fp16 beside an fp16 one): the reduced mode of this
    # block runs the same head, and the oracle rounds the operands of the Linears the device runs on the 16-bit matrix cores the same way
    # (oracle/head.py: 'bf16' everywhere, or 'fp16' = forward operands fp16 / gradient operands bf16)
    head_red = model.head_dtype
    m0.set_head_dtype(head_red)
    head_pre = m0.head_bf16_linears()
    head_pre = tuple('embed.' + q for q in head_pre) + head_pre
    m0.set_head_dtype('fp32')
    with torch.no_grad():
        for mode in ('fp32', reduced):
            vcfg = dict(vit_cfg, emulate=reduced) if mode == reduced else vit_cfg
            feat, cls = OM.backbone_features(vc.reshape(b * v * t, *vc.shape[3:]), params, vcfg)
            with T.OH.emulating(head_pre if mode == reduced else (), 'fp16' if head_red == 'fp16' else 'bf16'):
                ref[mode] = (OM.forward_from_backbone(feat, cls, b * v, t, params, vcfg, head_cfg, mc.reshape(b * v, 1, t),
                                                      project=False, training=False),
                             OM.loss_from_backbone(feat, cls, sc, stc, mc, params, vcfg, head_cfg, scl_cfg, training=True))
    algo = get_algo(cfg0)
    out = {'sample': '%d videos = %d clips x %d frames of the resident batch, dropout 0; oracle %.0f s'
                     % (nv, b * v, t, time.time() - t0),
           'head_dtype': head_red, 'head_bf16_linears': list(m0_prefixes(head_pre)),
           'oracle_loss_fp32': round(float(ref['fp32'][1]), 6), 'oracle_loss_%s_emulating' % reduced: round(float(ref[reduced][1]), 6)}
    # Every device pass sees the parameters AND BatchNorm running statistics the oracle's `params` snapshot holds: the
    # train-mode loss pass updates running_mean / running_var even under no_grad, so the eval-mode embeddings of both modes
    # are taken first and the buffers are restored after each loss pass.
    embs, losses = {}, {}
    buffers = {k: b_.clone() for k, b_ in m0.named_buffers()}
    for mode in ('fp32', reduced):
        m0.compute_dtype = mode
        m0.set_head_dtype(head_red if mode == reduced else 'fp32')
        m0.eval()
        with torch.no_grad():
            embs[mode] = m0(videos.reshape(b * v, t, *videos.shape[3:]), t, video_masks=masks.reshape(b * v, 1, t).to(dev))
    for mode in ('fp32', reduced):
        m0.compute_dtype = mode
        m0.set_head_dtype(head_red if mode == reduced else 'fp32')
        m0.train()
        with torch.no_grad():
            losses[mode] = algo.compute_loss(m0, videos, seq_lens, steps, masks)['loss']
            for k, b_ in m0.named_buffers():
                b_.copy_(buffers[k])
    for mode in ('fp32', reduced):
        emb, loss = embs[mode], losses[mode]
        out['hip_loss_' + mode] = round(float(loss), 6)
        out['loss_rel_' + mode] = float('%.3e' % T.relerr(loss, ref[mode][1]))
        out['emb_maxrel_' + mode] = float('%.3e' % T.relerr(emb, ref[mode][0]))
        if mode == reduced:     # the dtype's own error: reduced-precision HIP against the plain fp32 oracle
            out['loss_rel_%s_vs_fp32_oracle' % reduced] = float('%.3e' % T.relerr(loss, ref['fp32'][1]))
            out['emb_maxrel_%s_vs_fp32_oracle' % reduced] = float('%.3e' % T.relerr(emb, ref['fp32'][0]))
    out['gate'] = ('fp32 loss and embeddings <= 1e-3 rel (north star); %s (the benchmarked dtype) against the %s-emulating '
                   'oracle: loss <= %g of max(|loss|, 0.25), embeddings <= %g (eval-mode outputs of the TRAINED head of this run)'
                   % ((reduced, reduced) + BF16_GATES))
    lemu = out['oracle_loss_%s_emulating' % reduced]
    loss_err_red = abs(out['hip_loss_' + reduced] - lemu) / max(abs(lemu), 0.25)
    out['ok'] = bool(out['loss_rel_fp32'] <= 1e-3 and out['emb_maxrel_fp32'] <= 1e-3 and
                     loss_err_red <= BF16_GATES[0] and out['emb_maxrel_' + reduced] <= BF16_GATES[1])
    if reduced == 'fp16':     # the accuracy mode: fp16 backbone + fp16-forward head must itself sit inside the north star's tolerance
        out['north_star_1e-3_met_by_fp16_mode'] = bool(out['emb_maxrel_fp16_vs_fp32_oracle'] <= 1e-3)
    return out


def launch_ranks(a):
    """`python bench.py --gpus N` outside a launcher: start `torch.distributed.run` with N ranks of this script as a CHILD
    process.  Nothing here may touch the GPU (a process that has initialised HIP must never exec or be replaced, and the
    children need the devices): torch.cuda.device_count() does not initialise it on this stack."""
    have = torch.cuda.device_count()
    if have < a.gpus:
        raise SystemExit('bench.py --gpus %d needs %d GPUs on this node, found %d' % (a.gpus, a.gpus, have))
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus), '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def plumbing(a, world, rank, first_step_done):
    """bench.py's N > 1 control flow without a device: gloo process group, the benchmark's model / optimizer / gradient buckets
    built on the CPU, ONE bucketed gradient all-reduce through GradReducer (the collectives of a step's tail), the barrier +
    max-over-ranks reduction of the timing, and rank 0's JSON line with `value` null.  No kernel runs and nothing is measured."""
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29511')
    stage('init_process_group(gloo) rendezvous at %s:%s' % (os.environ['MASTER_ADDR'], os.environ['MASTER_PORT']))
    dist.init_process_group('gloo', init_method='env://', world_size=world, rank=rank)
    from video_rep_learning_amd.utils import presets
    from video_rep_learning_amd.utils.optimizer import construct_optimizer
    from video_rep_learning_amd.models import build_model
    from video_rep_learning_amd.train import DataParallelModel
    stage('build_model (cpu)')
    cfg = presets.baseline_config_2(compute_dtype=a.dtype)
    torch.manual_seed(cfg.RNG_SEED)
    model = build_model(cfg, -1)
    wrapped = DataParallelModel(model)
    opt = construct_optimizer(wrapped, cfg)
    stage('gradient all-reduce (gloo)')
    red = opt.reducer
    opt.flat.flat_g.fill_(float(rank + 1))
    scale = red.finish()
    expect = sum(range(1, world + 1))
    assert bool((opt.flat.flat_g == expect).all()) and abs(scale * dist.get_world_size() - 1.0) < 1e-12
    first_step_done.set()
    tmax = torch.tensor([float(rank)], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.barrier()
    if rank == 0:
        ws = dist.get_world_size()
        print(json.dumps({'metric': 'video-clips/sec/node, ViT-B/16 32-frame MV-Former', 'value': None, 'unit': 'clips/s',
                          'n_gpus': ws, 'plumbing': True, 'data': 'none (no kernel ran)',
                          'config': {'backend': dist.get_backend(), 'world_size_seen': ws, 'parallelism': 'dp%d' % ws,
                                     'max_rank_seen': int(tmax.item()),
                                     'comm': {'allreduce_bytes_per_step': red.bytes_per_step(), 'buckets': len(red.buckets),
                                              'exposed_allreduce_ms_per_step': None}}}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    a = parse()
    if a.serial:
        a.no_lookahead = True
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        launch_ranks(a)                    # never returns
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if a.gpus != world:
        raise SystemExit('bench.py --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (or run `python bench.py --gpus N` '
                         'bare and let it start the ranks)' % (a.gpus, world))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    first_step_done = start_init_watchdog(rank, a.init_timeout)
    # test hooks (tests/test_gpu_bench.py runs two ranks on the ONE GPU of a test box, where RCCL refuses duplicate
    # devices): MVF_BENCH_SHARE_GPU=1 puts every rank on device 0, MVF_BENCH_BACKEND=gloo swaps the process group.
    # Only with --test-hooks: a measurement must never change its interconnect because of a leftover variable.
    hooks = [k for k in ('MVF_BENCH_BACKEND', 'MVF_BENCH_SHARE_GPU') if k in os.environ]
    if hooks and not a.test_hooks:
        raise SystemExit('bench.py: %s set in the environment without --test-hooks: refusing to run (these hooks swap RCCL for '
                         'gloo / stack the ranks on one GPU; they are for tests/ only)' % ', '.join(hooks))
    backend = os.environ.get('MVF_BENCH_BACKEND', 'nccl') if a.test_hooks else 'nccl'
    if a.plumbing:
        if not a.test_hooks or backend != 'gloo':
            raise SystemExit('bench.py --plumbing needs --test-hooks and MVF_BENCH_BACKEND=gloo (it measures nothing)')
        return plumbing(a, world, rank, first_step_done)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (the HIP path has no CPU fallback)')
    if a.test_hooks and os.environ.get('MVF_BENCH_SHARE_GPU') == '1':
        local = 0
    stage('set_device %d' % local)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if os.environ.get('MVF_STREAMS_FIRST', '0') != '0':
        # experiment knob (tools/r6_rccl.sh): create the backbone's streams BEFORE the process group creates RCCL's, so that the
        # hardware queues are dealt in that order
        from video_rep_learning_amd import ops as _ops
        for role in os.environ['MVF_STREAMS_FIRST'].split(','):
            if role == 'dummy':
                _KEEP_STREAMS.append(torch.cuda.Stream(device=dev))
            elif role not in ('0', '1'):
                _ops.backbone_stream(role, dev)
        if os.environ['MVF_STREAMS_FIRST'] == '1':
            for role in ('side', 'lane1'):
                _ops.backbone_stream(role, dev)
    if world > 1 or os.environ.get('MVF_FORCE_REDUCER') == '1':
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        stage('init_process_group(%s) rendezvous at %s:%s' % (backend, os.environ['MASTER_ADDR'], os.environ['MASTER_PORT']))
        dist.init_process_group(backend, init_method='env://', world_size=world, rank=rank)
        # what the process group itself says, not what the environment asked for
        if dist.get_world_size() != world:
            raise SystemExit('bench.py: process group has %d ranks, WORLD_SIZE=%d' % (dist.get_world_size(), world))
    backend_seen = dist.get_backend() if dist.is_initialized() else None
    world_seen = dist.get_world_size() if dist.is_initialized() else 1
    # which physical GPU every rank really sits on (device index + PCI address as the runtime reports them): a scaling run has
    # to show N DISTINCT devices, not N ranks on one
    prop = torch.cuda.get_device_properties(local)
    me = {'rank': rank, 'device': local, 'name': prop.name,
          'pci': '%04x:%02x:%02x' % (getattr(prop, 'pci_domain_id', 0), getattr(prop, 'pci_bus_id', 0), getattr(prop, 'pci_device_id', 0)),
          'uuid': str(getattr(prop, 'uuid', ''))}
    ranks_seen = [me]
    if dist.is_initialized() and world_seen > 1:
        ranks_seen = [None] * world_seen
        dist.all_gather_object(ranks_seen, me)

    from video_rep_learning_amd import _lib
    from video_rep_learning_amd.utils import presets
    from video_rep_learning_amd.utils.optimizer import construct_optimizer
    from video_rep_learning_amd.models import build_model
    from video_rep_learning_amd.algos import get_algo
    from video_rep_learning_amd.train import DataParallelModel
    from video_rep_learning_amd.datasets import synthetic

    from video_rep_learning_amd import ops
    if a.serial:
        ops.VIT_LANE_MIN_ROWS = 1 << 62
    cfg = presets.baseline_config_2(compute_dtype=a.dtype)     # penn_mvf.yml + ViT-B/16, T=32, B=4 (dropout 0.1 kept)
    torch.manual_seed(cfg.RNG_SEED)
    stage('build_model')
    model = build_model(cfg, local).to(dev)
    from video_rep_learning_amd.utils import distributed as du
    gemm_cus = 0
    if du.collectives_active():
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
        gemm_cus = du.reserve_collective_cus(local)      # as train.main does: 8 CUs stay free for RCCL's kernels
    wrapped = DataParallelModel(model)
    opt = construct_optimizer(wrapped, cfg)
    algo = get_algo(cfg)
    loader = synthetic.SyntheticClips(cfg.TRAIN.BATCH_SIZE, cfg.TRAIN.NUM_FRAMES, cfg.IMAGE_SIZE, iters=1, seed=1234 + rank,
                                      device=dev, resident=True)
    (v0, v1), _lab, seq_lens, steps, masks, _n = next(iter(loader))
    videos = torch.stack([v0, v1], dim=1)                     # [B, 2, T, 3, H, W], resident in HBM
    seq_lens, steps, masks = seq_lens.to(dev), steps.to(dev), masks.to(dev)
    model.train()
    clip = cfg.OPTIMIZER.GRAD_CLIP

    def step(lookahead=not a.no_lookahead):
        # one-batch lookahead as in train.train(): the frozen-backbone forward of the NEXT batch is started on the side
        # stream before this batch's head work is enqueued; the head consumes the forward launched one step EARLIER (the
        # stash is FIFO and was primed below).  Every step still launches exactly one backbone forward and runs one
        # head forward/backward/update; the resident synthetic batch is the same tensor each step.
        if lookahead:
            wrapped.prefetch(videos)
        opt.zero_grad()
        loss = algo.compute_loss(wrapped, videos, seq_lens, steps, masks)['loss']
        ops.backward(loss)                 # as train.train does: loss.backward() with a cached seed gradient
        opt.step(max_norm=clip)
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    stage('first step (kernel library load, first collectives)')
    if not a.no_lookahead:
        wrapped.prefetch(videos)     # prime the pipeline: step k's head reads the forward launched in step k-1
    for i in range(a.warmup):
        step()
        if i == 0:
            torch.cuda.synchronize()
            first_step_done.set()
            stage('warm-up')
    fence()
    first_step_done.set()
    stage('timed region')
    reducer = getattr(opt, 'reducer', None)
    if reducer is not None and reducer.active:
        reducer.exposed, reducer.timing = [], True
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    comm = None
    if world > 1:      # always present with more than one rank, also when no collective ran (then it says so)
        comm = {'allreduce_bytes_per_step': 0, 'buckets': 0, 'exposed_allreduce_ms_per_step': None, 'reducer_active': False}
    if reducer is not None and reducer.active:
        reducer.timing = False
        # the gradient all-reduce as the process group ran it: payload per step and the time the compute stream spent waiting
        # for it between the last backward kernel and the optimizer (device events; what backward did not hide)
        comm = {'allreduce_bytes_per_step': reducer.bytes_per_step(), 'buckets': len(reducer.buckets),
                'exposed_allreduce_ms_per_step': None if (ex := reducer.exposed_ms()) is None else round(ex, 4), 'reducer_active': True}
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()
    # ---- the same loop, extended to a settled state: every rank derives the same step count from the all-reduced time ----
    sustained = None
    if a.min_sustain_s > 0 and dt < a.min_sustain_s:
        stage('sustained region')
        n2 = max(a.steps, min(20000, int((a.min_sustain_s - dt) / (dt / a.steps)) + 1))
        t0 = time.perf_counter()
        for _ in range(n2):
            loss = step()
        fence()
        dt2 = time.perf_counter() - t0
        t2 = torch.tensor([dt2], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        sustained = {'steps': n2, 'seconds': round(t2.item(), 4), 'ms_per_step': round(t2.item() / n2 * 1e3, 3)}
    stage('roofline steps')
    last_loss = float(loss.item())

    # ---- roofline of the dominant kernel: per-launch HIP-event timing during extra (untimed) steps ----
    roof = None
    # Kernel timing wants the kernels one at a time: the timed region above keeps two backbone lanes and the head in
    # flight together (a launch's event-to-event time would then include the other streams' kernels), so these extra
    # steps run the backbone as one lane, without lookahead -- the same serial order rocprofv3 records.  EVERY rank runs
    # them (a step contains the gradient all-reduce and the SyncBN exchanges); only rank 0 records and reports.
    lane_rows, ops.VIT_LANE_MIN_ROWS = ops.VIT_LANE_MIN_ROWS, 1 << 62
    step(lookahead=False)                  # drains the primed forward
    torch.cuda.synchronize()
    if rank == 0:
        _lib.call('mvf_prof_enable', 1)
    for _ in range(max(a.profile_steps, 1)):
        step(lookahead=False)
    torch.cuda.synchronize()
    ops.VIT_LANE_MIN_ROWS = lane_rows
    if rank == 0:
        _lib.call('mvf_prof_enable', 0)
        G = 16
        ms, fl = (ctypes.c_double * G)(), (ctypes.c_double * G)()
        cnt, epi, nn, kk = (ctypes.c_int * G)(), (ctypes.c_int * G)(), (ctypes.c_int * G)(), (ctypes.c_int * G)()
        ng = ctypes.c_int(0)
        _lib.call('mvf_prof_collect', ms, fl, cnt, epi, nn, kk, G, ctypes.byref(ng))
        groups = []
        for g in range(ng.value):
            name = GEMM_NAMES.get((epi[g], nn[g], kk[g]), 'gemm epi%d N=%d K=%d' % (epi[g], nn[g], kk[g]))
            groups.append({'name': name, 'epi': epi[g], 'n': nn[g], 'k': kk[g], 'launches': cnt[g], 'ms': ms[g], 'flop': fl[g],
                           'avg_us': round(ms[g] * 1e3 / max(cnt[g], 1), 1),
                           'tflops': round(fl[g] / (ms[g] * 1e-3) / 1e12, 1) if ms[g] > 0 else 0.0})
        dom = max(groups, key=lambda r: r['ms'])       # the GEMM shape the step spends most time in
        peak = PEAK_BF16_TFLOPS if a.dtype in ('bf16', 'fp16') else PEAK_F32_TFLOPS     # (fp16 MFMA: the bf16 rate)
        ach = dom['flop'] / (dom['ms'] * 1e-3) / 1e12
        tot_ms, tot_fl = sum(r['ms'] for r in groups), sum(r['flop'] for r in groups)
        kern = ('vit_qkv_attn_kernel' if dom['epi'] == 5 else 'gemm_tc256_kernel' if a.dtype in ('bf16', 'fp16') else 'gemm_tc_kernel<float>') + \
            ' / ' + dom['name']
        roof = {'bound': 'mfma', 'achieved': round(ach, 1), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
                'traffic': pmc_traffic(dom['name']), 'kernel': kern, 'launches': dom['launches'],
                'timing': 'HIP events around each launch on its stream, kernels serialized (1 backbone lane, no lookahead)',
                'avg_launch_us': dom['avg_us'], 'flop_per_launch': dom['flop'] / max(dom['launches'], 1),
                **({'sustained_loop_rate': {'tflops': SUSTAINED_BF16_LOOP_TFLOPS, 'source': 'profiles/r03/power_probe.txt',
                                            'all_gemm_frac_of_it': round(tot_fl / (tot_ms * 1e-3) / 1e12 / SUSTAINED_BF16_LOOP_TFLOPS, 4)}}
                   if a.dtype in ('bf16', 'fp16') else {}),
                'all_gemm': {'achieved': round(tot_fl / (tot_ms * 1e-3) / 1e12, 1), 'frac': round(tot_fl / (tot_ms * 1e-3) / 1e12 / peak, 4),
                             'ms_per_step': round(tot_ms / max(a.profile_steps, 1), 3)},
                'by_kernel': {r['name']: {'launches': r['launches'], 'avg_us': r['avg_us'], 'tflops': r['tflops']}
                              for r in groups},
                # the same launches under the names rocprofv3 prints: one kernel per epilogue, so proj and fc2 (and
                # nothing else) share `gemm_tc256_kernel<2, false>`; compare with `bench.py --serial` under rocprofv3
                'rocprof_kernels': rocprof_names(groups, a.dtype, ops.VIT_LN_FOLD, os.environ.get('MVF_PROJ_DEFER', '1') != '0', nwg=gemm_wgs()),
                'ln_fold': {0: 'off', 1: 'norm1 (blocks 1..) and norm2 folded into qkv / fc1', 2: 'norm1 of blocks 1.. folded into the qkv GEMM (default)', 3: 'norm2 folded into fc1'}[ops.VIT_LN_FOLD]}
    if world > 1:
        dist.barrier()

    if rank == 0:
        clips = world_seen * cfg.TRAIN.BATCH_SIZE * 2 * a.steps
        value = clips / dt
        out = {
            'metric': 'video-clips/sec/node, ViT-B/16 32-frame MV-Former', 'value': round(value, 2), 'unit': 'clips/s',
            'n_gpus': world_seen, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(dt / a.steps * 1e3, 3),
            'timed_region_s': round(dt, 4),
            'ms_per_step_sustained': None if sustained is None else sustained['ms_per_step'], 'sustained_region': sustained,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: PennAction MV-Former (penn_mvf.yml + ViT-B/16), 32 frames, '
                                   'batch 4/GPU = 8 clips/GPU/step, full train step (frozen backbone fwd, head fwd+bwd, SCL, '
                                   'grad all-reduce, clip+Adam), dropout 0.1', 'global_batch': 4 * world_seen, 'frames': 32,
                       'parallelism': 'dp%d' % world_seen, 'frames_per_sec': round(value * 32, 1), 'samples_per_sec': round(value / 2, 2),
                       # algorithmic work (1.1925 TFLOP per clip = 9.54 TFLOP per step and GPU) over the measured time
                       'tflops_algorithmic_per_gpu': round(value / world_seen * TFLOP_PER_CLIP, 1),
                       'tflop_per_step_per_gpu': round(cfg.TRAIN.BATCH_SIZE * 2 * TFLOP_PER_CLIP, 2),
                       'head_dtype': model.head_dtype, 'last_loss': round(last_loss, 4),
                       'gemm_cu_budget': gemm_cus or 'all',
                       # hardware queues the runtime deals this process's streams over (read when it loaded; 8 asked for under data
                       # parallel unless the environment says otherwise: MVF_HW_QUEUES=0 keeps the runtime's default)
                       'gpu_max_hw_queues': os.environ.get('GPU_MAX_HW_QUEUES', 'runtime default (4)'),
                       # the process group as it actually ran (None / 1: no group): never the environment's word for it
                       'backend': backend_seen, 'world_size_seen': world_seen, 'ranks': ranks_seen,
                       'distinct_gpus': len({(r['pci'], r['uuid']) for r in ranks_seen}),
                       **({'comm': comm} if comm is not None else {})},
            'roofline': roof,
        }
        if roof is not None:
            # whole-step figures beside the dominant kernel's: algorithmic TFLOP of a step over the step time over the peak, for the
            # K requested steps and for the settled extension of the same loop
            tf_step = cfg.TRAIN.BATCH_SIZE * 2 * TFLOP_PER_CLIP
            peak_step = PEAK_BF16_TFLOPS if a.dtype in ('bf16', 'fp16') else PEAK_F32_TFLOPS
            roof['step_frac'] = round(tf_step / (dt / a.steps) / peak_step, 4)
            roof['sustained_frac'] = None if sustained is None else round(tf_step / (sustained['ms_per_step'] * 1e-3) / peak_step, 4)
            roof['frac_note'] = ('frac: the dominant kernel alone, HIP events in one-kernel-at-a-time steps right after the %s; step_frac / '
                                 'sustained_frac: %.2f algorithmic TFLOP per step over ms_per_step / ms_per_step_sustained'
                                 % ('settled extension' if sustained else 'timed region', tf_step))
        parity_ok = True
        if world == 1 and not a.no_cpu_baseline:
            try:
                out['cpu_baseline'] = cpu_baseline(cfg, model)
            except Exception as e:  # the baseline must never sink the measurement
                out['cpu_baseline'] = {'value': None, 'unit': 'clips/s', 'cores': os.cpu_count(), 'kind': 'port',
                                       'sample': 'failed: %r' % (e,)}
            out['parity'] = parity_block(model, (videos, seq_lens, steps, masks), max(1, min(a.parity_videos, videos.shape[0])), dev,
                                         reduced='fp16' if a.dtype == 'fp16' else 'bf16')
            parity_ok = out['parity']['ok']
        print(json.dumps(out), flush=True)
        if not parity_ok:
            print('bench.py: parity gate FAILED: %r' % (out['parity'],), file=sys.stderr, flush=True)
            raise SystemExit(3)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
