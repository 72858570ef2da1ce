"""TransformerModel: frozen ViT backbone -> MV-Former head -> (projection) -> L2 normalise.

Same constructor contract and forward signature as CARL_MVF/models/transformer.py:16-244 for the path the
MV-Former configs take (timm ViT backbone, fully frozen, FUSION_TYPE 'smart'); `cfg.MODEL.BASE_MODEL.OUT_CHANNEL`
is set/multiplied exactly as transformer.py:43-54,90 does, state-dict keys are `backbone.model.*`, `embed.*`,
`ssl_projection.*`.  The backbone runs as one C-ABI call (mvf_vit_fwd) per forward instead of per
FRAMES_PER_BATCH chunk of T (per-frame independent, so the numbers are the same) and hands its tapped blocks to
the head without the hook/concat/CLS-drop/movedim copies of transformer.py:199-218,322-331.
Also built: late fusion (TransformerEmbModel on the CLS embedding or on max/avg-pooled spatial tokens), the partially frozen
backbone (ViTFrontEnd / ViTBackEnd, 0 <= LAYER < depth, models/vit.py), MODEL.CLS_RES, TRAIN.BACKBONE_WARMUP.
Not on this path (raise with a clear message): ResNet-50 backbones, the classification head, the SwiGLU DINOv2-giant."""
import torch
import torch.nn as nn

from .. import ops
from . import vit as vitlib
from .mvformer import MultiEntityTransformerEmbModel, Taps
from .utils import Encoder, PositionalEncoder
from .resnet_c2d import MLPHead


class FeatureExtractor(nn.Module):
    """Holder that keeps the reference's key prefix `backbone.model.*` (transformer.py:306-333).  `layers` are
    timm-style names ('blocks.3'); the taps come straight out of the HIP forward instead of forward hooks."""

    def __init__(self, model, layers, return_output=True):
        super().__init__()
        self.model = model
        self.layers = list(layers)
        self.tap_ids = tuple(int(l.split('.')[-1]) for l in self.layers)
        self.return_output = return_output

    def forward(self, x, dtype='bf16', frames_per_chunk=0):
        if isinstance(self.model, vitlib.ViTBackEnd):        # x = the front end's residual stream [F, N, D]
            return self.model(x, self.tap_ids, fast=(dtype == 'bf16' and x.shape[-1] % 128 == 0))
        return self.model.forward_taps(x, self.tap_ids, dtype=dtype, frames_per_chunk=frames_per_chunk)


class TransformerEmbModel(nn.Module):
    """Late fusion (models/transformer.py:248-300): one feature vector per frame (CLS embedding, or the spatial tokens
    max/avg-pooled by ops.token_pool) -> [Dropout, Linear, BatchNorm1d, ReLU] x k -> video_emb (+ sin/cos table in the GEMM
    epilogue) -> temporal encoder over the T frames -> embedding_layer.  Parameter names as the reference's."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        em = cfg.MODEL.EMBEDDER_MODEL
        drop_rate = em.FC_DROPOUT_RATE
        in_channels = cfg.MODEL.BASE_MODEL.OUT_CHANNEL
        self.embedding_size = em.EMBEDDING_SIZE
        hidden_channels = em.HIDDEN_SIZE
        assert em.FLATTEN_METHOD in ['max_pool', 'avg_pool']
        self.flatten_method = em.FLATTEN_METHOD
        self.pooling = nn.Identity()              # the pooling itself is ops.token_pool (parameter-free)
        layers = []
        for channels, _activate in em.FC_LAYERS:
            channels = channels * em.CAPACITY_SCALAR
            layers += [nn.Dropout(drop_rate), nn.Linear(in_channels, channels), nn.BatchNorm1d(channels), nn.ReLU(True)]
            in_channels = channels
        self.fc_layers = nn.Sequential(*layers)
        self.video_emb = nn.Linear(in_channels, hidden_channels)
        self.video_pos_enc = PositionalEncoder(cfg, hidden_channels, drop_rate, seq_len=cfg.TRAIN.NUM_FRAMES)
        if em.NUM_LAYERS > 0:
            self.video_encoder = Encoder(hidden_channels, drop_rate, em.NUM_HEADS, em.D_FF, em.NUM_LAYERS)
            self.video_encoder.head_dtype = ops.head_dtype_of(cfg)
        self.embedding_layer = nn.Linear(hidden_channels, self.embedding_size)
        self.drop_state = ops.DropoutState(seed=int(cfg.RNG_SEED) if 'RNG_SEED' in cfg else 0)
        self.sync_group = None

    def set_warmup_status(self, new_status):      # train.py:79 calls it on every embed model
        pass

    def forward(self, x, n_clips, n_frames, video_masks=None):
        """x [Bc*T, C] -> [Bc, T, E]."""
        mods = list(self.fc_layers)
        for i in range(0, len(mods), 4):
            x = ops.dropout_add(x, None, mods[i].p, self.training, self.drop_state)
            bn = mods[i + 2]
            if bn.training != self.training:
                bn.train(self.training)
            x = ops.batch_norm(ops.linear(x, mods[i + 1].weight, mods[i + 1].bias), bn.weight, bn.bias, bn.running_mean,
                               bn.running_var, self.training and bn.training, momentum=bn.momentum, eps=bn.eps, relu=True,
                               sync=isinstance(bn, nn.SyncBatchNorm), group=self.sync_group)
        x = ops.linear(x, self.video_emb.weight, self.video_emb.bias, table=self.video_pos_enc.table(n_frames, x.device),
                       tab_div=1, tab_mod=n_frames)
        x = ops.dropout_add(x, None, self.video_pos_enc.dout_p, self.training, self.drop_state)
        x = x.view(n_clips, n_frames, -1)
        if self.cfg.MODEL.EMBEDDER_MODEL.NUM_LAYERS > 0:
            x = self.video_encoder(x, src_mask=video_masks, drop_state=self.drop_state)
        x = ops.linear(x.reshape(n_clips * n_frames, -1), self.embedding_layer.weight, self.embedding_layer.bias)
        return x.view(n_clips, n_frames, self.embedding_size)


class TransformerModel(nn.Module):
    def __init__(self, cfg, local_rank=None):
        super().__init__()
        self.cfg = cfg
        em = cfg.MODEL.EMBEDDER_MODEL
        self.fusion_type = em.FUSION_TYPE if 'FUSION_TYPE' in em else 'late'
        self.use_cls_res = bool('CLS_RES' in cfg.MODEL and cfg.MODEL.CLS_RES)
        net = cfg.MODEL.BASE_MODEL.NETWORK
        if 'TIMM-' not in net:
            raise NotImplementedError('only TIMM-* ViT backbones are on the MI355X path (got %s); the ResNet-50 '
                                      'CARL baselines are out of scope' % net)
        if self.fusion_type not in ('smart', 'late'):
            print('WARNING: invalid setting for cfg.MODEL.EMBEDDER_MODEL.FUSION_TYPE:')
            print(self.fusion_type)
            exit(-1)
        if self.use_cls_res and self.fusion_type == 'late':
            print('ERROR: CLS_RES cannot be used with late fusion')
            exit(-1)
        self.backbone_type = 'timm'
        # late fusion reads either the CLS embedding or the spatial tokens (transformer.py:66-70)
        self.late_type = em.LATE_TYPE if 'LATE_TYPE' in em else 'cls'
        assert self.late_type in ['cls', 'spatial']
        name = net[5:]
        if name in vitlib.VIT_UNSUPPORTED:
            raise NotImplementedError('TIMM model %s is not supported on the MI355X path: %s' % (name, vitlib.VIT_UNSUPPORTED[name]))
        if name not in vitlib.VIT_ZOO:
            print('ERROR: unknown/unsupported TIMM model:')
            print(name)
            exit()
        dim, blk_count = vitlib.VIT_ZOO[name][0], vitlib.VIT_ZOO[name][1]
        cfg.MODEL.BASE_MODEL.OUT_CHANNEL = dim
        weights = cfg.MODEL.BASE_MODEL.WEIGHTS if 'WEIGHTS' in cfg.MODEL.BASE_MODEL else None
        model = vitlib.create_model(name, pretrained=False, weights=weights, img_size=cfg.IMAGE_SIZE)
        if self.use_cls_res:
            self.cls_res_res = nn.Linear(dim, em.EMBEDDING_SIZE)
        self.uses_taps = self.fusion_type != 'late' or self.late_type == 'spatial'
        if not self.uses_taps:
            extract_ids = []                      # late fusion on the CLS embedding: no block is tapped
        elif 'SMART_FEATS' not in em:
            extract_ids = ['blocks.11']
        else:
            extract_ids = ['blocks.%s' % t for t in str(em.SMART_FEATS).split(',')]
            cfg.MODEL.BASE_MODEL.OUT_CHANNEL *= len(extract_ids)
        layer = cfg.MODEL.BASE_MODEL.LAYER
        self.split_layer = None
        if layer < 0 or layer >= blk_count:       # fully frozen (transformer.py:93-99)
            self.backbone = FeatureExtractor(model, extract_ids)
            self.res_finetune = nn.Identity()
        else:                                     # frozen front end + trainable back end (transformer.py:100-116)
            new_ids = []
            for e_id in extract_ids:
                b_idx = int(e_id.split('.')[-1]) - layer
                if b_idx < 0:
                    print('ERROR: cannot request extract of %s as it is not in ViTBackEnd wrapper' % e_id)
                    exit(-1)
                new_ids.append('blocks.%i' % b_idx)
            self.split_layer = layer
            self.backbone = vitlib.ViTFrontEnd(model, layer)
            self.res_finetune = FeatureExtractor(vitlib.ViTBackEnd(model, layer), new_ids)
        for p in self.backbone.parameters():      # frozen: never in the optimizer, never all-reduced
            p.requires_grad_(False)
        self.embed = MultiEntityTransformerEmbModel(cfg) if self.fusion_type == 'smart' else TransformerEmbModel(cfg)
        # FUSION_CLS / CLS_GRAD_ONLY: validated and announced by the reference's constructor (transformer.py:144-163) and
        # read nowhere else -- kept as the same two flags
        self.fuse_cls = bool('FUSION_CLS' in em and em.FUSION_CLS is True)
        if self.fuse_cls:
            print('FUSION_CLS enabled, cls token will be included with spatial tokens')
        self.cls_grad_only = bool('CLS_GRAD_ONLY' in em and em.CLS_GRAD_ONLY is True)
        if self.cls_grad_only:
            if not self.fuse_cls:
                print('WARNING: Invalid config')
                print('CLS_GRAD_ONLY can only be used with FUSION_CLS enabled')
                exit(-1)
            print('CLS_GRAD_ONLY enabled, gradients will only pass to the backbone through the CLS token')
        self.embedding_size = self.embed.embedding_size
        if cfg.MODEL.PROJECTION:
            self.ssl_projection = MLPHead(cfg)
        if cfg.TRAINING_ALGO == 'classification':
            raise NotImplementedError('classification algo is out of scope (configs_mvf/* all use scl)')
        mi = cfg.MI355X if 'MI355X' in cfg else {}
        self.compute_dtype = mi['COMPUTE_DTYPE'] if 'COMPUTE_DTYPE' in mi else \
            ('bf16' if ('USE_AMP' in cfg and cfg.USE_AMP) else 'fp32')
        self.frames_per_chunk = int(mi['FRAMES_PER_CHUNK']) if 'FRAMES_PER_CHUNK' in mi else 0
        self.head_dtype = ops.head_dtype_of(cfg)
        # one pack of bf16 weight images for every row-chain module of the head: refreshed by one launch per optimizer step
        shared = ops.HeadPack()
        for m in self.modules():
            for attr in ('_pack', '_pack_trunk', '_pack_tail'):
                if isinstance(getattr(m, attr, None), ops.HeadPack):
                    setattr(m, attr, shared)
        if self.compute_dtype in ('fp16', 'f16') and getattr(self, 'split_layer', None) is not None:
            raise NotImplementedError('MI355X.COMPUTE_DTYPE fp16 covers the FROZEN backbone (MODEL.BASE_MODEL.LAYER >= depth); the '
                                      'trainable back-end blocks of a partially frozen one run in bf16 or fp32')

    def train(self, mode=True):
        super().train(mode)
        self.backbone.eval()       # transformer.py:186: the frozen backbone always runs in eval()
        return self

    def set_head_dtype(self, dt):
        """'bf16' / 'fp16': the trainable head's Linears on the 16-bit matrix cores (row-chain kernels, csrc/head_chain.hip) where their
        shapes allow it -- 'fp16': IEEE fp16 operands in the forward GEMMs, bf16 in the gradient GEMMs --; 'fp32': the fp32 kernels.
        (The frozen backbone's dtype is `compute_dtype`.)"""
        assert dt in ('bf16', 'fp16', 'fp32'), dt
        self.head_dtype = dt
        for m in self.modules():
            if m is not self and hasattr(m, 'head_dtype'):
                m.head_dtype = dt

    def head_bf16_linears(self):
        """Name prefixes (relative to `embed.`) of the Linears that run with 16-bit operands in the current mode (bf16, or with
        HEAD_DTYPE fp16: fp16 in the forward and bf16 in the two gradient products) -- what an emulating checker has to round
        (oracle/head.py emulating)."""
        pre = []
        enc = getattr(self.embed, 'video_encoder', None)
        if enc is not None and enc.chain_active():
            pre.append('video_encoder.')
        if getattr(self.embed, 'trunk_chain_active', None) is not None and self.embed.trunk_chain_active():
            pre += ['fc_layers.%d' % (4 * i + 1) for i in range(len(list(self.embed.fc_layers)) // 4)] + ['video_emb']
        if getattr(self.embed, 'tail_chain_active', None) is not None and self.embed.tail_chain_active():
            pre.append('embedding_layer')
        if self.cfg.MODEL.PROJECTION and self.ssl_projection.chain_active():
            pre += ['ssl_projection.net.0', 'ssl_projection.net.3', 'net.0', 'net.3']
        return tuple(pre)

    # ---- backbone pipeline: the frozen ViT runs on its own HIP stream, optionally one batch ahead of the head ----
    # The backbone has no trainable state, so the ViT forward of batch i+1 does not depend on the optimizer step of
    # batch i: `prefetch(x_next)` enqueues it on the side stream while the (latency-bound, mostly-idle-CU) head
    # forward/backward of batch i runs on the caller's stream.  `forward(x)` consumes the oldest prefetched result
    # for the same tensor, or launches the backbone itself when there is none -- the numbers are identical either way.
    @staticmethod
    def _key(x):
        return (x.data_ptr(), tuple(x.shape), x._version)

    def _launch_backbone(self, x, ready_event=None):
        bc, t, c, h, w = x.shape
        if not x.is_cuda:          # the first kernel call of a step: fail loudly, there is no CPU compute path
            raise ops._lib.MvfError('the MV-Former forward received a %s tensor: its kernels only run on a gfx950 device '
                                    '(no CPU fallback)' % x.device)
        ops.gemm_spare_mode(self.training and torch.is_grad_enabled())
        cur = torch.cuda.current_stream(x.device)
        if getattr(self, '_side', None) is None or self._side.device != x.device:
            self._side = ops.backbone_stream('side', x.device)      # one per process and device (ops.backbone_stream: hardware queues)
        if ready_event is None:            # "x is ready": everything enqueued on the caller's stream so far
            ready_event = torch.cuda.Event()
            ready_event.record(cur)
        self._side.wait_event(ready_event)
        with torch.cuda.stream(self._side):
            out = self.backbone(x.reshape(bc * t, c, h, w), dtype=self.compute_dtype, frames_per_chunk=self.frames_per_chunk)
            # fully frozen: (taps, cls); partially frozen: the fp32 residual stream the trainable blocks start from
            taps, cls = out if self.split_layer is None else ((out,), None)
            done = torch.cuda.Event()
            done.record(self._side)
        x.record_stream(self._side)
        return taps, cls, done

    def prefetch(self, x, ready_event=None):
        """Enqueue the backbone forward of x [Bc, T, 3, H, W] (the NEXT batch) on the side stream.  Call it right after
        x has been produced and BEFORE enqueuing the current batch's head work, or pass the event that marks x ready."""
        if not x.is_cuda:
            raise ops._lib.MvfError('prefetch received a %s tensor (no CPU fallback)' % x.device)
        if not hasattr(self, '_stash'):
            self._stash = []
        self._stash.append((self._key(x),) + self._launch_backbone(x, ready_event))

    def features(self, x):
        """[Bc, T, 3, H, W] -> (Taps, cls_emb [Bc*T, D])."""
        bc, t, c, h, w = x.shape
        key, hit = self._key(x), None
        for i, e in enumerate(getattr(self, '_stash', [])):
            if e[0] == key:
                hit = self._stash.pop(i)
                break
        taps, cls, done = hit[1:] if hit is not None else self._launch_backbone(x)
        cur = torch.cuda.current_stream(x.device)
        cur.wait_event(done)
        for tns in list(taps) + ([cls] if cls is not None else []):
            tns.record_stream(cur)
        if self.split_layer is not None:          # trainable blocks: on the caller's stream, recorded by autograd
            taps, cls = self.res_finetune(taps[0], dtype=self.compute_dtype)
        ntok = (h // self.backbone.model.patch_size) * (w // self.backbone.model.patch_size)
        return Taps(taps, bc, t, ntok), cls

    def forward(self, x, num_frames=None, video_masks=None, project=False, classification=False):
        if classification:
            raise NotImplementedError('classification head is out of scope')
        if video_masks is not None:
            video_masks = video_masks.to(x.device)     # DDP's input scatter did this in the reference
        feats, cls_emb = self.features(x)
        if self.fusion_type == 'smart':
            x = self.embed(feats, video_masks=video_masks, cls_emb=cls_emb)
        elif self.late_type == 'cls':             # transformer.py:192-196: the CLS embedding as a 1 x 1 feature map
            x = self.embed(cls_emb, feats.n_clips, feats.n_frames, video_masks=video_masks)
        else:
            pooled = ops.token_pool(feats.tensors, feats.n_clips * feats.n_frames, feats.n_tokens, self.embed.flatten_method)
            x = self.embed(pooled, feats.n_clips, feats.n_frames, video_masks=video_masks)
        if self.cfg.MODEL.PROJECTION and project:
            x = self.ssl_projection(x, normalize=True)
        elif self.cfg.MODEL.L2_NORMALIZE:
            x = ops.l2_normalize(x)
        if self.use_cls_res:
            cls_res = ops.linear(cls_emb, self.cls_res_res.weight, self.cls_res_res.bias).view(x.shape[0], x.shape[1], -1)
            if self.cfg.MODEL.L2_NORMALIZE:
                cls_res = ops.l2_normalize(cls_res)
            x = x + cls_res
            if self.cfg.MODEL.L2_NORMALIZE:
                x = ops.l2_normalize(x)
        return x
