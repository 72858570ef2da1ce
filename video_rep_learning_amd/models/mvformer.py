"""MV-Former head: learned-query multi-entity pooling + per-entity FC stack + joint temporal transformer.

Same classes, constructor logic, optional-key probing and parameter names as CARL_MVF/models/mvformer.py
(MultiEntityTransformerEmbModel :15-200, LearnableTokenPooling :207-266, LSTPCrossAtt :275-414) -- reference
checkpoints load with load_state_dict -- but the forward is a chain of HIP ops and three things are laid out
differently on purpose:
  * input: the tapped backbone blocks arrive as separate token-major buffers [F*N, D] (`Taps`), not as one
    channel-concatenated NCHW tensor that is then movedim'ed back (transformer.py:199-218, mvformer.py:244-246);
  * pooling uses the streaming rewrite documented in csrc/lstp_pool.hip (no K/V projection of all tokens), all
    clips in one launch instead of the Python loop of mvformer.py:255-264;
  * rows are kept in (clip, entity, frame) order from the pooling output on, which is the order the temporal
    encoder consumes, so the movedim/reshape copies of mvformer.py:154-159 disappear.
"""
import math

import torch
import torch.nn as nn

from .. import ops
from .utils import PositionalEncoder, Encoder


class Taps:
    """Backbone features for the head: `tensors[j]` is block taps[j]'s output, [F*N, D], CLS dropped."""

    def __init__(self, tensors, n_clips, n_frames, n_tokens):
        self.tensors = list(tensors)
        self.n_clips, self.n_frames, self.n_tokens = n_clips, n_frames, n_tokens

    @staticmethod
    def from_nchw(x, n_taps):
        """The reference's [B, T, C, h, w] layout (C = n_taps * D) -> Taps (copies; API-compat path only)."""
        b, t, c, h, w = x.shape
        d = c // n_taps
        toks = x.reshape(b * t, c, h * w).transpose(1, 2)   # [F, N, C]
        return Taps([toks[:, :, j * d:(j + 1) * d].reshape(b * t * h * w, d).contiguous() for j in range(n_taps)],
                    b, t, h * w)


def _em(cfg, key, default):
    return cfg.MODEL.EMBEDDER_MODEL[key] if key in cfg.MODEL.EMBEDDER_MODEL else default


class LSTPCrossAtt(nn.Module):
    def __init__(self, cfg, num_static, num_dynamic, d_model_K, d_model_V, d_model, d_dyn_in=None, dout_p=0.0):
        super().__init__()
        self.cfg = cfg
        self.d_model_K, self.d_model_V, self.d_model = d_model_K, d_model_V, d_model
        self.pass_through = bool(_em(cfg, 'VAL_PASS', False))
        self.disjoint_att = bool(_em(cfg, 'SMART_DISJOINT', False))
        self.ln_keys = bool(_em(cfg, 'SMART_LN_KEYS', False))
        self.dyn_ctrl = _em(cfg, 'DYNAMIC_CTRL', 'separate')
        assert self.dyn_ctrl in ['separate', 'first', 'average']
        if num_static == 0 and num_dynamic == 0:
            print('ERROR: cannot have both num_static == 0 and num_dynamic == 0')
            exit(-1)
        self.linear_K2d = nn.Linear(d_model_K, d_model)
        self.linear_V2d = nn.Identity() if self.pass_through else nn.Linear(d_model_V, d_model)
        self.num_s, self.stat = num_static, num_static > 0
        if self.stat:
            self.Q_s = nn.Parameter(torch.empty([1, num_static, d_model], dtype=torch.float32))
            nn.init.kaiming_uniform_(self.Q_s, a=math.sqrt(5))
            self.Q_s_b = nn.Parameter(torch.empty(d_model, dtype=torch.float32))
            fan_in, _ = nn.init._calculate_fan_in_and_fan_out(self.Q_s)
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            nn.init.uniform_(self.Q_s_b, -bound, bound)
        self.num_d, self.dyn = num_dynamic, num_dynamic > 0
        if self.dyn:
            self.d_dyn_in = d_dyn_in if d_dyn_in is not None else d_model_V
            self.in2dynQ = nn.Linear(self.d_dyn_in, d_model * num_dynamic)
        self.visual = True
        self.attn_holder = nn.Identity()
        self.attn_matrix = None

    def query_vectors(self, dyn_in, n_clips, n_frames, project=True):
        """wq = q W_K: [nq, C] for static-only queries, else [Bc, nq, T, C] (one query set per frame); project=False: the
        queries themselves ([nq, d] / [Bc, nq, T, d])."""
        wk = self.linear_K2d.weight
        if not self.dyn and project and torch.is_grad_enabled():
            return ops.static_query(self.Q_s, self.Q_s_b, wk)                 # (Q_s + Q_s_b) W_K, gradients written in place
        qs = (self.Q_s + self.Q_s_b)[0] if self.stat else None            # [nst, d]
        if not self.dyn:
            return ops.matmul(qs, wk) if project else qs
        assert dyn_in is not None
        d = dyn_in.view(n_clips, n_frames, -1)
        if self.dyn_ctrl == 'first':
            d = d[:, :1]
        elif self.dyn_ctrl == 'average':
            d = d.mean(1, keepdim=True)
        qd = ops.linear(d.reshape(-1, d.shape[-1]), self.in2dynQ.weight, self.in2dynQ.bias)
        qd = qd.view(n_clips, -1, self.num_d, self.d_model).expand(n_clips, n_frames, self.num_d, self.d_model)
        if self.stat:
            qd = torch.cat([qs.view(1, 1, self.num_s, self.d_model).expand(n_clips, n_frames, -1, -1), qd], 2)
        q = qd.permute(0, 2, 1, 3).reshape(-1, self.d_model)                # rows (clip, query, frame)
        if not project:
            return q.view(n_clips, -1, n_frames, self.d_model)
        return ops.matmul(q, wk).view(n_clips, -1, n_frames, wk.shape[1])

    def forward(self, taps, dyn_in=None):
        """-> [Bc, nq, T, d_out] rows in (clip, entity, frame) order."""
        nq = self.num_s + self.num_d
        F = taps.n_clips * taps.n_frames
        holder = {}
        if self.ln_keys:
            # SMART_LN_KEYS (mvformer.py:399-400): scores against F.normalize(K).  The normalisation depends on the token,
            # so the keys ARE projected here (one [F*N, C] x [C, d] GEMM on an fp32 copy of the taps) -- the ablation's cost
            xcat = torch.cat([t.float() for t in taps.tensors], 1)
            kn = ops.l2_normalize(ops.linear(xcat, self.linear_K2d.weight, self.linear_K2d.bias))
            q = self.query_vectors(dyn_in, taps.n_clips, taps.n_frames, project=False)
            if self.dyn:       # one query set per frame (mvformer.py:365-396): scores[f, n, j] = kn[f, n] . q[f, j]
                scores = ops.frame_scores(kn, q, F, taps.n_tokens, taps.n_frames, nq)
            else:
                scores = ops.linear(kn, q, None)                                   # [F*N, nq]
            pooled, rowsum = ops.lstp_pool_from_scores(scores, taps.tensors, F, taps.n_tokens, taps.n_frames, nq,
                                                       self.d_model, disjoint=self.disjoint_att, holder=holder)
        else:
            vec = self.query_vectors(dyn_in, taps.n_clips, taps.n_frames)
            pooled, rowsum = ops.lstp_pool(vec, taps.tensors, F, taps.n_tokens, taps.n_frames, nq, self.d_model,
                                           disjoint=self.disjoint_att, holder=holder)
        if self.visual:
            self.attn_matrix = holder['attn'].detach()                    # [F, nq, N]
            _ = self.attn_holder(self.attn_matrix)
        if self.pass_through:
            return pooled
        wv, bv = self.linear_V2d.weight, self.linear_V2d.bias
        if not self.disjoint_att:
            return ops.linear(pooled, wv, bv)                             # softmax rows sum to 1: plain bias
        return ops.linear(pooled, wv, None) + rowsum.unsqueeze(-1) * bv


class LearnableTokenPooling(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.nst = _em(cfg, 'SMART_TOKENS', 5)
        self.nsdt = _em(cfg, 'SMART_DYNAMIC_TOKENS', 0)
        self.spc = _em(cfg, 'SMART_POOL_CHANNELS', 384)
        self.in_c = cfg.MODEL.BASE_MODEL.OUT_CHANNEL
        d_dyn_in = self.in_c
        if 'SMART_FEATS' in cfg.MODEL.EMBEDDER_MODEL:
            sfl = str(cfg.MODEL.EMBEDDER_MODEL.SMART_FEATS)
            if ',' in sfl:
                d_dyn_in = int(d_dyn_in / len(sfl.split(',')))
        self.cross_att = LSTPCrossAtt(cfg=cfg, num_static=self.nst, num_dynamic=self.nsdt, d_model_K=self.in_c,
                                      d_model_V=self.in_c, d_model=self.spc, d_dyn_in=d_dyn_in)

    def forward(self, taps, dyn_in=None):
        return self.cross_att(taps, dyn_in if self.nsdt > 0 else None)


class FWBPooling(nn.Module):
    """Fixed-width baseline (mvformer.py:421-463): the ntok "entities" are slices of ONE linear map of the frame's CLS
    embedding, no spatial pooling.  lin_conv(cls).reshape([F, -1, ntok]) is channel-major, so entity j of channel c is
    output column c * ntok + j."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        if 'SMART_TOKENS' not in cfg.MODEL.EMBEDDER_MODEL:
            print('Using default number of SMART_TOKENS: 5')
        if 'SMART_POOL_CHANNELS' not in cfg.MODEL.EMBEDDER_MODEL:
            print('Using default number of SMART_POOL_CHANNELS: 384')
        print('RUNNING FIXED WIDTH BASELINE')
        self.nst = _em(cfg, 'SMART_TOKENS', 5)
        self.nsdt = _em(cfg, 'SMART_DYNAMIC_TOKENS', 0)
        self.spc = _em(cfg, 'SMART_POOL_CHANNELS', 384)
        self.in_c = cfg.MODEL.BASE_MODEL.OUT_CHANNEL
        d_dyn_in = self.in_c
        if 'SMART_FEATS' in cfg.MODEL.EMBEDDER_MODEL:
            sfl = str(cfg.MODEL.EMBEDDER_MODEL.SMART_FEATS)
            if ',' in sfl:
                d_dyn_in = int(d_dyn_in / len(sfl.split(',')))
        self.lin_conv = nn.Linear(d_dyn_in, self.spc * (self.nst + self.nsdt))

    def forward(self, taps, cls_in):
        """-> [Bc, ntok, T, spc] rows in (clip, entity, frame) order, like LearnableTokenPooling."""
        if cls_in is None:
            raise ops._lib.MvfError('FIXED_WIDTH_BASELINE needs the backbone CLS embedding')
        tt = self.nst + self.nsdt
        y = ops.linear(cls_in, self.lin_conv.weight, self.lin_conv.bias)          # [Bc*T, spc*tt]
        return y.view(taps.n_clips, taps.n_frames, self.spc, tt).permute(0, 3, 1, 2).contiguous()


class MultiEntityTransformerEmbModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        print('Using Smart Pooling')
        self.cfg = cfg
        drop_rate = cfg.MODEL.EMBEDDER_MODEL.FC_DROPOUT_RATE
        self.drop_rate = drop_rate
        in_channels = _em(cfg, 'SMART_POOL_CHANNELS', 384)
        if _em(cfg, 'VAL_PASS', False):
            in_channels = cfg.MODEL.BASE_MODEL.OUT_CHANNEL
        self.nst = _em(cfg, 'SMART_TOKENS', 5)
        self.nsdt = _em(cfg, 'SMART_DYNAMIC_TOKENS', 0)
        self.one_hot_pos = _em(cfg, 'SMART_ONE_HOT', 'none')
        assert self.one_hot_pos in ['none', 'pool', 'enc']
        if self.one_hot_pos == 'pool':
            in_channels += (self.nst + self.nsdt)
        self.fwb = bool(_em(cfg, 'FIXED_WIDTH_BASELINE', False))
        cap_scalar = cfg.MODEL.EMBEDDER_MODEL.CAPACITY_SCALAR
        fc_params = _em(cfg, 'FC_LAYERS', None)
        self.embedding_size = cfg.MODEL.EMBEDDER_MODEL.EMBEDDING_SIZE
        hidden_channels = cfg.MODEL.EMBEDDER_MODEL.HIDDEN_SIZE
        self.pooling = FWBPooling(cfg) if self.fwb else LearnableTokenPooling(cfg)      # mvformer.py:63-67
        if fc_params is None:
            self.fc_layers = nn.Identity()
        else:
            layers = []
            for channels, _activate in fc_params:
                channels = channels * cap_scalar
                layers += [nn.Dropout(drop_rate), nn.Linear(in_channels, channels), nn.BatchNorm1d(channels), nn.ReLU(True)]
                in_channels = channels
            self.fc_layers = nn.Sequential(*layers)
        if self.one_hot_pos == 'enc':
            hidden_channels -= self.nst
        self.video_emb = nn.Linear(in_channels, hidden_channels)
        self.video_pos_enc = PositionalEncoder(cfg, hidden_channels, drop_rate, seq_len=cfg.TRAIN.NUM_FRAMES)
        if self.one_hot_pos == 'enc':
            hidden_channels += self.nst
        if cfg.MODEL.EMBEDDER_MODEL.NUM_LAYERS > 0:
            self.video_encoder = Encoder(hidden_channels, drop_rate, cfg.MODEL.EMBEDDER_MODEL.NUM_HEADS,
                                         cfg.MODEL.EMBEDDER_MODEL.D_FF, cfg.MODEL.EMBEDDER_MODEL.NUM_LAYERS)
            self.video_encoder.head_dtype = ops.head_dtype_of(cfg)
        self.embedding_layer = nn.Linear(hidden_channels, self.embedding_size)
        self.smart_final = _em(cfg, 'SMART_FINAL', 'max')
        assert self.smart_final in ['max', 'one', 'avg', 'lin']
        if self.smart_final == 'lin':
            self.lin_final = nn.Linear((self.nst + self.nsdt) * hidden_channels, hidden_channels)
        self.in_backbone_warmup = 'BACKBONE_WARMUP' in self.cfg.TRAIN
        self.drop_state = ops.DropoutState(seed=int(cfg.RNG_SEED) if 'RNG_SEED' in cfg else 0)
        self.sync_group = None
        # 'bf16': FC stack / video_emb and entity reduction / embedding layer on the row-chain kernels (csrc/head_rowlin.hip)
        self.head_dtype = ops.head_dtype_of(cfg)
        self._pack_trunk, self._pack_tail = ops.HeadPack(), ops.HeadPack()

    def set_warmup_status(self, new_status):
        self.in_backbone_warmup = new_status

    def invalidate_packed(self):
        """The fused optimizer updated the parameters through raw pointers: the bf16 operand copies are stale."""
        self._pack_trunk.invalidate()
        self._pack_tail.invalidate()

    def _bn_chainable(self, bn):
        return bn.momentum is not None and bn.affine and bn.track_running_stats

    def _bn_sync(self, bn):
        """RowLinStage.sync of the stage in front of `bn`: a SyncBatchNorm with live collectives exchanges its statistics between launches."""
        from ..utils.distributed import collectives_active
        return (self.sync_group,) if isinstance(bn, nn.SyncBatchNorm) and collectives_active() else None

    def trunk_chain_active(self, c_in=None):
        """The FC stack + video_emb run as row chains (one launch per Linear each way): bf16 head, the one-hot (if any) appended
        before the stack, every width within the kernels' panels (a SyncBatchNorm's exchange happens between the launches)."""
        if not ops.chain_dtype(self.head_dtype) or isinstance(self.fc_layers, nn.Identity) or self.one_hot_pos == 'enc':
            return False
        mods = list(self.fc_layers)
        lins = [mods[i + 1] for i in range(0, len(mods), 4)] + [self.video_emb]
        if not all(self._bn_chainable(mods[i + 2]) for i in range(0, len(mods), 4)):
            return False
        k0 = lins[0].weight.shape[1] - ((self.nst + self.nsdt) if self.one_hot_pos == 'pool' else 0)
        if c_in is not None and c_in != k0:
            return False
        return k0 % 4 == 0 and all(ops.rowlin_supported(l.weight.shape[1] if j else k0, l.weight.shape[0]) and l.weight.shape[1] <= 512
                                   for j, l in enumerate(lins))

    def tail_chain_active(self):
        return ops.chain_dtype(self.head_dtype) and self.smart_final in ('one', 'avg', 'max') and \
            ops.rowlin_supported(self.embedding_layer.weight.shape[1], self.embedding_layer.weight.shape[0])

    def _trunk_chain(self, x, ntok, T):
        """[rows (clip, entity, frame), c] -> [rows, hidden]: one-hot, (dropout, Linear, BatchNorm, ReLU) x n, video_emb + sin/cos
        table, dropout -- mvformer.py:144-160 -- as n + 1 launches."""
        mods = list(self.fc_layers)
        nblk = len(mods) // 4
        params, stages, eval_stats = [], [], []
        prev = None
        for i in range(nblk):
            drop, lin, bn = mods[4 * i], mods[4 * i + 1], mods[4 * i + 2]
            if bn.training != self.training:
                bn.train(self.training)
            iw = len(params)
            params += [lin.weight, lin.bias]
            kin = lin.weight.shape[1]
            st = ops.RowLinStage(iw, iw + 1, bn_in=prev, onehot=(ntok, T) if (i == 0 and self.one_hot_pos == 'pool') else None,
                                 drop_in=ops.drop_args(drop.p, self.training, self.drop_state, x.shape[0] * kin),
                                 bn_out=(bn.running_mean, bn.running_var, bn.momentum), sync=self._bn_sync(bn))
            eval_stats.append(None if i == 0 else (mods[4 * i - 2].running_mean, mods[4 * i - 2].running_var))
            stages.append(st)
            ig = len(params)
            params += [bn.weight, bn.bias]
            prev = (ig, ig + 1, bn.eps, True)
        iw = len(params)
        params += [self.video_emb.weight, self.video_emb.bias]
        last_bn = mods[4 * nblk - 2]
        stages.append(ops.RowLinStage(iw, iw + 1, bn_in=prev, table=(self.video_pos_enc.table(T, x.device), T),
                                      drop_out=ops.drop_args(self.video_pos_enc.dout_p, self.training, self.drop_state,
                                                             x.shape[0] * self.video_emb.weight.shape[0])))
        eval_stats.append((last_bn.running_mean, last_bn.running_var))
        self._pack_trunk.set_f16(self.head_dtype == 'fp16')
        return ops.rowlin_chain(x, stages, params, self.training, self._pack_trunk, tuple(eval_stats), tag='trunk.')

    def _bn(self, x, bn, relu):
        return ops.batch_norm(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, self.training and bn.training,
                              momentum=bn.momentum, eps=bn.eps, relu=relu, sync=isinstance(bn, nn.SyncBatchNorm),
                              group=self.sync_group)

    def forward(self, x, video_masks=None, cls_emb=None):
        taps = x if isinstance(x, Taps) else Taps.from_nchw(x, len(str(_em(self.cfg, 'SMART_FEATS', '11')).split(',')))
        Bc, T = taps.n_clips, taps.n_frames
        if self.in_backbone_warmup:      # mvformer.py:131-132: the spatial features (not cls_emb) are detached during the
            taps = Taps([t.detach() for t in taps.tensors], taps.n_clips, taps.n_frames, taps.n_tokens)   # warm-up epochs
        x = self.pooling(taps, cls_emb)                                    # [Bc, ntok, T, c]
        ntok = x.shape[1]
        x = x.reshape(Bc * ntok * T, -1)
        if x.is_cuda and self.trunk_chain_active(x.shape[1]):
            x = self._trunk_chain(x, ntok, T)
        else:
            if self.one_hot_pos == 'pool':
                x = ops.concat_onehot(x, ntok, T)
            if not isinstance(self.fc_layers, nn.Identity):
                mods = list(self.fc_layers)
                for i in range(0, len(mods), 4):
                    x = ops.dropout_add(x, None, mods[i].p, self.training, self.drop_state)
                    bn = mods[i + 2]
                    if bn.training != self.training:
                        bn.train(self.training)
                    x = self._bn(ops.linear(x, mods[i + 1].weight, mods[i + 1].bias), bn, relu=True)
            # video_emb GEMM with the sin/cos table added in its epilogue (row r -> frame r % T), then PE dropout
            x = ops.linear(x, self.video_emb.weight, self.video_emb.bias, table=self.video_pos_enc.table(T, x.device),
                           tab_div=1, tab_mod=T)
            x = ops.dropout_add(x, None, self.video_pos_enc.dout_p, self.training, self.drop_state)
        if self.one_hot_pos == 'enc':
            x = ops.concat_onehot(x, ntok, T)
        x = x.view(Bc, ntok * T, -1)
        if self.cfg.MODEL.EMBEDDER_MODEL.NUM_LAYERS > 0:
            # mvformer.py:171-172 tiles the frame mask to the joint sequence ([Bc, 1, ntok * T]); the attention kernels read the
            # [Bc, T] mask periodically instead (key (j, t) -> column t), so nothing is copied
            vm = video_masks.reshape(Bc, 1, T) if video_masks is not None else None
            x = self.video_encoder(x, src_mask=vm, drop_state=self.drop_state)
        if x.is_cuda and self.tail_chain_active():
            # entity reduction + embedding layer (mvformer.py:181-199) as one launch each way
            st = ops.RowLinStage(0, 1, gather=(ntok, T, {'one': 0, 'avg': 1, 'max': 2}[self.smart_final]))
            self._pack_tail.set_f16(self.head_dtype == 'fp16')
            x = ops.rowlin_chain(x.reshape(Bc * ntok * T, -1), [st], [self.embedding_layer.weight, self.embedding_layer.bias],
                                 self.training, self._pack_tail, tag='tail.')
            return x.view(Bc, T, self.embedding_size)
        x = x.view(Bc, ntok, T, -1)
        if self.smart_final == 'lin':
            x = x.permute(0, 2, 1, 3).reshape(Bc * T, -1)
            x = ops.linear(x, self.lin_final.weight, self.lin_final.bias)
        else:
            x = ops.final_reduce(x, self.smart_final)
        x = ops.linear(x.reshape(Bc * T, -1), self.embedding_layer.weight, self.embedding_layer.bias)
        return x.view(Bc, T, self.embedding_size)
