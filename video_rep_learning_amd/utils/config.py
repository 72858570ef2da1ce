"""Experiment configuration: attribute-access dict + the default tree.

Mirrors CARL_MVF/utils/config.py:6-247 (the defaults `load_config` starts from) so that the reference's
`configs/*.yml` and `configs_mvf/*.yml` drop in unchanged.  `easydict` is not a dependency here: `EasyDict`
below re-implements the subset of its behaviour the code base relies on (attribute get/set, recursive
conversion of nested dicts on construction / assignment / update, `in`, iteration as a dict)."""


class EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def update(self, other=None, **kw):
        # shallow, like easydict/dict.update: a nested dict in `other` REPLACES the default sub-tree
        # (CARL_MVF/utils/parser.py:78 relies on this: penn_mvf.yml's MODEL block replaces the default MODEL)
        for k, v in dict(other or {}, **kw).items():
            self[k] = v


def default_dict():
    """Plain-dict restatement of the defaults (CARL_MVF/utils/config.py)."""
    return {
        'LOGDIR': '/tmp/scl_transformer_logs',
        'DATASETS': ['pouring'],
        'SSL': True,
        'PATH_TO_DATASET': 'pouring',
        'TRAINING_ALGO': 'scl',
        'IMAGE_SIZE': 224,
        'NUM_GPUS': 1,
        'SHARD_ID': 0,
        'RNG_SEED': 1,
        'TRAIN': {'MAX_EPOCHS': 500, 'BATCH_SIZE': 1, 'NUM_FRAMES': 240},
        'EVAL': {
            'BATCH_SIZE': 1, 'NUM_FRAMES': 240, 'VAL_INTERVAL': 50,
            'TASKS': ['kendalls_tau', 'retrieval', 'classification', 'event_completion'],
            'FRAMES_PER_BATCH': 1000, 'KENDALLS_TAU_STRIDE': 5, 'KENDALLS_TAU_DISTANCE': 'sqeuclidean',
            'CLASSIFICATION_FRACTIONS': [0.1, 0.5, 1.0], 'RETRIEVAL_KS': [5, 10, 15],
        },
        'MODEL': {
            'EMBEDDER_TYPE': 'transformer',
            'BASE_MODEL': {'NETWORK': 'Resnet50_byol', 'LAYER': 3, 'FRAMES_PER_BATCH': 40},
            'TRAIN_BASE': 'frozen',
            'EMBEDDER_MODEL': {
                'HIDDEN_SIZE': 256, 'D_FF': 1024, 'NUM_HEADS': 8, 'NUM_LAYERS': 3,
                'CONV_LAYERS': [(256, 3, 1), (256, 3, 1)], 'FLATTEN_METHOD': 'max_pool',
                'FC_LAYERS': [(256, True), (256, True)], 'CAPACITY_SCALAR': 2, 'EMBEDDING_SIZE': 128,
                'FC_DROPOUT_RATE': 0.1, 'USE_BN': True,
            },
            'L2_NORMALIZE': True, 'PROJECTION': True, 'PROJECTION_HIDDEN_SIZE': 512, 'PROJECTION_SIZE': 128,
        },
        'SCL': {'LABEL_VARIENCE': 10.0, 'SOFTMAX_TEMPERATURE': 0.1, 'POSITIVE_TYPE': 'gauss',
                'NEGATIVE_TYPE': 'single_noself', 'POSITIVE_WINDOW': 5},
        'TCC': {'CYCLE_LENGTH': 2, 'LABEL_SMOOTHING': 0.1, 'SOFTMAX_TEMPERATURE': 0.1,
                'LOSS_TYPE': 'regression_mse_var', 'NORMALIZE_INDICES': True, 'VARIANCE_LAMBDA': 0.001,
                'FRACTION': 1.0, 'HUBER_DELTA': 0.1, 'SIMILARITY_TYPE': 'l2'},
        'TCN': {'POSITIVE_WINDOW': 5, 'REG_LAMBDA': 0.002},
        'OPTIMIZER': {'TYPE': 'AdamOptimizer', 'WEIGHT_DECAY': 0.00001, 'GRAD_CLIP': 10,
                      'LR': {'INITIAL_LR': 0.0001, 'DECAY_TYPE': 'cosine', 'WARMUP_LR': 0.0001, 'FINAL_LR': 0.0,
                             'NUM_WARMUP_STEPS': 1}},
        'DATA': {'FRACTION': 1.0, 'ADDITION_TRAINSET': False, 'SAMPLING_STRATEGY': 'time_augment', 'NUM_CONTEXTS': 1,
                 'CONTEXT_STRIDE': 1, 'SAMPLING_REGION': 1.5, 'CONSISTENT_OFFSET': 0.2, 'FRAME_LABELS': True,
                 'SAMPLE_ALL_STRIDE': 1, 'NUM_WORKERS': 4},
        'AUGMENTATION': {'STRENGTH': 1.0, 'RANDOM_FLIP': True, 'RANDOM_CROP': True, 'BRIGHTNESS': True,
                         'BRIGHTNESS_MAX_DELTA': 0.8, 'CONTRAST': True, 'CONTRAST_MAX_DELTA': 0.8, 'HUE': True,
                         'HUE_MAX_DELTA': 0.2, 'SATURATION': True, 'SATURATION_MAX_DELTA': 0.8},
        'LOGGING': {'REPORT_INTERVAL': 20},
        'CHECKPOINT': {'SAVE_INTERVAL': 50},
    }


def get_cfg():
    """A fresh copy of the default config (the reference hands out its module-level singleton; a copy keeps
    repeated `load_config` calls in one process -- tests, bench -- independent)."""
    return EasyDict(default_dict())


# ---- build-specific (optional) keys, all probed with `in` so reference YAMLs stay valid -------------
# cfg.MI355X.COMPUTE_DTYPE   'bf16' | 'fp32' | 'fp16' | 'fp8'   backbone compute dtype (default bf16 under USE_AMP, else fp32 = parity
#                            mode; fp16 = the reference's own autocast dtype, frozen backbones only; fp8 = MX-fp8 GEMM operands)
# cfg.MI355X.HEAD_DTYPE      'bf16' | 'fp32'    the trainable head's Linears: bf16 operands on the matrix cores in row-chain kernels (fp32 master
#                            weights, statistics, loss, optimizer) -- default beside a bf16 / fp8 backbone or USE_AMP -- or the fp32 kernels
#                            (default in fp32 / fp16 mode)
# cfg.MI355X.FRAMES_PER_CHUNK int              frames per backbone pass (0 = MODEL.BASE_MODEL.FRAMES_PER_BATCH*clips)
# cfg.MI355X.GATHER_EMBEDDINGS bool            cross-GPU embedding all-gather for the SCL negatives
# cfg.MODEL.BASE_MODEL.WEIGHTS path            timm-format state dict for the backbone (no network download)
