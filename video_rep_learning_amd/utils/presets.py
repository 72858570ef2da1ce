"""In-repo restatement of the MV-Former experiment settings so that bench / smoke / tests do not need the
reference's YAML files at run time (they are absent on the GPU box).  `penn_mvf()` is the content of
CARL_MVF/configs_mvf/penn_mvf.yml as a dict (tests/test_config.py asserts equality with the real file when the
reference is mounted); it is applied with the same shallow `cfg.update` a YAML would get."""
from .config import get_cfg

_DATASETS_PENN = ['baseball_pitch', 'baseball_swing', 'bench_press', 'bowl', 'clean_and_jerk', 'golf_swing',
                  'jumping_jacks', 'pushup', 'pullup', 'situp', 'squat', 'tennis_forehand', 'tennis_serve']


def penn_mvf():
    return {
        'SSL': True, 'USE_AMP': True,
        'AUGMENTATION': {'STRENGTH': 1.0, 'BRIGHTNESS': True, 'BRIGHTNESS_MAX_DELTA': 0.8, 'CONTRAST': True,
                         'CONTRAST_MAX_DELTA': 0.8, 'HUE': True, 'HUE_MAX_DELTA': 0.2, 'RANDOM_CROP': True,
                         'RANDOM_FLIP': True, 'SATURATION': True, 'SATURATION_MAX_DELTA': 0.8},
        'CHECKPOINT': {'SAVE_INTERVAL': 20},
        'DATA': {'FRAME_LABELS': True, 'NUM_CONTEXTS': 1, 'CONTEXT_STRIDE': 1, 'NUM_WORKERS': 16, 'SAMPLE_ALL_STRIDE': 1,
                 'SAMPLING_STRATEGY': 'time_augment', 'SAMPLING_REGION': 1.5, 'CONSISTENT_OFFSET': 0.2},
        'DATASETS': list(_DATASETS_PENN),
        'EVAL': {'BATCH_SIZE': 1, 'CLASSIFICATION_FRACTIONS': [1.0], 'FRAMES_PER_BATCH': 1000,
                 'KENDALLS_TAU_DISTANCE': 'sqeuclidean', 'KENDALLS_TAU_STRIDE': 2, 'RETRIEVAL_KS': [5, 10, 15],
                 'NUM_FRAMES': 80, 'TASKS': ['kendalls_tau', 'retrieval', 'classification', 'event_completion'],
                 'VAL_INTERVAL': 50},
        'IMAGE_SIZE': 224,
        'LOGDIR': '/tmp/scl_transformer_action_logs',
        'LOGGING': {'REPORT_INTERVAL': 20},
        'MODEL': {
            'BASE_MODEL': {'LAYER': 12, 'NETWORK': 'TIMM-vit_base_patch8_224.dino', 'FRAMES_PER_BATCH': 40},
            'EMBEDDER_MODEL': {
                'HIDDEN_SIZE': 256, 'D_FF': 1024, 'NUM_HEADS': 8, 'NUM_LAYERS': 3, 'CAPACITY_SCALAR': 2,
                'CONV_LAYERS': [[256, 3, 1], [256, 3, 1]], 'EMBEDDING_SIZE': 128, 'FC_DROPOUT_RATE': 0.1,
                'FC_LAYERS': [[256, True], [256, True]], 'FLATTEN_METHOD': 'max_pool', 'USE_BN': True,
                'FUSION_TYPE': 'smart', 'SMART_TOKENS': 3, 'SMART_ONE_HOT': 'pool', 'SMART_FEATS': '3,7,11',
                'SMART_FINAL': 'one'},
            'EMBEDDER_TYPE': 'transformer', 'L2_NORMALIZE': True, 'PROJECTION': True, 'PROJECTION_HIDDEN_SIZE': 512,
            'PROJECTION_SIZE': 128, 'TRAIN_BASE': 'frozen'},
        'NUM_GPUS': 1,
        'OPTIMIZER': {'GRAD_CLIP': 10, 'LR': {'DECAY_TYPE': 'cosine', 'INITIAL_LR': 0.0001, 'NUM_WARMUP_STEPS': 1,
                                              'WARMUP_LR': 0.0, 'FINAL_LR': 0.0},
                      'TYPE': 'AdamOptimizer', 'WEIGHT_DECAY': 1.0e-05},
        'PATH_TO_DATASET': 'penn_action',
        'RNG_SEED': 1, 'SHARD_ID': 0,
        'SCL': {'LABEL_VARIENCE': 10.0, 'POSITIVE_TYPE': 'gauss', 'NEGATIVE_TYPE': 'single_noself',
                'SOFTMAX_TEMPERATURE': 0.1, 'POSITIVE_WINDOW': 5},
        'TRAIN': {'BATCH_SIZE': 1, 'MAX_EPOCHS': 500, 'NUM_FRAMES': 80},
        'TRAINING_ALGO': 'scl',
    }


def make_cfg(base=None, network=None, num_frames=None, batch_size=None, image_size=None, compute_dtype=None,
             dropout=None, head_dtype=None, **embedder_overrides):
    """Defaults <- preset (shallow update, like a YAML) <- the equivalents of `--opts` overrides."""
    cfg = get_cfg()
    cfg.update(penn_mvf() if base is None else base)
    if network is not None:
        cfg.MODEL.BASE_MODEL.NETWORK = network
    if num_frames is not None:
        cfg.TRAIN.NUM_FRAMES = num_frames
    if batch_size is not None:
        cfg.TRAIN.BATCH_SIZE = batch_size
    if image_size is not None:
        cfg.IMAGE_SIZE = image_size
    if dropout is not None:
        cfg.MODEL.EMBEDDER_MODEL.FC_DROPOUT_RATE = dropout
    for k, v in embedder_overrides.items():
        cfg.MODEL.EMBEDDER_MODEL[k] = v
    if compute_dtype is not None:
        cfg.MI355X = {'COMPUTE_DTYPE': compute_dtype}
    if head_dtype is not None:          # 'bf16' | 'fp32': the trainable head's GEMM operand dtype (default: ops.head_dtype_of)
        mi = dict(cfg.MI355X) if 'MI355X' in cfg else {}
        mi['HEAD_DTYPE'] = head_dtype
        cfg.MI355X = mi
    cfg.EVAL.BATCH_SIZE = cfg.TRAIN.BATCH_SIZE
    cfg.EVAL.NUM_FRAMES = cfg.TRAIN.NUM_FRAMES
    return cfg


def baseline_config_2(compute_dtype='bf16'):
    """BASELINE.json configs[1]: PennAction MV-Former, ViT-B/16, 32 frames, batch 4 per GPU
    (penn_mvf.yml + --opts MODEL.BASE_MODEL.NETWORK TIMM-vit_base_patch16_224.dino TRAIN.NUM_FRAMES 32 TRAIN.BATCH_SIZE 4)."""
    return make_cfg(network='TIMM-vit_base_patch16_224.dino', num_frames=32, batch_size=4, compute_dtype=compute_dtype)
