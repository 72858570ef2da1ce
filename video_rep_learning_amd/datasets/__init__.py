from .synthetic import SyntheticClips, construct_dataloader, get_data_preprocess  # noqa: F401
