from .synthetic import SyntheticCOCODataset, BatchCollator, make_data_loader  # noqa: F401
