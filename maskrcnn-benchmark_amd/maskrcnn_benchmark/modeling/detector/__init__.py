from .detectors import build_detection_model  # noqa: F401
