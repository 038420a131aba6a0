"""Model registry / construction API (mirrors casapose/pose_models of the reference)."""
from .models_factory import Classifiers, ModelsFactory  # noqa: F401
