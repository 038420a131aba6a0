"""`from casapose.pose_models.tfkeras import Classifiers` of the reference
(casapose/pose_models/tfkeras.py:6-17) -- same name, MI355X backend."""
from .models_factory import Classifiers, ModelsFactory as TFKerasModelsFactory  # noqa: F401
