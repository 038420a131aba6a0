"""Alias: the reference README spells the module `model_factory` (README.md:116); the real
module is models_factory.  Both import."""
from .models_factory import *  # noqa: F401,F403
from .models_factory import Classifiers, ModelsFactory  # noqa: F401
