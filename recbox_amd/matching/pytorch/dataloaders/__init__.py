from .h5_generator import TrainGenerator, collate, collate_unique, negative_sampling  # noqa: F401
