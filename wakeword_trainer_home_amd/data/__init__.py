from .feature_extraction import FeatureExtractor
from .augmentation import AudioAugmentation, SpecAugment
from .dataset import SyntheticClipDataset, make_synthetic_batch

__all__ = ["FeatureExtractor", "SpecAugment", "SyntheticClipDataset", "make_synthetic_batch"]
