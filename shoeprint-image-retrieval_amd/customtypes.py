"""Array aliases at the boundary (reference customtypes.py:7-16): 2-D float image / feature map,
3-D float stack of feature maps [C, h, w], and the dataset-kind literal."""

from typing import Any, Literal, TypeAlias

import numpy as np

ImageArrayType: TypeAlias = np.ndarray[tuple[int, int], np.dtype[np.floating[Any]]]
FeatureMapsArrayType: TypeAlias = np.ndarray[tuple[int, int, int], np.dtype[np.floating[Any]]]
DatasetTypeType: TypeAlias = Literal["FID-300", "Impress", "WVU2019"]
