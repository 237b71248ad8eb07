"""Mirror of torchreid.metrics (reference torchreid/metrics/{distance,rank,accuracy}.py)."""
from .accuracy import accuracy  # noqa: F401
from .distance import compute_distance_matrix, cosine_distance, euclidean_squared_distance  # noqa: F401
from .rank import evaluate_rank  # noqa: F401
