"""Diversity statistic of generated grasps (reference: diverse_grasp/diversity.py:7-15): k-means (20 clusters) over
the [n,61] parameter vectors, entropy of the cluster histogram and mean distance to the assigned centre.
Offline analysis on the host, as in the reference (scipy)."""
import json
from typing import Iterable, Tuple

import numpy as np
import scipy.cluster.vq
from scipy.stats import entropy


def diversity(params_list, cls_num: int = 20, seed: int = 0) -> Tuple[float, float]:
    x = np.asarray(params_list, dtype=np.float64)
    codes, _ = scipy.cluster.vq.kmeans(x, cls_num, seed=seed)
    vecs, dist = scipy.cluster.vq.vq(x, codes)
    counts, _ = np.histogram(vecs, len(codes))
    return float(entropy(counts)), float(np.mean(dist))


def load_params(json_paths: Iterable[str]) -> np.ndarray:
    """recon_params[i][0] of every file, as diverse_grasp/diversity.py:30-41 reads them."""
    rows = []
    for p in json_paths:
        with open(p) as f:
            data = json.load(f)
        rows += [g[0] for g in data["recon_params"]]
    return np.asarray(rows)
