"""Test infrastructure (NOT part of the product): pins the device-side quantisation step against the third-party code the reference
calls for it.  The reference fits `sklearn.cluster.KMeans(n_clusters).fit(pixels / 255)` (utils.py:287) and labels every frame with
`kmeans.predict(x.view(-1, 1))` behind `ToTensor` (main.py:21-38); scikit-learn is not vendored in the reference repository.  This script
runs exactly those two calls (scikit-learn as installed in the build container) on synthetic MovingMNIST-like frames and stores inputs
and expected outputs -- data only -- in tests/golden/kmeans_q{2,4}.npz:
  frames uint8 (5, 64, 64), centres float64 (q,) = cluster_centers_.ravel(), labels uint8 (5, 64, 64) = predict(ToTensor(frames)),
  data_mean / data_std as utils.py:296-305 computes them (mean / std of the labels), sklearn version.
usage: python oracle/make_kmeans_fixture.py"""
import os

import numpy as np
import sklearn
from sklearn.cluster import KMeans

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def frames(rng):
    """Dark background, a few bright soft-edged digits' worth of blobs, plus one ramp frame holding every uint8 value."""
    f = np.zeros((5, 64, 64), np.float32)
    yy, xx = np.mgrid[0:64, 0:64]
    for i in range(4):
        for _ in range(2):
            cy, cx, r = rng.uniform(12, 52), rng.uniform(12, 52), rng.uniform(5, 9)
            f[i] = np.maximum(f[i], np.clip(1.25 - np.hypot(yy - cy, xx - cx) / r, 0, 1))
        f[i] += rng.normal(0, 0.01, (64, 64)).clip(0, 1)
    out = (f.clip(0, 1) * 255).round().astype(np.uint8)
    out[4] = (np.arange(4096) % 256).reshape(64, 64).astype(np.uint8)
    return out


def main():
    rng = np.random.default_rng(1234)
    fr = frames(rng)
    x = fr.astype(np.float32) / 255.0                          # ToTensor
    for q in (2, 4):
        km = KMeans(n_clusters=q, n_init=3, random_state=q).fit((fr[:4].reshape(-1, 1) / 255))       # utils.py:284-287
        # main.py:25 hands predict() the float32 ToTensor values; scikit-learn converts them to the dtype it was fitted in (float64) --
        # releases that no longer do so implicitly need the cast spelled out
        labels = km.predict(x.reshape(-1, 1).astype(np.float64)).reshape(fr.shape)
        np.savez_compressed(os.path.join(OUT, f"kmeans_q{q}.npz"), frames=fr, centres=km.cluster_centers_.ravel(),
                            labels=labels.astype(np.uint8), data_mean=np.float64(round(labels.mean(), 4)),
                            data_std=np.float64(round(labels.std(), 4)), sklearn_version=np.array(sklearn.__version__))
        print(q, km.cluster_centers_.ravel(), np.bincount(labels.ravel()))


if __name__ == "__main__":
    main()
