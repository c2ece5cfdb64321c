import json

import numpy as np

from shoeprint_image_retrieval_amd import feature_cache as fc


def test_round_trip_ragged_and_key(tmp_path):
    rng = np.random.default_rng(0)
    maps = [rng.standard_normal(s).astype(np.float32) for s in ((3, 8, 5), (3, 9, 4), (3, 8, 5))]
    path = str(tmp_path / "gallery.f32")
    key = {"block": 16, "scale": 0.78125, "crop": [0.1, 0.2], "files": ["001.png", "002.png", "003.png"]}
    fc.save_features(path, maps, key)
    got = fc.load_features(path, key)
    assert len(got) == 3
    for a, b in zip(maps, got):
        np.testing.assert_array_equal(a, b)
        assert b.dtype == np.float32 and b.flags["C_CONTIGUOUS"]
    assert fc.load_features(path, dict(key, block=15)) is None          # other extraction: not this cache
    assert fc.load_features(str(tmp_path / "missing"), key) is None
    with open(path, "ab") as fh:                                        # size no longer matches the index
        fh.write(b"x")
    assert fc.load_features(path, key) is None


def test_float16_and_empty(tmp_path):
    maps = [np.arange(24, dtype=np.float32).reshape(2, 3, 4)]
    path = str(tmp_path / "g.f16")
    fc.save_features(path, maps, dtype=np.float16)
    got = fc.load_features(path)
    assert got[0].dtype == np.float16
    np.testing.assert_array_equal(got[0], maps[0].astype(np.float16))
    assert json.load(open(path + ".json"))["items"][0]["offset"] % 256 == 0
    fc.save_features(path, [])
    assert fc.load_features(path) == []
