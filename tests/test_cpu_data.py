"""Record databases of the reference's LMDB-cached datasets (reference fullbatch/data/lmdb_datasets.py): the reader
``fullbatchtraining_amd.data.LMDBRecords`` against stores the REFERENCE's own writer filled and items its own reader returned
(tests/golden/make_golden.py --r2 ran both against the dict-backed stand-in for the absent third-party `lmdb` package)."""
import numpy as np
import pytest
import torch

from tests.helpers import DictLMDB


def _store(data, tag):
    keys = [k.encode("latin1") for k in data[f"lmdb/{tag}/keys"]]
    lengths = data[f"lmdb/{tag}/value_lengths"]
    blob = data[f"lmdb/{tag}/values"].tobytes()
    store, pos = {}, 0
    for k, n in zip(keys, lengths):
        store[k] = blob[pos:pos + int(n)]
        pos += int(n)
    assert pos == len(blob)
    return store


@pytest.mark.parametrize("tag,rounds,chw", [("chw_r1", 1, True), ("hwc_r2", 2, False)])
@pytest.mark.parametrize("access", ["get", "cursor"])
def test_records_written_by_the_reference(golden, tag, rounds, chw, access):
    from fullbatchtraining_amd.data import LMDBRecords

    data, _ = golden
    base, labels = data["lmdb/base_images"], data["lmdb/base_labels"]
    rec = LMDBRecords(env=DictLMDB(_store(data, tag)), access=access)
    n = len(labels)
    assert len(rec) == rounds * n and rec.channels_first == chw
    assert rec.shape == ((3, 32, 32) if chw else (32, 32, 3))
    want = torch.from_numpy(base).permute(0, 3, 1, 2) if chw else torch.from_numpy(base)
    for i in (0, 1, 11, n - 1, 5, 10, 2):                      # out of order: the cursor has to re-seek (keys sort as b"0" < b"1" < b"10" < b"11" < b"2")
        block, label = rec[i]
        assert block.dtype == torch.uint8 and torch.equal(block, want[i]) and label == int(labels[i])
    if rounds == 2:                                             # N x CIFAR: the second round follows the first
        block, label = rec[n + 3]
        assert torch.equal(block, want[3]) and label == int(labels[3])
    images = rec.images_uint8()
    assert images.shape == (rounds * n, 3, 32, 32)
    assert torch.equal(images[:n], torch.from_numpy(base).permute(0, 3, 1, 2))
    assert torch.equal(images[-n:], torch.from_numpy(base).permute(0, 3, 1, 2))
    x, y = rec.as_feed(*data["lmdb/mean_std"])
    assert x.dtype == torch.float32 and y.dtype == torch.long and y.tolist() == list(labels) * rounds
    if chw:        # what the REFERENCE's reader returned for these indices (uint8 / 255, then the live Normalize)
        idx = data["lmdb/chw_r1/item_index"]
        assert np.array_equal(data["lmdb/chw_r1/item_labels"], labels[idx])
        assert np.allclose(x[idx].numpy(), data["lmdb/chw_r1/item_images"], rtol=1e-6, atol=1e-7)


def test_unfinished_database_and_missing_package(golden):
    from fullbatchtraining_amd.data import LMDBRecords

    data, _ = golden
    store = _store(data, "chw_r1")
    del store[b"__len__"]                                       # the writer puts the four summary records last
    with pytest.raises(ValueError):
        LMDBRecords(env=DictLMDB(store))
    try:
        import lmdb  # noqa: F401
    except ImportError:
        with pytest.raises(RuntimeError):                       # no silent stand-in for the third-party store
            LMDBRecords(path="/nonexistent/db.lmdb")
    with pytest.raises(ValueError):
        LMDBRecords()
